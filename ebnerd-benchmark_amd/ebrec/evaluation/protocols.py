"""What the evaluator asks of a metric (the interface the reference declares in evaluation/protocols.py:5-17).

A metric object has a ``name`` (the key of its result in ``MetricEvaluator.evaluations``) and a
``calculate(y_true, y_score)`` that takes the per-impression label and score arrays and returns one float.  That pair is
the whole structural type: ``isinstance(x, Metric)`` holds for ANY object with those two members.  The call forwarding
and the printed form the reference's metrics have live in ``MetricBase``, which the concrete metrics of this package
derive from -- outside the Protocol, so that they are not requirements on third-party metric objects.
"""
from __future__ import annotations

from typing import Protocol, Sequence, runtime_checkable


@runtime_checkable
class Metric(Protocol):
    name: str

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        """one float from the impressions' labels and scores"""
        ...


class MetricBase:
    """Shared behaviour of this package's metrics: calling the object runs ``calculate``; ``str`` / ``repr`` print the
    reference's form ``<Callable Metric: name>: params: {...}`` (evaluation/protocols.py:10-14)."""

    name: str = ""

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        raise NotImplementedError(f"{type(self).__name__} does not define calculate()")

    def __call__(self, y_true: Sequence, y_score: Sequence) -> float:
        return self.calculate(y_true, y_score)

    def __str__(self) -> str:
        return f"<Callable Metric: {self.name}>: params: {vars(self)}"

    __repr__ = __str__
