"""What the evaluator asks of a metric (the interface the reference declares in evaluation/protocols.py:5-17).

``Metric`` is the reference's type: a ``name`` (the key of the result in ``MetricEvaluator.evaluations``), a
``calculate(y_true, y_score)`` returning one float, and -- as DEFAULT bodies, exactly where the reference keeps them
(protocols.py:10-17) -- the call forwarding and the printed form ``<Callable Metric: name>: params: {...}``.  A user metric
written the reference's way, ``class Foo(Metric): def calculate(...)``, is therefore callable and accepted by
``MetricEvaluator``.

``MetricLike`` is the structural check (``isinstance(x, MetricLike)`` holds for ANY object with ``name`` + ``calculate``,
base class or not); ``MetricBase`` is the concrete base this package's metrics derive from.
"""
from __future__ import annotations

from typing import Protocol, Sequence, runtime_checkable


@runtime_checkable
class MetricLike(Protocol):
    name: str

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        """one float from the impressions' labels and scores"""
        ...


@runtime_checkable
class Metric(Protocol):
    name: str

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        """one float from the impressions' labels and scores"""
        ...

    def __call__(self, y_true: Sequence, y_score: Sequence) -> float:
        return self.calculate(y_true, y_score)

    def __str__(self) -> str:
        return f"<Callable Metric: {self.name}>: params: {vars(self)}"

    def __repr__(self) -> str:
        return str(self)


class MetricBase(Metric):
    """Base of this package's metrics: ``Metric``'s defaults plus a ``calculate`` that says what is missing."""

    name: str = ""

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        raise NotImplementedError(f"{type(self).__name__} does not define calculate()")
