"""What the evaluator asks of a metric (the interface the reference declares in evaluation/protocols.py:5-17).

A metric object has a ``name`` (the key of its result in ``MetricEvaluator.evaluations``), a ``calculate(y_true, y_score)``
that takes the per-impression label and score arrays and returns one float, and is callable with the same arguments.
``MetricEvaluator`` only ever uses ``name`` and the call.
"""
from __future__ import annotations

from typing import Protocol, Sequence, runtime_checkable


@runtime_checkable
class Metric(Protocol):
    """Structural type: anything with ``name`` and ``calculate`` is a metric; subclassing this class adds the call
    forwarding and a printable form."""

    name: str

    def calculate(self, y_true: Sequence, y_score: Sequence) -> float:
        """one float from the impressions' labels and scores"""
        ...

    def __call__(self, y_true: Sequence, y_score: Sequence) -> float:
        return self.calculate(y_true, y_score)

    def _settings(self) -> dict:
        return {k: v for k, v in vars(self).items() if k != "name"}

    def __repr__(self) -> str:
        extra = ", ".join(f"{k}={v!r}" for k, v in sorted(self._settings().items()))
        return f"{type(self).__name__}(name={getattr(self, 'name', None)!r}{', ' + extra if extra else ''})"

    __str__ = __repr__
