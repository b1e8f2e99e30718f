"""Per-impression metric wrappers and the MetricEvaluator (reference: evaluation/metrics_protocols.py).

Each wrapper is the mean over impressions of a 1-D metric -- ``np.mean([metric(l, p) for l, p in zip(labels,
predictions)])`` (metrics_protocols.py:77-86 etc.).  Same names (``auc``, ``mrr``, ``ndcg@k``, ``logloss``, ``rmse``,
``accuracy``, ``f1``), same clipping in LogLossScore (``[10e-12, 1 - 10e-12]``, line 99), same in-place binarisation
side effect of the threshold metrics.  Beyond-accuracy metrics are out of scope.
"""
from __future__ import annotations

import json

import numpy as np

from .metrics import accuracy_score, f1_score, log_loss, mean_squared_error, mrr_score, ndcg_score, roc_auc_score
from .protocols import Metric, MetricBase  # noqa: F401  (Metric: the structural type, re-exported)
from .utils import convert_to_binary


def _mean_over_impressions(fn, y_true, y_pred) -> float:
    return float(np.mean([fn(labels, preds) for labels, preds in zip(y_true, y_pred)]))


class AccuracyScore(MetricBase):
    def __init__(self, threshold: float = 0.5):
        self.threshold = threshold
        self.name = "accuracy"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: accuracy_score(l, convert_to_binary(p, self.threshold)), y_true, y_pred)


class F1Score(MetricBase):
    def __init__(self, threshold: float = 0.5):
        self.threshold = threshold
        self.name = "f1"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: f1_score(l, convert_to_binary(p, self.threshold)), y_true, y_pred)


class RootMeanSquaredError(MetricBase):
    def __init__(self):
        self.name = "rmse"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: np.sqrt(mean_squared_error(l, p)), y_true, y_pred)


class AucScore(MetricBase):
    def __init__(self):
        self.name = "auc"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(roc_auc_score, y_true, y_pred)


class LogLossScore(MetricBase):
    def __init__(self):
        self.name = "logloss"

    def calculate(self, y_true, y_pred) -> float:
        clip = lambda p: [max(min(x, 1.0 - 10e-12), 10e-12) for x in p]
        return _mean_over_impressions(lambda l, p: log_loss(l, clip(p)), y_true, y_pred)


class MrrScore(MetricBase):
    def __init__(self):
        self.name = "mrr"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(mrr_score, y_true, y_pred)


class NdcgScore(MetricBase):
    def __init__(self, k: int):
        self.k = k
        self.name = f"ndcg@{k}"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: ndcg_score(l, p, self.k), y_true, y_pred)


def _reject_non_callables(candidates) -> list:
    """The reference's acceptance rule for `metric_functions` (metrics_protocols.py:191-203): a non-empty collection whose
    members can all be called; anything else is a TypeError that lists the types of the offending members."""
    candidates = list(candidates) if not isinstance(candidates, (list, tuple)) else candidates
    offenders = [type(c) for c in candidates if not callable(c)]
    if offenders or len(candidates) == 0:
        raise TypeError(f"Following object(s) are not callable: {offenders}")
    return candidates


class MetricEvaluator:
    """Runs a list of metrics over per-impression labels and predictions (metrics_protocols.py:141-217).

    ``MetricEvaluator(labels, predictions, metric_functions).evaluate()`` stores ``{metric.name: metric(labels, predictions)}``
    in ``.evaluations`` and returns the evaluator itself, so that ``.evaluate().evaluations`` chains.  Assigning to
    ``metric_functions`` -- in the constructor or later -- is checked by `_reject_non_callables`.  Printed, an evaluator shows
    its results as indented JSON (``{}`` before ``evaluate()`` has run)."""

    _HEAD = "<MetricEvaluator class>:"

    def __init__(self, labels, predictions, metric_functions):
        self.labels, self.predictions = labels, predictions
        self.metric_functions = metric_functions
        self.evaluations = {}

    def __setattr__(self, attr, value):
        if attr == "metric_functions":
            value = _reject_non_callables(value)
        object.__setattr__(self, attr, value)

    def evaluate(self):
        results = {}
        for metric in self.metric_functions:
            results[metric.name] = metric(self.labels, self.predictions)
        self.evaluations = results
        return self

    def __repr__(self):
        body = f" \n {json.dumps(self.evaluations, indent=4)}" if self.evaluations else f" {self.evaluations}"
        return self._HEAD + body

    __str__ = __repr__
