"""Per-impression metric wrappers and the MetricEvaluator (reference: evaluation/metrics_protocols.py).

Each wrapper is the mean over impressions of a 1-D metric -- ``np.mean([metric(l, p) for l, p in zip(labels,
predictions)])`` (metrics_protocols.py:77-86 etc.).  Same names (``auc``, ``mrr``, ``ndcg@k``, ``logloss``, ``rmse``,
``accuracy``, ``f1``), same clipping in LogLossScore (``[10e-12, 1 - 10e-12]``, line 99), same in-place binarisation
side effect of the threshold metrics.  Beyond-accuracy metrics are out of scope.
"""
from __future__ import annotations

import json
from itertools import compress
from typing import Iterable

import numpy as np

from .metrics import accuracy_score, f1_score, log_loss, mean_squared_error, mrr_score, ndcg_score, roc_auc_score
from .protocols import Metric
from .utils import convert_to_binary


def _mean_over_impressions(fn, y_true, y_pred) -> float:
    return float(np.mean([fn(labels, preds) for labels, preds in zip(y_true, y_pred)]))


class AccuracyScore(Metric):
    def __init__(self, threshold: float = 0.5):
        self.threshold = threshold
        self.name = "accuracy"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: accuracy_score(l, convert_to_binary(p, self.threshold)), y_true, y_pred)


class F1Score(Metric):
    def __init__(self, threshold: float = 0.5):
        self.threshold = threshold
        self.name = "f1"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: f1_score(l, convert_to_binary(p, self.threshold)), y_true, y_pred)


class RootMeanSquaredError(Metric):
    def __init__(self):
        self.name = "rmse"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: np.sqrt(mean_squared_error(l, p)), y_true, y_pred)


class AucScore(Metric):
    def __init__(self):
        self.name = "auc"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(roc_auc_score, y_true, y_pred)


class LogLossScore(Metric):
    def __init__(self):
        self.name = "logloss"

    def calculate(self, y_true, y_pred) -> float:
        clip = lambda p: [max(min(x, 1.0 - 10e-12), 10e-12) for x in p]
        return _mean_over_impressions(lambda l, p: log_loss(l, clip(p)), y_true, y_pred)


class MrrScore(Metric):
    def __init__(self):
        self.name = "mrr"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(mrr_score, y_true, y_pred)


class NdcgScore(Metric):
    def __init__(self, k: int):
        self.k = k
        self.name = f"ndcg@{k}"

    def calculate(self, y_true, y_pred) -> float:
        return _mean_over_impressions(lambda l, p: ndcg_score(l, p, self.k), y_true, y_pred)


class MetricEvaluator:
    """``MetricEvaluator(labels, predictions, metric_functions).evaluate()`` fills ``.evaluations`` and returns
    the evaluator itself (metrics_protocols.py:184-189)."""

    def __init__(self, labels, predictions, metric_functions):
        self.labels = labels
        self.predictions = predictions
        self.metric_functions = metric_functions
        self.evaluations = dict()

    def evaluate(self):
        self.evaluations = {m.name: m(self.labels, self.predictions) for m in self.metric_functions}
        return self

    @property
    def metric_functions(self):
        return self.__metric_functions

    @metric_functions.setter
    def metric_functions(self, values):
        invalid = [not callable(item) for item in values]
        if not any(invalid) and invalid:
            self.__metric_functions = values
        else:
            raise TypeError(f"Following object(s) are not callable: {[type(i) for i in compress(values, invalid)]}")

    def __str__(self):
        if self.evaluations:
            return f"<MetricEvaluator class>: \n {json.dumps(self.evaluations, indent=4)}"
        return f"<MetricEvaluator class>: {self.evaluations}"

    def __repr__(self):
        return str(self)
