"""The five sklearn metrics the reference imports (evaluation/metrics/_sklearn.py:1-14), restated in
numpy for 1-D binary inputs so the evaluator has no sklearn dependency on the GPU box.  Pinned
against sklearn through the reference-generated golden vectors (tests/golden/metrics_golden.json)."""
from __future__ import annotations

import numpy as np


def _rankdata_average(a: np.ndarray) -> np.ndarray:
    order = np.argsort(a, kind="mergesort")
    s = a[order]
    boundary = np.concatenate(([True], s[1:] != s[:-1]))
    starts = np.flatnonzero(boundary)
    counts = np.diff(np.append(starts, len(a)))
    avg = starts + (counts + 1) / 2.0  # average 1-based rank of each tie group
    ranks = np.empty(len(a), dtype=np.float64)
    ranks[order] = np.repeat(avg, counts)
    return ranks


def roc_auc_score(y_true, y_score) -> float:
    """Area under the ROC curve = Mann-Whitney U with average ranks for ties."""
    y = np.asarray(y_true).reshape(-1)
    s = np.asarray(y_score, dtype=np.float64).reshape(-1)
    classes = np.unique(y)
    if len(classes) != 2:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    pos = y == classes[1]
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    ranks = _rankdata_average(s)
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def mean_squared_error(y_true, y_pred) -> float:
    d = np.asarray(y_true, dtype=np.float64) - np.asarray(y_pred, dtype=np.float64)
    return float(np.mean(d * d))


def accuracy_score(y_true, y_pred) -> float:
    return float(np.mean(np.asarray(y_true) == np.asarray(y_pred)))


def f1_score(y_true, y_pred) -> float:
    """Binary F1 for the positive label 1; 0.0 when there is nothing to score (sklearn zero_division)."""
    y, p = np.asarray(y_true) == 1, np.asarray(y_pred) == 1
    tp = float(np.sum(y & p))
    denom = 2 * tp + float(np.sum(~y & p)) + float(np.sum(y & ~p))
    return 0.0 if denom == 0 else 2 * tp / denom


def log_loss(y_true, y_pred) -> float:
    """Binary cross-entropy of P(label == larger class); y_true must contain both labels."""
    y = np.asarray(y_true).reshape(-1)
    p = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    classes = np.unique(y)
    if len(classes) < 2:
        raise ValueError(f"y_true contains only one label ({classes[0]}). Please provide the true labels explicitly through the labels argument.")
    t = (y == classes[1]).astype(np.float64)
    eps = np.finfo(np.float64).eps
    p = np.clip(p, eps, 1 - eps)
    return float(-np.mean(t * np.log(p) + (1 - t) * np.log(1 - p)))
