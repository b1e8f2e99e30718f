"""Pairwise AUC without tie credit (reference: evaluation/metrics/_classification.py:4-53)."""
from __future__ import annotations

import numpy as np


def auc_score_custom(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """P(score_pos > score_neg) over all (pos, neg) pairs; ties count as losses, which is where it
    differs from roc_auc_score (_classification.py:41-53)."""
    y_true = np.asarray(y_true).astype(np.bool_)
    y_pred = np.asarray(y_pred)
    pos, neg = y_pred[y_true], y_pred[~y_true]
    return (pos[:, None] > neg[None, :]).sum() / (len(pos) * len(neg))
