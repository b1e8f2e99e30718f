"""Per-impression ranking metrics (reference: evaluation/metrics/_ranking.py:4-155): same argsort-
descending tie handling (``np.argsort(y_pred)[::-1]``), same formulas."""
from __future__ import annotations

import numpy as np


def _order_desc(y_pred) -> np.ndarray:
    return np.argsort(y_pred)[::-1]


def reciprocal_rank_score(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """1 / rank of the first positive (_ranking.py:49-52)."""
    ranked = np.take(y_true, _order_desc(y_pred))
    return 1.0 / (np.argmax(ranked) + 1)


def dcg_score(y_true: np.ndarray, y_pred: np.ndarray, k: int = 10) -> float:
    """sum_{i<k} (2^rel_i - 1) / log2(i + 2) over the top-k by score (_ranking.py:84-89)."""
    k = min(np.shape(y_true)[-1], k)
    ranked = np.take(y_true, _order_desc(y_pred)[:k])
    gains = 2 ** ranked - 1
    discounts = np.log2(np.arange(len(ranked)) + 2)
    return np.sum(gains / discounts)


def ndcg_score(y_true: np.ndarray, y_pred: np.ndarray, k: int = 10) -> float:
    """DCG normalised by the ideal DCG (_ranking.py:121-123)."""
    return dcg_score(y_true, y_pred, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """sum_i rel_i / rank_i divided by the number of positives (_ranking.py:152-155)."""
    ranked = np.take(y_true, _order_desc(y_pred))
    return np.sum(ranked / (np.arange(len(ranked)) + 1)) / np.sum(ranked)
