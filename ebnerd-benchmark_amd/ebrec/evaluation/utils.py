"""Helpers of the evaluator that the metric wrappers use (reference: evaluation/utils.py:6-10,13-32)."""
from __future__ import annotations

from typing import Iterable

import numpy as np


def convert_to_binary(y_pred: np.ndarray, threshold: float):
    """1 where y_pred >= threshold else 0.  Like the reference (utils.py:6-10) this works IN PLACE on
    an ndarray argument (np.asarray does not copy), so a later metric sees the binarised scores."""
    y_pred = np.asarray(y_pred)
    hit = y_pred >= threshold
    y_pred[hit] = 1
    y_pred[~hit] = 0
    return y_pred


def is_iterable_nested_dtype(iterable: Iterable, dtypes) -> bool:
    return isinstance(iterable[0], dtypes)
