"""Batch producers of the NRMS / NRMSDocVec path: same constructor arguments and the same
``len(loader)`` / ``loader[idx] -> ((his, pred), y)`` protocol as the reference's
``NRMSDataLoader`` / ``NRMSDataLoaderPretransform`` (dataloader.py:19-180), without polars or
TensorFlow (a ``tf.keras.utils.Sequence`` is only "something with __len__ and __getitem__").

Shapes (dataloader.py:83-119):
  train mode   his (B, H, T)        pred (B, C, T)        y (B, C)          -- every in-view list has C ids
  eval mode    his (sum C_i, H, T)  pred (sum C_i, 1, T)  y (sum C_i, 1)    -- history repeated per candidate
with T = len(article_dict value): token ids (int64) for NRMS, a document vector (float) for DocVec.
Unknown / null / padded article ids map to row 0 of the lookup matrix (dataloader.py:43; Appendix B).

Article ids are mapped to lookup-matrix rows ONCE in ``__post_init__`` into padded int32 index arrays
(both loader classes: the per-batch polars transform of ``NRMSDataLoader`` (dataloader.py:68-81) exists
for memory reasons that do not apply to an index array), so ``__getitem__`` is two numpy gathers.
``compact_eval_batch`` additionally hands the scorer un-repeated histories (§8f row 1).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ebrec.utils._constants import DEFAULT_INVIEW_ARTICLES_COL, DEFAULT_LABELS_COL, DEFAULT_USER_COL
from ebrec.utils._frames import list_column, to_pandas
from ebrec.utils._python import create_lookup_objects, repeat_by_list_values_from_matrix


def _map_ids(cells, mapping: dict) -> tuple[np.ndarray, np.ndarray]:
    """Ragged article ids (the cells of a list-valued column: lists, tuples, numpy arrays; None = a null list) -> (flat lookup
    rows int32, offsets); unknown / null ids -> row 0.  A null cell counts as ONE null id (polars explodes null -> null), like
    ``as_list``.  Integer columns take a vectorised path (one concatenate + one binary search over the sorted article ids:
    the reference's per-row polars transform and ``to_list()`` were where its loader time went); anything
    else (None inside a list, mixed types) falls back to the per-element dictionary lookup."""
    cells = list(cells)
    lens = np.fromiter((1 if c is None or isinstance(c, float) else len(c) for c in cells), dtype=np.int64, count=len(cells))
    offsets = np.concatenate(([0], np.cumsum(lens)))
    total = int(offsets[-1])
    if total and mapping:
        try:
            null = np.array([np.iinfo(np.int64).min], dtype=np.int64)
            flat_ids = np.concatenate([null if (c is None or isinstance(c, float)) else np.asarray(c) for c in cells])
        except ValueError:
            flat_ids = None
        if flat_ids is not None and flat_ids.dtype.kind in "iu" and flat_ids.size == total:
            keys = np.fromiter(mapping.keys(), dtype=np.int64, count=len(mapping))
            rows = np.fromiter(mapping.values(), dtype=np.int32, count=len(mapping))
            order = np.argsort(keys, kind="stable")
            keys, rows = keys[order], rows[order]
            flat_ids = flat_ids.astype(np.int64, copy=False)
            pos = np.minimum(np.searchsorted(keys, flat_ids), len(keys) - 1)
            return np.where(keys[pos] == flat_ids, rows[pos], 0).astype(np.int32), offsets
    get = mapping.get
    flat = np.fromiter((get(a, 0) for c in cells for a in ([None] if (c is None or isinstance(c, float)) else
                                                          (c.tolist() if isinstance(c, np.ndarray) else c))),
                       dtype=np.int32, count=total)
    return flat, offsets


@dataclass
class NewsrecDataLoader:
    """Base loader (reference dataloader.py:19-63)."""

    behaviors: object
    history_column: str
    article_dict: dict
    unknown_representation: str
    eval_mode: bool = False
    batch_size: int = 32
    inview_col: str = DEFAULT_INVIEW_ARTICLES_COL
    labels_col: str = DEFAULT_LABELS_COL
    user_col: str = DEFAULT_USER_COL
    kwargs: dict = None

    def __post_init__(self):
        self.lookup_article_index, self.lookup_article_matrix = create_lookup_objects(
            self.article_dict, unknown_representation=self.unknown_representation)
        self.unknown_index = [0]
        self.X, self.y = self.load_data()
        if self.kwargs is not None:
            self.set_kwargs(self.kwargs)

    def __len__(self) -> int:
        return int(np.ceil(len(self.X) / float(self.batch_size)))

    def __getitem__(self, idx):
        raise ValueError("Function '__getitem__' needs to be implemented.")

    def load_data(self):
        """X = behaviors without the labels column plus ``n_samples`` (in-view length); y = the labels."""
        df = to_pandas(self.behaviors)
        y = list_column(df, self.labels_col)
        X = df.drop(columns=[self.labels_col]).reset_index(drop=True)
        X["n_samples"] = [len(l) for l in list_column(X, self.inview_col)]
        return X, y

    def set_kwargs(self, kwargs: dict):
        for key, value in kwargs.items():
            setattr(self, key, value)


@dataclass
class NRMSDataLoader(NewsrecDataLoader):
    """reference dataloader.py:66-119."""

    def __post_init__(self):
        super().__post_init__()
        self._his_flat, self._his_off = _map_ids(self.X[self.history_column].tolist(), self.lookup_article_index)
        self._inv_flat, self._inv_off = _map_ids(self.X[self.inview_col].tolist(), self.lookup_article_index)
        self._y_flat = np.concatenate([np.asarray(l, dtype=np.int64) for l in self.y]) if len(self.y) else np.zeros(0, np.int64)
        hl = np.diff(self._his_off)
        if len(hl) and hl.min() != hl.max():
            raise ValueError("history lists must all have the same length (truncate_history with a padding value)")
        self._H = int(hl[0]) if len(hl) else 0

    # ---- ragged helpers --------------------------------------------------------------------
    def _rows(self, idx):
        lo = idx * self.batch_size
        return lo, min(lo + self.batch_size, len(self.X))

    def _history_rows(self, lo, hi) -> np.ndarray:
        return self._his_flat[self._his_off[lo]: self._his_off[hi]].reshape(hi - lo, self._H)

    def __getitem__(self, idx):
        """his_input_title (samples, history_size, T), pred_input_title (samples, npratio+1, T), batch_y."""
        lo, hi = self._rows(idx)
        his_idx = self._history_rows(lo, hi)
        inv = self._inv_flat[self._inv_off[lo]: self._inv_off[hi]]
        ylab = self._y_flat[self._inv_off[lo]: self._inv_off[hi]]
        if self.eval_mode:
            repeats = np.diff(self._inv_off[lo: hi + 1])
            batch_y = ylab.reshape(-1, 1)
            his_input_title = repeat_by_list_values_from_matrix(his_idx, matrix=self.lookup_article_matrix, repeats=repeats)
            pred_input_title = self.lookup_article_matrix[inv][:, None, :]
        else:
            lens = np.diff(self._inv_off[lo: hi + 1])
            if len(lens) and lens.min() != lens.max():
                raise ValueError("train mode needs equal-length in-view lists (sampling_strategy_wu2019); use eval_mode=True")
            C = int(lens[0]) if len(lens) else 0
            batch_y = ylab.reshape(hi - lo, C)
            his_input_title = self.lookup_article_matrix[his_idx]
            pred_input_title = self.lookup_article_matrix[inv.reshape(hi - lo, C)]
        return (his_input_title, pred_input_title), batch_y

    def index_batch(self, idx):
        """Train batch as article-row numbers of ``lookup_article_matrix`` -- ((his (B,H), pred (B,C)) int32, y (B,C)):
        what a model that keeps the matrix in HBM needs per step (device-side batch assembly)."""
        lo, hi = self._rows(idx)
        lens = np.diff(self._inv_off[lo: hi + 1])
        if len(lens) and lens.min() != lens.max():
            raise ValueError("train mode needs equal-length in-view lists (sampling_strategy_wu2019)")
        C = int(lens[0]) if len(lens) else 0
        sl = slice(self._inv_off[lo], self._inv_off[hi])
        return (self._history_rows(lo, hi), self._inv_flat[sl].reshape(hi - lo, C)), self._y_flat[sl].reshape(hi - lo, C)

    def index_eval_batch(self, idx):
        """Eval batch as article-row numbers of ``lookup_article_matrix``: (his (b,H) int32, cand (sum C_i,) int32,
        impression_of_row (sum C_i,) int32, y (sum C_i, 1)) -- what a scorer that has encoded every article of the
        matrix ONCE needs per batch."""
        lo, hi = self._rows(idx)
        sl = slice(self._inv_off[lo], self._inv_off[hi])
        repeats = np.diff(self._inv_off[lo: hi + 1])
        rows = np.repeat(np.arange(hi - lo, dtype=np.int32), repeats)
        return self._history_rows(lo, hi), self._inv_flat[sl], rows, self._y_flat[sl].reshape(-1, 1)

    def compact_eval_batch(self, idx):
        """Eval batch WITHOUT the per-candidate repetition of the history: (his (b,H,T), pred (sum C_i, T),
        impression_of_row (sum C_i,), y (sum C_i, 1)).  Scores are identical to the repeated layout."""
        lo, hi = self._rows(idx)
        inv = self._inv_flat[self._inv_off[lo]: self._inv_off[hi]]
        repeats = np.diff(self._inv_off[lo: hi + 1])
        rows = np.repeat(np.arange(hi - lo, dtype=np.int32), repeats)
        y = self._y_flat[self._inv_off[lo]: self._inv_off[hi]].reshape(-1, 1)
        return self.lookup_article_matrix[self._history_rows(lo, hi)], self.lookup_article_matrix[inv], rows, y


@dataclass
class NRMSDataLoaderPretransform(NRMSDataLoader):
    """reference dataloader.py:122-180: the whole frame is mapped up front (which NRMSDataLoader here
    does as well -- the class is kept so ``--nrms_loader NRMSDataLoaderPretransform`` keeps working)."""
