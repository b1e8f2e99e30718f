"""Behaviour-frame pipeline of the NRMS drivers (reference: utils/_behaviors.py), on pandas.

Same function names, arguments and row semantics as the reference; random choices use a seeded
``numpy.random.Generator`` instead of polars' sampler, so sampled negatives / shuffles agree in
distribution, not element by element.
"""
from __future__ import annotations

import warnings
from pathlib import Path

import numpy as np
import pandas as pd

from ._constants import (DEFAULT_CLICKED_ARTICLES_COL, DEFAULT_HISTORY_ARTICLE_ID_COL, DEFAULT_INVIEW_ARTICLES_COL,
                         DEFAULT_LABELS_COL, DEFAULT_USER_COL)
from ._frames import list_column, to_pandas, with_column


def _check_columns_in_df(df: pd.DataFrame, columns: list[str]) -> None:
    missing = [c for c in columns if c not in df.columns]
    if missing:
        raise ValueError(f"Invalid input provided. The dataframe does not contain columns {missing}.")


def shuffle_list_column(df, column: str, seed: int = None) -> pd.DataFrame:
    """Shuffles every list of `column` independently (reference _behaviors.py:110-158)."""
    df = to_pandas(df)
    rng = np.random.default_rng(seed)
    return with_column(df, column, [list(rng.permutation(np.array(l, dtype=object))) for l in list_column(df, column)])


def create_binary_labels_column(df, shuffle: bool = True, seed: int = None, clicked_col: str = DEFAULT_CLICKED_ARTICLES_COL,
                                inview_col: str = DEFAULT_INVIEW_ARTICLES_COL, label_col: str = DEFAULT_LABELS_COL) -> pd.DataFrame:
    """labels[i] = 1 if inview[i] is one of the clicked ids else 0 (int8), after an optional shuffle
    of the in-view list; a null clicked list gives all zeros (reference _behaviors.py:22-107)."""
    df = to_pandas(df)
    _check_columns_in_df(df, [inview_col, clicked_col])
    cols = list(df.columns)
    if shuffle:
        df = shuffle_list_column(df, inview_col, seed)
    labels = []
    for inview, clicked in zip(list_column(df, inview_col), list_column(df, clicked_col)):
        cs = set(clicked) - {None}
        labels.append([int(a in cs) for a in inview])
    out = with_column(df, label_col, [np.asarray(l, dtype=np.int8).tolist() for l in labels])
    return out[cols + [label_col]] if label_col not in cols else out[cols]


def truncate_history(df, column: str, history_size: int, padding_value=None, enable_warning: bool = True) -> pd.DataFrame:
    """Keeps the LAST history_size entries; with a padding value, shorter lists are LEFT-padded to exactly
    history_size (reference _behaviors.py:582-654)."""
    df = to_pandas(df)
    if enable_warning:
        warnings.warn("truncate_history: The history IDs expeced in ascending order")
    out = []
    for l in list_column(df, column):
        if padding_value is not None and len(l) < history_size:
            l = [padding_value] * (history_size - len(l)) + l
        out.append(l[-history_size:])
    return with_column(df, column, out)


def ebnerd_from_path(path, history_size: int = 30, padding: int = 0, user_col: str = DEFAULT_USER_COL,
                     history_aids_col: str = DEFAULT_HISTORY_ARTICLE_ID_COL) -> pd.DataFrame:
    """behaviors.parquet left-joined with the truncated/padded history of each user
    (reference _behaviors.py:161-192)."""
    path = Path(path)
    hist = pd.read_parquet(path / "history.parquet", columns=[user_col, history_aids_col])
    hist = truncate_history(hist, history_aids_col, history_size, padding_value=padding, enable_warning=False)
    return pd.read_parquet(path / "behaviors.parquet").merge(hist, on=user_col, how="left")


def remove_positives_from_inview(df, inview_col: str = DEFAULT_INVIEW_ARTICLES_COL,
                                 clicked_col: str = DEFAULT_CLICKED_ARTICLES_COL) -> pd.DataFrame:
    """In-view lists without the clicked ids (reference _behaviors.py:371-420)."""
    df = to_pandas(df)
    _check_columns_in_df(df, [inview_col, clicked_col])
    neg = [[a for a in inview if a not in clicked] for inview, clicked in
           zip(list_column(df, inview_col), list_column(df, clicked_col))]
    return with_column(df, inview_col, neg)


def sample_article_ids(df, n: int, with_replacement: bool = False, seed: int = None,
                       inview_col: str = DEFAULT_INVIEW_ARTICLES_COL) -> pd.DataFrame:
    """n ids drawn from each in-view list; an empty list yields n nulls (reference _behaviors.py:275-368)."""
    df = to_pandas(df)
    _check_columns_in_df(df, [inview_col])
    rng = np.random.default_rng(seed)
    out = []
    for l in list_column(df, inview_col):
        if len(l) == 0:
            out.append([None] * n)
            continue
        if not with_replacement and len(l) < n:
            raise ValueError(f"cannot take a larger sample ({n}) than population ({len(l)}) when 'with_replacement=False'")
        out.append([l[i] for i in rng.choice(len(l), size=n, replace=with_replacement)])
    return with_column(df, inview_col, out)


def sampling_strategy_wu2019(df, npratio: int, shuffle: bool = False, with_replacement: bool = True, seed: int = None,
                             inview_col: str = DEFAULT_INVIEW_ARTICLES_COL,
                             clicked_col: str = DEFAULT_CLICKED_ARTICLES_COL) -> pd.DataFrame:
    """Wu et al. 2019 negative sampling: one row per clicked article, `npratio` negatives sampled from
    the non-clicked in-view ids, the positive appended, optionally shuffled; clicked column becomes
    [positive] (reference _behaviors.py:423-579)."""
    df = remove_positives_from_inview(to_pandas(df), inview_col, clicked_col)
    df = df.explode(clicked_col, ignore_index=True)
    df = sample_article_ids(df, n=npratio, with_replacement=with_replacement, seed=seed, inview_col=inview_col)
    pos = df[clicked_col].tolist()
    df = with_column(df, inview_col, [l + [p] for l, p in zip(list_column(df, inview_col), pos)])
    df = with_column(df, clicked_col, [[p] for p in pos])
    if shuffle:
        df = shuffle_list_column(df, inview_col, seed)
    return df


def add_prediction_scores(df, scores, prediction_scores_col: str = "scores",
                          inview_col: str = DEFAULT_INVIEW_ARTICLES_COL) -> pd.DataFrame:
    """Re-nests flat (or already nested) scores by the in-view lengths, in row order
    (reference _behaviors.py:1024-1089)."""
    df = to_pandas(df)
    flat = []
    for s in scores:
        if isinstance(s, (list, tuple, np.ndarray)):
            flat.extend(np.asarray(s).reshape(-1).tolist())
        else:
            flat.append(s)
    lens = [len(l) for l in list_column(df, inview_col)]
    if sum(lens) != len(flat):
        raise ValueError(f"got {len(flat)} scores for {sum(lens)} in-view articles")
    out, pos = [], 0
    for n in lens:
        out.append(flat[pos:pos + n])
        pos += n
    return with_column(df, prediction_scores_col, out)
