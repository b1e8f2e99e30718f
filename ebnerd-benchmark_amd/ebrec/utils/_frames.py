"""Frame plumbing.  The reference moves polars DataFrames around; polars is not a dependency here.
Every util and loader accepts a pandas DataFrame, a pyarrow Table, a polars DataFrame (when polars
is installed) or a dict of columns, and works on pandas internally.  List-valued cells may be python
lists, tuples or numpy arrays; ``None`` stands for a null list (polars null)."""
from __future__ import annotations

import numpy as np
import pandas as pd


def to_pandas(df) -> pd.DataFrame:
    if isinstance(df, pd.DataFrame):
        return df
    if isinstance(df, dict):
        return pd.DataFrame({k: pd.Series(list(v), dtype=object) if _is_nested(v) else v for k, v in df.items()})
    if hasattr(df, "to_pandas"):  # pyarrow.Table, polars.DataFrame / LazyFrame.collect()
        if hasattr(df, "collect"):
            df = df.collect()
        return df.to_pandas()
    raise TypeError(f"cannot interpret {type(df)} as a frame")


def _is_nested(col) -> bool:
    for v in col:
        if v is None:
            continue
        return isinstance(v, (list, tuple, np.ndarray))
    return False


def as_list(cell) -> list:
    """One list-valued cell as a python list; a null cell becomes [None] (polars explodes null -> null)."""
    if cell is None or (isinstance(cell, float) and np.isnan(cell)):
        return [None]
    if isinstance(cell, np.ndarray):
        return cell.tolist()
    return list(cell)


def list_column(df: pd.DataFrame, column: str) -> list[list]:
    return [as_list(c) for c in df[column].tolist()]


def with_column(df: pd.DataFrame, name: str, values) -> pd.DataFrame:
    out = df.copy(deep=False)
    out[name] = pd.Series(list(values), index=df.index, dtype=object)
    return out


def split_df_chunks(df, n_chunks: int) -> list[pd.DataFrame]:
    """n_chunks consecutive row chunks, the remainder rows appended to the last one
    (reference: utils/_polars.py:395-406)."""
    df = to_pandas(df)
    size = len(df) // n_chunks
    chunks = [df.iloc[i * size:(i + 1) * size] for i in range(n_chunks)]
    if len(df) % n_chunks != 0:
        chunks[-1] = pd.concat([chunks[-1], df.iloc[n_chunks * size:]])
    return chunks


def concat_str_columns(df, columns: list[str]):
    """Adds the space-joined string column "<c1>-<c2>-..." and returns (frame, name)
    (reference: utils/_polars.py:569-571)."""
    df = to_pandas(df)
    name = "-".join(columns)
    joined = df[columns[0]].astype(str)
    for c in columns[1:]:
        joined = joined + " " + df[c].astype(str)
    out = df.copy(deep=False)
    out[name] = joined
    return out, name


def slice_join_dataframes(df1, df2, on: str, how: str) -> pd.DataFrame:
    """Join of two frames (reference: utils/_polars.py:68-86; the slicing there is a memory trick)."""
    return to_pandas(df1).merge(to_pandas(df2), on=on, how=how)
