"""Article-side helpers (reference: utils/_articles.py:21-79)."""
from __future__ import annotations

from ._constants import DEFAULT_ARTICLE_ID_COL
from ._frames import to_pandas


def create_article_id_to_value_mapping(df, value_col: str, article_col: str = DEFAULT_ARTICLE_ID_COL) -> dict:
    """{article_id: value} in frame order (reference _articles.py:21-28)."""
    df = to_pandas(df)
    values = [v.tolist() if hasattr(v, "tolist") else v for v in df[value_col].tolist()]
    return dict(zip(df[article_col].tolist(), values))


def convert_text2encoding_with_transformers(df, tokenizer, column: str, max_length: int = None):
    """Adds "<column>_encode_<tokenizer name>" with token ids: add_special_tokens=False, and when
    max_length is given padding="max_length" + truncation (reference _articles.py:31-79).
    Returns (frame, new column name)."""
    df = to_pandas(df)
    new_col = f"{column}_encode_{tokenizer.name_or_path}"
    kwargs = dict(add_special_tokens=False)
    if max_length is not None:
        kwargs.update(padding="max_length", truncation=True, max_length=max_length)
    ids = tokenizer(df[column].astype(str).tolist(), **kwargs)["input_ids"]
    out = df.copy(deep=False)
    out[new_col] = list(ids)
    return out, new_col
