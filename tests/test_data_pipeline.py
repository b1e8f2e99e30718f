"""Host data path: utils + loaders against the reference's documented
behaviour (docstring known answers) and against the reference's own parquet test fixtures
(tests/golden/ebnerd/*.parquet = /root/reference/test/data/ebnerd, data files only).  No GPU needed."""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from ebrec.models.newsrec.dataloader import NRMSDataLoader, NRMSDataLoaderPretransform, NewsrecDataLoader
from ebrec.utils._articles import create_article_id_to_value_mapping
from ebrec.utils._behaviors import (add_prediction_scores, create_binary_labels_column, ebnerd_from_path,
                                    remove_positives_from_inview, sample_article_ids, sampling_strategy_wu2019,
                                    truncate_history)
from ebrec.utils._constants import (DEFAULT_ARTICLE_ID_COL, DEFAULT_CLICKED_ARTICLES_COL, DEFAULT_HISTORY_ARTICLE_ID_COL,
                                    DEFAULT_INVIEW_ARTICLES_COL, DEFAULT_LABELS_COL, DEFAULT_USER_COL)
from ebrec.utils._frames import concat_str_columns, list_column, split_df_chunks
from ebrec.utils._python import (create_lookup_objects, rank_predictions_by_score, read_submission_file,
                                 repeat_by_list_values_from_matrix, write_submission_file)

DATA = Path(__file__).parent / "golden" / "ebnerd"


# ------------------------------------------------------------------ docstring known answers
def test_create_lookup_objects_docstring():  # _python.py:440-465
    data = {10: np.array([0.1, 0.2, 0.3]), 20: np.array([0.4, 0.5, 0.6]), 30: np.array([0.7, 0.8, 0.9])}
    idx, mat = create_lookup_objects(data, "zeros")
    assert idx == {10: 1, 20: 2, 30: 3}
    np.testing.assert_array_equal(mat, [[0, 0, 0], [0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]])
    _, mat = create_lookup_objects(data, "mean")
    np.testing.assert_allclose(mat[0], [0.4, 0.5, 0.6])
    with pytest.raises(ValueError, match="not a specified method"):
        create_lookup_objects(data, "median")
    _, tok = create_lookup_objects({7: [3, 4], 9: [5, 6]}, "zeros")
    assert tok.dtype == np.int64 and tok[0].tolist() == [0, 0]


def test_repeat_by_list_values_from_matrix_docstring():  # _python.py:376-387
    out = repeat_by_list_values_from_matrix(np.array([[1, 0], [0, 0]]), np.array([[7, 8, 9], [10, 11, 12]]), np.array([1, 2]))
    np.testing.assert_array_equal(out, [[[10, 11, 12], [7, 8, 9]], [[7, 8, 9], [7, 8, 9]], [[7, 8, 9], [7, 8, 9]]])


def test_rank_predictions_docstring():  # _python.py:51-57
    rows = [[0.2, 0.1, 0.3], [0.1, 0.2], [0.4, 0.2, 0.1, 0.3]]
    assert [rank_predictions_by_score(r).tolist() for r in rows] == [[2, 3, 1], [2, 1], [1, 3, 4, 2]]


def test_submission_file_roundtrip(tmp_path):  # _python.py:76-82, 95-101
    ids, scores = [237, 291, 320], [[0.2, 0.1, 0.3], [0.1, 0.2], [0.4, 0.2, 0.1, 0.3]]
    p = tmp_path / "predictions.txt"
    write_submission_file(ids, scores, path=p, rm_file=False)
    assert p.read_text().splitlines() == ["237 [0.2,0.1,0.3]", "291 [0.1,0.2]", "320 [0.4,0.2,0.1,0.3]"]
    assert (tmp_path / "predictions.zip").exists()
    assert read_submission_file(p) == (ids, scores)
    with pytest.raises(ValueError):
        write_submission_file(ids, scores, path=p, filename_zip="x.txt")


def test_create_binary_labels_docstring():  # _behaviors.py:46-84
    df = pd.DataFrame({DEFAULT_INVIEW_ARTICLES_COL: [[1, 2, 3], [4, 5, 6], [7, 8]],
                       DEFAULT_CLICKED_ARTICLES_COL: [[2, 3, 4], [3, 5], None]})
    assert create_binary_labels_column(df, shuffle=False)[DEFAULT_LABELS_COL].tolist() == [[0, 1, 1], [0, 1, 0], [0, 0]]
    sh = create_binary_labels_column(df, shuffle=True, seed=123)
    assert [sum(l) for l in sh[DEFAULT_LABELS_COL]] == [2, 1, 0]
    for inv, lab, clicked in zip(sh[DEFAULT_INVIEW_ARTICLES_COL], sh[DEFAULT_LABELS_COL], [[2, 3, 4], [3, 5], []]):
        assert [int(a in clicked) for a in inv] == lab
    with pytest.raises(ValueError):
        create_binary_labels_column(df.drop(columns=[DEFAULT_CLICKED_ARTICLES_COL]))


def test_truncate_history_docstring():  # _behaviors.py:607-642
    df = pd.DataFrame({"id": [1, 2, 3], "history": [["a", "b", "c"], ["d", "e", "f", "g"], ["h", "i"]]})
    assert truncate_history(df, "history", 3, enable_warning=False)["history"].tolist() == [["a", "b", "c"], ["e", "f", "g"], ["h", "i"]]
    assert truncate_history(df, "history", 3, "-", enable_warning=False)["history"].tolist() == [["a", "b", "c"], ["e", "f", "g"], ["-", "h", "i"]]
    with pytest.warns(UserWarning):
        truncate_history(df, "history", 2)


def test_sampling_strategy_wu2019_semantics():  # _behaviors.py:423-579
    df = pd.DataFrame({"impression_id": [1, 2, 3], DEFAULT_USER_COL: [1, 1, 2],
                       DEFAULT_INVIEW_ARTICLES_COL: [[1, 2, 3, 4], [5, 6], [7, 8, 9]],
                       DEFAULT_CLICKED_ARTICLES_COL: [[1, 3], [5, 6], [9]]})
    assert remove_positives_from_inview(df)[DEFAULT_INVIEW_ARTICLES_COL].tolist() == [[2, 4], [], [7, 8]]
    out = sampling_strategy_wu2019(df, npratio=2, shuffle=False, with_replacement=False, seed=1)
    assert out["impression_id"].tolist() == [1, 1, 2, 2, 3]  # one row per clicked article
    assert out[DEFAULT_CLICKED_ARTICLES_COL].tolist() == [[1], [3], [5], [6], [9]]
    inv = out[DEFAULT_INVIEW_ARTICLES_COL].tolist()
    assert all(len(l) == 3 for l in inv)  # npratio negatives + the positive, positive LAST when not shuffled
    assert [l[-1] for l in inv] == [1, 3, 5, 6, 9]
    assert sorted(inv[0][:2]) == [2, 4] and sorted(inv[4][:2]) == [7, 8]
    assert inv[2][:2] == [None, None]  # impression without negatives -> nulls -> unknown article row 0
    sh = sampling_strategy_wu2019(df, npratio=4, shuffle=True, with_replacement=True, seed=7)
    lab = create_binary_labels_column(sh, shuffle=False)[DEFAULT_LABELS_COL].tolist()
    assert all(len(l) == 5 and sum(l) == 1 for l in lab)
    with pytest.raises(ValueError):
        sample_article_ids(df, n=5, with_replacement=False)


def test_add_prediction_scores_docstring():  # _behaviors.py:1043-1063
    df = pd.DataFrame({"id": [1, 2], DEFAULT_INVIEW_ARTICLES_COL: [[1, 2, 3], [4, 5]]})
    out = add_prediction_scores(df, [[0.3], [0.4], [0.5], [0.6], [0.7]], prediction_scores_col="p")
    assert out["p"].tolist() == [[0.3, 0.4, 0.5], [0.6, 0.7]]
    out = add_prediction_scores(df, np.array([0.3, 0.4, 0.5, 0.6, 0.7]).reshape(-1, 1))
    assert out["scores"].tolist() == [[0.3, 0.4, 0.5], [0.6, 0.7]]
    with pytest.raises(ValueError):
        add_prediction_scores(df, [0.1, 0.2])


def test_split_chunks_and_concat_str():  # _polars.py:395-406, 560-571
    df = pd.DataFrame({"a": range(11), "first": list("abcdefghijk"), "last": list("ABCDEFGHIJK")})
    chunks = split_df_chunks(df, 3)
    assert [len(c) for c in chunks] == [3, 3, 5] and pd.concat(chunks)["a"].tolist() == list(range(11))
    out, name = concat_str_columns(df, ["first", "last"])
    assert name == "first-last" and out[name].tolist()[:2] == ["a A", "b B"]


# ------------------------------------------------------------------ the reference's own fixtures
@pytest.fixture(scope="module")
def frames():
    """Mirror of the module-level setup of /root/reference/test/dataloader/test_newsrec.py:31-63."""
    rng = np.random.default_rng(0)
    hist = pd.read_parquet(DATA / "history.parquet", columns=[DEFAULT_USER_COL, DEFAULT_HISTORY_ARTICLE_ID_COL])
    hist = truncate_history(hist, DEFAULT_HISTORY_ARTICLE_ID_COL, 3, enable_warning=False)  # list.tail(3)
    beh = pd.read_parquet(DATA / "behaviors.parquet", columns=[DEFAULT_USER_COL, DEFAULT_INVIEW_ARTICLES_COL, DEFAULT_CLICKED_ARTICLES_COL])
    beh["n"] = [len(l) for l in beh[DEFAULT_INVIEW_ARTICLES_COL]]
    beh = beh.merge(hist, on=DEFAULT_USER_COL, how="left")
    beh = beh[beh[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()].reset_index(drop=True)
    beh = create_binary_labels_column(beh, shuffle=True, seed=0)
    # articles.parquet is missing from the reference checkout (.MISSING_LARGE_BLOBS): synthesise ids + tokens
    ids = sorted({a for l in list_column(beh, DEFAULT_INVIEW_ARTICLES_COL) for a in l} |
                 {a for l in list_column(beh, DEFAULT_HISTORY_ARTICLE_ID_COL) for a in l})
    ids = ids[: len(ids) * 9 // 10]  # leave 10 % of the ids unknown
    articles = pd.DataFrame({DEFAULT_ARTICLE_ID_COL: ids, "tokens": rng.integers(1, 20, (len(ids), 10)).tolist()})
    mapping = create_article_id_to_value_mapping(articles, value_col="tokens")
    train = beh[beh["n"] == beh["n"].min()].reset_index(drop=True)
    return beh, train, mapping


def test_reference_fixture_shape(frames):
    beh, train, mapping = frames
    raw = pd.read_parquet(DATA / "behaviors.parquet")
    assert len(raw) == 1046 and raw[DEFAULT_USER_COL].nunique() == 38
    assert len(pd.read_parquet(DATA / "history.parquet")) == 44
    assert len(train) > 0 and len(next(iter(mapping.values()))) == 10


@pytest.mark.parametrize("cls", [NRMSDataLoader, NRMSDataLoaderPretransform])
def test_nrms_loader_train_mode_like_reference_test(frames, cls):
    """test_newsrec.py:66-91: len, tuple structure, integer tokens, integer labels."""
    beh, train, mapping = frames
    loader = cls(behaviors=train, article_dict=mapping, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                 unknown_representation="zeros", eval_mode=False, batch_size=100)
    assert len(loader) == int(np.ceil(len(train) / 100))
    batch = loader[0]
    assert isinstance(batch, tuple) and len(batch) == 2 and len(batch[0]) == 2
    (his, pred), y = batch
    n = min(100, len(train))
    C = int(train["n"].min())
    assert his.shape == (n, 3, 10) and pred.shape == (n, C, 10) and y.shape == (n, C)
    assert np.issubdtype(his.dtype, np.integer) and np.issubdtype(pred.dtype, np.integer) and np.issubdtype(y.dtype, np.integer)
    # content: row 0 of the batch is the token rows of that impression's articles, unknown ids -> zeros
    for c, aid in enumerate(list_column(train, DEFAULT_INVIEW_ARTICLES_COL)[0]):
        assert pred[0, c].tolist() == mapping.get(aid, [0] * 10)
    for h, aid in enumerate(list_column(train, DEFAULT_HISTORY_ARTICLE_ID_COL)[0]):
        assert his[0, h].tolist() == mapping.get(aid, [0] * 10)
    assert y[0].tolist() == list_column(train, DEFAULT_LABELS_COL)[0]
    last = loader[len(loader) - 1]
    assert len(last[1]) == len(train) - 100 * (len(loader) - 1)


def test_nrms_loader_eval_mode_like_reference_test(frames):
    """test_newsrec.py:93-105: the number of labels of batch 0 == sum of the first 100 in-view lengths."""
    beh, train, mapping = frames
    loader = NRMSDataLoader(behaviors=beh, article_dict=mapping, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                            unknown_representation="zeros", eval_mode=True, batch_size=100)
    (his, pred), y = loader[0]
    want = int(beh["n"].iloc[:100].sum())
    assert y.shape == (want, 1) and his.shape == (want, 3, 10) and pred.shape == (want, 1, 10)
    # history is repeated once per candidate, in row order (dataloader.py:99-103)
    n0 = int(beh["n"].iloc[0])
    assert (his[:n0] == his[0]).all()
    hc, pc, rows, yc = loader.compact_eval_batch(0)
    assert hc.shape == (100, 3, 10) and pc.shape == (want, 10) and rows.shape == (want,)
    np.testing.assert_array_equal(hc[rows], his)
    np.testing.assert_array_equal(pc, pred[:, 0])
    np.testing.assert_array_equal(yc, y)
    total = sum(len(loader[i][1]) for i in range(len(loader)))
    assert total == int(beh["n"].sum())
    # the same batch as article-row numbers of the lookup matrix (what the article-caching scorer consumes)
    hi, ci, rows_i, yi = loader.index_eval_batch(0)
    np.testing.assert_array_equal(loader.lookup_article_matrix[hi], hc)
    np.testing.assert_array_equal(loader.lookup_article_matrix[ci], pc)
    np.testing.assert_array_equal(rows_i, rows)
    np.testing.assert_array_equal(yi, yc)


def test_loader_errors_and_unknown_mean(frames):
    beh, train, mapping = frames
    with pytest.raises(ValueError, match="__getitem__"):
        NewsrecDataLoader(behaviors=train, article_dict=mapping, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                          unknown_representation="zeros")[0]
    with pytest.raises(ValueError):
        NRMSDataLoader(behaviors=train, article_dict=mapping, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                       unknown_representation="nope")
    with pytest.raises(ValueError, match="equal-length"):
        NRMSDataLoader(behaviors=beh, article_dict=mapping, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                       unknown_representation="zeros", eval_mode=False, batch_size=100)[0]
    vec = {k: np.asarray(v, dtype=np.float32) / 7 for k, v in mapping.items()}  # DocVec-style float vectors
    loader = NRMSDataLoader(behaviors=train, article_dict=vec, history_column=DEFAULT_HISTORY_ARTICLE_ID_COL,
                            unknown_representation="mean", batch_size=8, kwargs={"tag": "x"})
    (his, pred), y = loader[0]
    assert his.dtype == np.float32 and loader.tag == "x"
    np.testing.assert_allclose(loader.lookup_article_matrix[0], np.mean(list(vec.values()), axis=0), rtol=1e-6)


def test_ebnerd_from_path_pads_and_joins():
    df = ebnerd_from_path(DATA, history_size=20, padding=0)
    assert len(df) == 1046
    known = df[df[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()]
    assert all(len(h) == 20 for h in known[DEFAULT_HISTORY_ARTICLE_ID_COL])
    raw = pd.read_parquet(DATA / "history.parquet").set_index(DEFAULT_USER_COL)[DEFAULT_HISTORY_ARTICLE_ID_COL]
    r = known.iloc[0]
    full = list(raw[r[DEFAULT_USER_COL]])
    want = ([0] * max(0, 20 - len(full)) + full)[-20:]
    assert list(r[DEFAULT_HISTORY_ARTICLE_ID_COL]) == want
