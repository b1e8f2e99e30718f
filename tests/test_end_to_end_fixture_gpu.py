"""End to end through the REAL pipeline on the reference's own fixture parquets (tests/golden/ebnerd = the reference's
test/data/ebnerd; articles synthesised as the reference's test_newsrec.py:37 does, its articles.parquet being absent):

    ebnerd_from_path -> sampling_strategy_wu2019 -> create_binary_labels_column -> NRMSDataLoaderPretransform ->
    NRMSModel.model.fit (2 epochs, shuffled batch order, dropout on) -> scorer.predict(eval loader) ->
    add_prediction_scores -> MetricEvaluator(AUC, MRR, nDCG)

compared STEP BY STEP with the float64 oracle fed the same batches in the same order (same counter-based dropout stream,
Keras-form Adam).  It is the nearest available stand-in for "AUC on ebnerd_small within +-0.002 of the reference" while
neither the dataset nor TensorFlow can be had: the whole chain is deterministic, so the metrics must agree to rounding.
"""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from oracle import nrms_numpy as on

pytestmark = pytest.mark.gpu
DATA = Path(__file__).parent / "golden" / "ebnerd"


def test_fit_predict_evaluate_on_the_reference_fixtures_follows_the_oracle(hip):
    from ebrec.evaluation import AucScore, MetricEvaluator, MrrScore, NdcgScore
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.callbacks import Callback
    from ebrec.models.newsrec.dataloader import NRMSDataLoaderPretransform
    from ebrec.utils._articles import create_article_id_to_value_mapping
    from ebrec.utils._behaviors import add_prediction_scores, create_binary_labels_column, ebnerd_from_path, sampling_strategy_wu2019
    from ebrec.utils._constants import (DEFAULT_ARTICLE_ID_COL, DEFAULT_HISTORY_ARTICLE_ID_COL, DEFAULT_INVIEW_ARTICLES_COL,
                                        DEFAULT_LABELS_COL)
    from ebrec.utils._frames import list_column

    H, T, V, D, seed = 10, 12, 200, 32, 7
    rng = np.random.default_rng(0)
    df = ebnerd_from_path(DATA, history_size=H, padding=0)
    df = df[df[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()].reset_index(drop=True)
    ids = sorted({a for l in list_column(df, DEFAULT_INVIEW_ARTICLES_COL) for a in l} |
                 {a for l in list_column(df, DEFAULT_HISTORY_ARTICLE_ID_COL) for a in l if a})
    ids = ids[: len(ids) * 9 // 10]  # 10 % of the articles unknown -> row 0 of the lookup matrix
    articles = pd.DataFrame({DEFAULT_ARTICLE_ID_COL: ids, "tokens": rng.integers(1, V, (len(ids), T)).tolist()})
    mapping = create_article_id_to_value_mapping(articles, value_col="tokens")
    n_train = int(len(df) * 0.8)
    train = create_binary_labels_column(sampling_strategy_wu2019(df.iloc[:n_train], npratio=4, shuffle=True, with_replacement=True, seed=123))
    valid = create_binary_labels_column(df.iloc[n_train:].reset_index(drop=True), shuffle=False)
    mk = lambda frame, ev, bs: NRMSDataLoaderPretransform(behaviors=frame, article_dict=mapping, unknown_representation="zeros",
                                                          history_column=DEFAULT_HISTORY_ARTICLE_ID_COL, eval_mode=ev, batch_size=bs)
    tl = mk(train, False, 32)
    assert len(tl) >= 10

    class hp:
        title_size, history_size, head_num, head_dim, attention_hidden_dim = T, H, 4, 8, 16
        optimizer, loss, dropout, learning_rate = "adam", "cross_entropy_loss", 0.2, 1e-3
        newsencoder_units_per_layer, newsencoder_l2_regularization = None, 1e-4

    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=3)
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    model = NRMSModel(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list([P[k] for k in on.PARAM_ORDER])

    class StepLosses(Callback):
        def on_train_begin(self, logs=None):
            self.losses = []

        def on_train_batch_end(self, batch, logs=None):
            self.losses.append(float(self.model._engine.loss_dev.item()))

    rec = StepLosses()
    epochs = 2
    hist = model.model.fit(tl, epochs=epochs, verbose=0, callbacks=[rec])

    # ---- the same run in the float64 oracle: same batches, same order (fit() shuffles batch ORDER with default_rng(seed))
    order_rng = np.random.default_rng(seed)
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    t, want_losses, epoch_means = 0, [], []
    for _ in range(epochs):
        tot = rows = 0
        for idx in order_rng.permutation(len(tl)):
            (his, pred), y = tl[int(idx)]
            t += 1
            L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, hp.loss, on.Drop(hp.dropout, seed, t))
            for k in P:
                on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=hp.learning_rate)
            want_losses.append(L)
            tot, rows = tot + L * len(his), rows + len(his)
        epoch_means.append(tot / rows)
    assert len(rec.losses) == len(want_losses) == epochs * len(tl)
    np.testing.assert_allclose(rec.losses, want_losses, rtol=0, atol=2e-4)  # every optimizer step of both epochs
    np.testing.assert_allclose(hist.history["loss"], epoch_means, rtol=0, atol=1e-4)

    # ---- scorer.predict on the held-out rows (ragged in-view lists, eval loader) -> the reference's evaluator
    vl = mk(valid, True, 16)
    got = model.scorer.predict(vl)
    want = np.concatenate([on.scorer_forward(*vl[i][0], P, hp.head_num, hp.head_dim) for i in range(len(vl))])
    assert got.shape == want.shape == (sum(len(l) for l in list_column(valid, DEFAULT_INVIEW_ARTICLES_COL)), 1)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-4)
    fns = lambda: [AucScore(), MrrScore(), NdcgScore(k=5), NdcgScore(k=10)]
    labels = list_column(valid, DEFAULT_LABELS_COL)
    m_got = MetricEvaluator(labels=labels, predictions=list_column(add_prediction_scores(valid, got.tolist()), "scores"), metric_functions=fns()).evaluate().evaluations
    m_want = MetricEvaluator(labels=labels, predictions=list_column(add_prediction_scores(valid, want.tolist()), "scores"), metric_functions=fns()).evaluate().evaluations
    assert set(m_got) == {"auc", "mrr", "ndcg@5", "ndcg@10"}
    for k in m_got:
        assert abs(m_got[k] - m_want[k]) <= 1e-6, (k, m_got[k], m_want[k])
    assert m_got["auc"] != 0.5  # the model has learnt something rank-relevant or anti-relevant: not a constant scorer
