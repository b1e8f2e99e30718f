"""Model-level parity of the MI355X NRMS host layer against the float64 oracle (gpu-marked).

These read like the tests the reference never had (no model numerics
tests): same constructor, same ``.model`` / ``.scorer`` / ``.newsencoder`` / ``.userencoder``
surface, outputs compared with the restated reference math on identical weights and inputs.
Forward tolerance: 1e-5 abs on probabilities / scores of O(1) -- 10x inside the 1e-4 fp32
budget of BASELINE.json's north_star.
"""
import numpy as np
import pytest
import torch

from oracle import nrms_numpy as on
from tests.hip_testutil import assert_close

pytestmark = pytest.mark.gpu


class HP:
    title_size = 30
    history_size = 20
    head_num = 20
    head_dim = 20
    attention_hidden_dim = 200
    optimizer = "adam"
    loss = "cross_entropy_loss"
    dropout = 0.2
    learning_rate = 1e-3
    newsencoder_units_per_layer = None
    newsencoder_l2_regularization = 1e-4


def make_hp(**kw):
    return type("hp", (HP,), kw)


def weight_list(P):
    return [P[k] for k in on.PARAM_ORDER]


def batch(rng, B, H, C, T, V, pad_frac=0.15, ids="uniform"):
    """ids="zipf": SURVEY.md 8(d)'s Z distribution (Zipf(1.0) over the V rows, id = frequency rank) -- with the padded history
    slots both distributions have, table row 0 then carries ~20 % of a batch's tokens."""
    if ids == "zipf":
        from bench import zipf_ids

        his, pred = zipf_ids(rng, (B, H, T), V), zipf_ids(rng, (B, C, T), V)
    else:
        his, pred = rng.integers(0, V, (B, H, T)), rng.integers(0, V, (B, C, T))
    pad = rng.random((B, H)) < pad_frac  # padded history slots = all-zero titles
    his[pad] = 0
    y = np.zeros((B, C), np.int8)
    y[np.arange(B), rng.integers(0, C, B)] = 1
    return his, pred, y


@pytest.fixture(scope="module")
def nrms(hip):
    from ebrec.models.newsrec import NRMSModel

    return NRMSModel


@pytest.mark.parametrize("cfg", [dict(), dict(history_size=50, title_size=30), dict(head_num=16, head_dim=16),
                                 dict(title_size=12, history_size=5, head_num=3, head_dim=8, attention_hidden_dim=17),
                                 dict(history_size=100, title_size=70, head_num=4, head_dim=20)])  # long-sequence attention kernels
def test_forward_matches_oracle_on_identical_weights(nrms, cfg):
    hp = make_hp(**cfg)
    V, D = 500, 300 if not cfg.get("head_dim") == 8 else 36
    rng = np.random.default_rng(0)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=3)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=1).from_keras_weight_list(weight_list(P))
    B, C = 7, 5
    his, pred, y = batch(rng, B, hp.history_size, C, hp.title_size, V)
    probs, s, _ = on.nrms_forward(his, pred, P, hp.head_num, hp.head_dim)
    got = m.model.predict((his, pred))
    assert got.shape == (B, C)
    assert_close(got, probs, rtol=0, atol=1e-5, what="softmax click probabilities")
    # scorer: sigmoid(u.n) per (history, single candidate) row, nrms.py:204-205
    one = pred[:, :1, :]
    sc = m.scorer.predict((his, one))
    assert sc.shape == (B, 1)
    assert_close(sc, on.scorer_forward(his, one, P, hp.head_num, hp.head_dim), rtol=0, atol=1e-5, what="scorer")
    # sub-models
    ne = m.newsencoder.predict(pred[0])
    want_ne, _ = on.news_encoder_fwd(pred[0], P, hp.head_num, hp.head_dim)
    assert_close(ne, want_ne, rtol=1e-5, atol=1e-5, what="newsencoder")
    ue = m.userencoder.predict(his)
    NEh, _ = on.news_encoder_fwd(his.reshape(-1, hp.title_size), P, hp.head_num, hp.head_dim)
    want_u, _ = on.user_encoder_from_news_fwd(NEh.reshape(B, hp.history_size, -1), P, hp.head_num, hp.head_dim)
    assert_close(ue, want_u, rtol=1e-5, atol=1e-5, what="userencoder")


@pytest.mark.parametrize("D", [300, 64])
def test_inference_with_the_gather_fused_into_the_projection_changes_no_bit(nrms, D):
    """encode_news / predict in inference mode fetch the projection's A operand straight from the table (table rows -> LDS -> MFMA,
    ebn_encoder_fwd_gather_f32: no Dropout between Embedding and K.dot at inference, nrms.py:125-139); against the two-step form
    (gather into X, then the same GEMM) the news vectors are bit-identical wherever both run the same block tile -- the same fma
    chains over the same operands -- for D with a partial last 16-deep slab (300) and without; at 270 token rows the planner gives
    the two-step form another tile (other summation order: rounding-level differences); fewer than 256 rows fall back to it; an
    out-of-range id raises either way."""
    hp = make_hp()
    V, seed = 777, 3
    rng = np.random.default_rng(5)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=2)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list(weight_list(P))
    eng = m._engine
    for n_titles in (700, 9, 3):  # 21000 / 270 token rows: fused; 90: the two-step fallback inside the same call
        ids = rng.integers(0, V, (n_titles, hp.title_size))
        ids[0] = 0
        eng.fuse_eval_gather = True
        a = eng.encode_news(ids).cpu().numpy()
        eng.fuse_eval_gather = False
        b = eng.encode_news(ids).cpu().numpy()
        if n_titles == 9:
            assert_close(a, b, rtol=2e-5, atol=1e-6, what="fused vs two-step news vectors, different block tiles")
        else:
            assert np.array_equal(a, b), n_titles
    ne = on.news_encoder_fwd(ids, P, hp.head_num, hp.head_dim)[0]
    assert_close(a, ne, rtol=1e-4, atol=1e-5, what="fused-gather news vectors vs oracle")
    eng.fuse_eval_gather = True
    bad = rng.integers(0, V, (40, hp.title_size))
    bad[7, 3] = V + 5
    with pytest.raises(IndexError):
        eng.encode_news(bad)


def test_one_model_scores_every_history_length_from_1_to_50(nrms):
    """The reference's history-length sweep (ebnerd_nrms_doc_hist.py:270-300) feeds ONE trained model histories truncated to
    1 ... 50 entries: no weight of the user encoder depends on H (layers.py:200-254, 55-81), so scorer / userencoder take H
    from the batch.  Every L = 1 ... 50 of the news-level attention kernels (one MFMA tile up to 32, 2x2 tiles beyond)
    against the float64 oracle."""
    hp = make_hp()  # history_size = 20: the length the model would have been trained with
    V, D = 300, 64
    rng = np.random.default_rng(15)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=12)
    m = nrms(hp, word2vec_embedding=P["emb"]).from_keras_weight_list(weight_list(P))
    for H in range(1, 51):
        B = 3 if H > 8 else 5
        his = rng.integers(0, V, (B, H, hp.title_size))
        if H > 2:
            his[0, : H // 2] = 0  # left padding of a short history
        one = rng.integers(0, V, (B, 1, hp.title_size))
        want = on.scorer_forward(his, one, P, hp.head_num, hp.head_dim)
        assert_close(m.scorer.predict((his, one)), want, rtol=0, atol=1e-5, what=f"scorer at history length {H}")
        if H in (1, 7, 33, 50):
            NEh, _ = on.news_encoder_fwd(his.reshape(-1, hp.title_size), P, hp.head_num, hp.head_dim)
            want_u, _ = on.user_encoder_from_news_fwd(NEh.reshape(B, H, -1), P, hp.head_num, hp.head_dim)
            assert_close(m.userencoder.predict(his), want_u, rtol=1e-5, atol=1e-5, what=f"userencoder at history length {H}")
    # training keeps hparams.history_size
    with pytest.raises(ValueError):
        m.train_step(rng.integers(0, V, (2, 7, hp.title_size)), rng.integers(0, V, (2, 5, hp.title_size)), np.eye(5)[[0, 1]])


def test_scorer_on_eval_loader_layout_dedups_but_keeps_row_order(nrms):
    """dataloader.py:99-107 layout: history repeated per candidate, pred (sum C_i, 1, T)."""
    hp = make_hp()
    V, D = 300, 64
    rng = np.random.default_rng(5)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=4)
    m = nrms(hp, word2vec_embedding=P["emb"]).from_keras_weight_list(weight_list(P))
    n_imp = 4
    his_imp = rng.integers(0, V, (n_imp, hp.history_size, hp.title_size))
    counts = [3, 1, 6, 2]
    his = np.repeat(his_imp, counts, axis=0)
    pred = rng.integers(0, V, (sum(counts), 1, hp.title_size))
    pred[4] = pred[5]  # a repeated candidate title
    got = m.scorer.predict((his, pred))
    want = on.scorer_forward(his, pred, P, hp.head_num, hp.head_dim)
    assert_close(got, want, rtol=0, atol=1e-5, what="ragged scorer")


@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss", "log_loss_probs"])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_training_steps_follow_the_oracle_trajectory(nrms, loss, p):
    """3 optimizer steps: loss values and every updated tensor vs float64 oracle + Keras-form Adam,
    with the shared counter-based dropout stream (training-mode parity).  "log_loss_probs" = hparams.loss "log_loss" with
    bce_on="probs": binary cross-entropy on the clipped softmax outputs, the hedge next to the logits form."""
    hp = make_hp(loss="log_loss" if loss == "log_loss_probs" else loss, dropout=p, learning_rate=1e-3)
    if loss == "log_loss_probs":
        nrms_cls = nrms
        nrms = lambda *a, **k: nrms_cls(*a, bce_on="probs", **k)
    V, D, seed = 400, 300, 11
    rng = np.random.default_rng(7)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=5)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list(weight_list(P))
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    B, C = 8, 5
    for t in range(1, 4):
        his, pred, y = batch(rng, B, hp.history_size, C, hp.title_size, V)
        L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, loss,
                                         on.Drop(p, seed, t) if p > 0 else None)
        got_L = float(m.train_step(his, pred, y).item())
        assert abs(got_L - L) <= 2e-5 * max(1.0, abs(L)), (t, got_L, L)
        for k in P:
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=1e-3)
    got = dict(zip(on.PARAM_ORDER, m.model.get_weights()))
    for k in on.PARAM_ORDER:
        # Adam's m/(sqrt(v)+eps) amplifies fp32 gradient noise where |g| ~ eps: compare the update
        step = np.abs(P[k] - on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=5)[k].astype(np.float32))
        assert_close(got[k].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"weights {k} after 3 steps")


@pytest.mark.parametrize("H", [20, 50, 100])
def test_fused_user_head_step_equals_the_step_with_separate_kernels(nrms, H):
    """The stage call runs the per-impression head of a step as ONE launch where it fits (H = 20, 50) and as its separate
    kernels otherwise (H = 100: 100 x 600 floats exceed a workgroup's LDS) -- and `fuse_user_head = False` forces the latter.
    Loss and every gradient buffer of the two forms agree to fp32 summation-order noise."""
    hp = make_hp(history_size=H, dropout=0.2)
    V, D, seed = 300, 64, 4
    rng = np.random.default_rng(21)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=9)
    his, pred, y = batch(rng, 5, H, 5, hp.title_size, V)
    out = []
    for fuse in (True, False):
        m = nrms(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list(weight_list(P))
        m._engine.fuse_user_head = fuse
        m._engine.keep_table_grad = True
        loss = float(m.train_step(his, pred, y).item())
        out.append((loss, m._engine.params.grad.cpu().numpy().astype(np.float64), m._engine.table_grad.cpu().numpy().astype(np.float64)))
    (l0, g0, t0), (l1, g1, t1) = out
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l1))
    assert_close(g0, g1, rtol=1e-4, atol=1e-7 + 2e-5 * np.abs(g1).max(), what="dense gradients, fused vs separate")
    assert_close(t0, t1, rtol=1e-4, atol=1e-7 + 2e-5 * np.abs(t1).max(), what="table gradient, fused vs separate")
    L, _, g = on.nrms_loss_and_grads(his, pred, y, {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}, hp.head_num, hp.head_dim,
                                     "cross_entropy_loss", on.Drop(0.2, seed, 1))
    assert abs(l0 - L) <= 2e-5 * max(1.0, abs(L))


@pytest.mark.parametrize("train_embedding", [False, True])
def test_split_precision_model_holds_the_exact_model_tolerances(nrms, train_embedding):
    """NRMSModel(precision="split"): the news encoder's projection GEMMs as bf16x6 split products on the bf16 matrix pipe.
    The SAME assertions as the exact path: forward probabilities to 1e-5 of the float64 oracle (10x inside north_star's 1e-4),
    loss / gradients / three-step Adam trajectory to the tolerances of test_training_steps_follow_the_oracle_trajectory."""
    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    V, D, seed = 400, 300, 11
    rng = np.random.default_rng(8)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=5)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=seed, precision="split", train_embedding=train_embedding).from_keras_weight_list(weight_list(P))
    assert m._engine.precision == "split"
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    P0 = {k: v.copy() for k, v in P.items()}
    his, pred, y = batch(rng, 7, hp.history_size, 5, hp.title_size, V)
    probs, _, _ = on.nrms_forward(his, pred, P, hp.head_num, hp.head_dim)
    assert_close(m.model.predict((his, pred)), probs, rtol=0, atol=1e-5, what="split precision: click probabilities")
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    m._engine.keep_table_grad = True
    for t in range(1, 4):
        his, pred, y = batch(rng, 8, hp.history_size, 5, hp.title_size, V)
        L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, "cross_entropy_loss", on.Drop(0.2, seed, t))
        got_L = float(m.train_step(his, pred, y).item())
        assert abs(got_L - L) <= 2e-5 * max(1.0, abs(L)), (t, got_L, L)
        if t == 1:
            want = np.concatenate([g["n_WQ"], g["n_WK"], g["n_WV"]], 1)
            assert_close(m._engine.params.g("n_Wqkv").cpu().numpy(), want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what="split precision: dWqkv")
            if train_embedding:
                assert_close(m._engine.table_grad.cpu().numpy(), g["emb"], rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(g["emb"]).max(), what="split precision: dEmb")
        for k in P:
            if k == "emb" and not train_embedding:
                continue
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=1e-3)
    got = dict(zip(on.PARAM_ORDER, m.model.get_weights()))
    for k in on.PARAM_ORDER:
        step = np.abs(P[k] - P0[k])
        assert_close(got[k].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"split precision: weights {k} after 3 steps")
    with pytest.raises(ValueError):
        nrms(hp, word_emb_dim=16, vocab_size=50, precision="bf16")


def test_gradients_match_oracle_directly(nrms):
    """Backward parity without the optimizer in the way: raw gradient buffers after one step."""
    hp = make_hp(dropout=0.2)
    V, D, seed = 300, 300, 2
    rng = np.random.default_rng(9)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=6)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list(weight_list(P))
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    his, pred, y = batch(rng, 6, hp.history_size, 5, hp.title_size, V)
    L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, "cross_entropy_loss", on.Drop(0.2, seed, 1))
    m._engine.keep_table_grad = True  # (the fused accumulator->Adam sweep never writes the fp32 gradient)
    m.train_step(his, pred, y)
    eng = m._engine
    E = eng.E
    for pre in ("n", "u"):
        gq = eng.params.g(f"{pre}_Wqkv").cpu().numpy()
        want = np.concatenate([g[f"{pre}_WQ"], g[f"{pre}_WK"], g[f"{pre}_WV"]], 1)
        assert_close(gq, want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what=f"{pre} dWqkv")
        for nm in ("W", "b", "q"):
            want = g[f"{pre}_{nm}"].reshape(eng.params.shapes[f"{pre}_{nm}"])
            assert_close(eng.params.g(f"{pre}_{nm}").cpu().numpy(), want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what=f"{pre} d{nm}")
    assert_close(eng.table_grad.cpu().numpy(), g["emb"], rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(g["emb"]).max(), what="dEmb")


def test_frozen_table_is_untouched_and_has_no_moments(nrms):
    hp = make_hp(dropout=0.0)
    rng = np.random.default_rng(1)
    emb = rng.standard_normal((200, 64)).astype(np.float32)
    m = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=False)
    his, pred, y = batch(rng, 4, hp.history_size, 5, hp.title_size, 200)
    w0 = m.model.get_weights()
    m.train_step(his, pred, y)
    w1 = m.model.get_weights()
    assert np.array_equal(w0[0], w1[0]) and not np.array_equal(w0[1], w1[1])
    assert not hasattr(m._engine, "table_m")


def test_seeded_init_reproduces_reference_quirks(nrms):
    hp = make_hp()
    m = nrms(hp, word_emb_dim=32, vocab_size=100, seed=42)
    w = m.model.get_weights()
    assert len(w) == 13 and w[0].shape == (100, 32)
    assert np.array_equal(w[1], w[2]) and np.array_equal(w[2], w[3])  # WQ=WK=WV at init (quirk 5)
    assert np.all(w[5] == 0) and w[6].shape == (200, 1)
    lim = np.sqrt(6.0 / (100 + 32))
    assert np.abs(w[0]).max() <= lim and np.abs(w[0]).max() > 0.9 * lim
    m2 = nrms(hp, word_emb_dim=32, vocab_size=100, seed=42)
    assert all(np.array_equal(a, b) for a, b in zip(w, m2.model.get_weights()))
    assert m.model.count_params() == 100 * 32 + 32 * 1200 + 400 * 200 + 200 + 200 + 400 * 1200 + 400 * 200 + 200 + 200


def test_out_of_range_token_raises(nrms):
    hp = make_hp()
    m = nrms(hp, word_emb_dim=16, vocab_size=50, seed=0)
    his = np.zeros((2, hp.history_size, hp.title_size), np.int64)
    pred = np.zeros((2, 3, hp.title_size), np.int64)
    pred[1, 2, 5] = 50
    with pytest.raises(IndexError):
        m.model.predict((his, pred))
    with pytest.raises(IndexError):  # device-side check for ids that are already on the GPU
        m._engine.forward(torch.from_numpy(his).cuda(), torch.from_numpy(pred).cuda())


def test_fit_predict_compile_save_load_surface(nrms, tmp_path):
    """The call sequence of nrms_dummy.py:46-47 and ebnerd_nrms.py:244-260."""
    from ebrec.models.newsrec.callbacks import EarlyStopping, ModelCheckpoint, ReduceLROnPlateau

    hp = make_hp(learning_rate=1e-3, dropout=0.0)  # no dropout: "the loss goes down on 70 memorised rows" must not depend on a mask draw
    rng = np.random.default_rng(3)
    V = 120
    m = nrms(hp, word2vec_embedding=rng.random((V, 40)), seed=5)
    m.model.compile(optimizer=m.model.optimizer, loss=m.model.loss, metrics=["AUC"])
    his, pred, y = batch(rng, 70, hp.history_size, 5, hp.title_size, V)
    ck = tmp_path / "weights.h5"
    hist = m.model.fit((his, pred), y, batch_size=32, epochs=5, verbose=0, validation_data=((his[:20], pred[:20]), y[:20]),
                       callbacks=[EarlyStopping(monitor="val_auc", mode="max", patience=4, restore_best_weights=True),
                                  ModelCheckpoint(filepath=str(ck), monitor="val_auc", mode="max", save_best_only=True,
                                                  save_weights_only=True),
                                  ReduceLROnPlateau(monitor="val_auc", mode="max", factor=0.2, patience=2, min_lr=1e-6)])
    assert set(hist.history) >= {"loss", "auc", "val_loss", "val_auc"} and len(hist.history["loss"]) == 5
    assert hist.history["loss"][-1] < hist.history["loss"][0]  # it learns the 70 rows
    assert ck.exists()
    p1 = m.model.predict((his, pred))
    assert p1.shape == (70, 5) and np.allclose(p1.sum(1), 1, atol=1e-5)
    m2 = nrms(hp, word2vec_embedding=rng.random((V, 40)), seed=9)
    m2.model.load_weights(str(ck))
    m.model.load_weights(str(ck))
    assert np.array_equal(m2.model.predict((his, pred)), m.model.predict((his, pred)))
    lines = []
    m.model.summary(print_fn=lines.append)
    assert any("news.attn.WQ" in l for l in lines)
    assert m.model.variables[0].name == "news.emb" and "cuda" in m.model.variables[0].device


@pytest.mark.parametrize("train_embedding", [False, True])
def test_hipgraph_replay_equals_kernel_by_kernel_launch(nrms, train_embedding):
    """enable_graphs(): same kernels replayed from a captured graph, dropout keys / Adam step sizes read
    from the device step state -> same trajectory as eager launches (bitwise with a frozen table; the
    trainable table goes through fp32 atomics, so only to rounding)."""
    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(21)
    V = 300
    emb = rng.standard_normal((V, 64)).astype(np.float32)
    ms = [nrms(hp, word2vec_embedding=emb, seed=13, train_embedding=train_embedding) for _ in range(2)]
    ms[1]._engine.enable_graphs()
    batches = [batch(rng, 6, hp.history_size, 5, hp.title_size, V) for _ in range(4)]
    losses = [[float(m.train_step(*b).item()) for b in batches] for m in ms]
    w0, w1 = ms[0].model.get_weights(), ms[1].model.get_weights()
    if train_embedding:
        assert losses[0] == losses[1]  # the trainable table is deterministic too (fixed-point accumulator)
        assert all(np.array_equal(a, b) for a, b in zip(w0, w1))
    else:
        assert losses[0] == losses[1]
        assert all(np.array_equal(a, b) for a, b in zip(w0, w1))
    assert ms[1]._engine.read_state().step == 4
    # a different batch shape captures its own graph and keeps working
    b2 = batch(rng, 3, hp.history_size, 4, hp.title_size, V)
    l0, l1 = (float(m.train_step(*b2).item()) for m in ms)
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))


@pytest.mark.parametrize("train_embedding", [False, True])
def test_row_sharded_table_path_on_one_gpu_equals_replicated(nrms, train_embedding):
    """shard_table=True at world size 1 exercises the whole routed-lookup code path (dedup, local gather,
    expand-with-dropout, per-unique-row gradient, owner scatter) against the replicated-table path."""
    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(31)
    V = 257
    emb = rng.standard_normal((V, 64)).astype(np.float32)
    a = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=train_embedding)
    b = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=train_embedding, shard_table=True)
    his, pred, y = batch(rng, 5, hp.history_size, 5, hp.title_size, V)
    assert np.array_equal(a.model.predict((his, pred)), b.model.predict((his, pred)))
    for _ in range(2):
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la))
    for wa, wb in zip(a.model.get_weights(), b.model.get_weights()):
        assert np.allclose(wa, wb, rtol=1e-4, atol=1e-6)


def test_scorer_with_article_cache_equals_per_batch_encoding(nrms):
    """scorer.predict(eval loader): encoding the loader's article matrix once and scoring batches from the cached news
    vectors gives the scores of the per-batch path (and of the reference layout with repeated histories)."""
    import pandas as pd

    from ebrec.models.newsrec.dataloader import NRMSDataLoader

    hp = make_hp(history_size=6, title_size=8)
    rng = np.random.default_rng(43)
    V, n_art, n = 150, 40, 50
    art_ids = np.arange(500, 500 + n_art)
    mapping = {int(a): rng.integers(1, V, 8).tolist() for a in art_ids}
    df = pd.DataFrame({"user_id": rng.integers(0, 9, n), "article_id_fixed": [rng.choice(np.append(art_ids, 0), 6).tolist() for _ in range(n)],
                       "article_ids_inview": [rng.choice(np.append(art_ids, 7), int(rng.integers(1, 9))).tolist() for _ in range(n)],
                       "labels": [[0] for _ in range(n)]})
    df["labels"] = [[0] * len(v) for v in df["article_ids_inview"]]
    loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros",
                            eval_mode=True, batch_size=16)
    m = nrms(hp, word2vec_embedding=rng.standard_normal((V, 32)).astype(np.float32), seed=3)
    cached = m.scorer.predict(loader)
    m.scorer.cache_articles = False
    per_batch = m.scorer.predict(loader)
    repeated = np.concatenate([m.scorer.predict(loader[i][0]) for i in range(len(loader))])
    assert cached.shape == per_batch.shape == (sum(len(v) for v in df["article_ids_inview"]), 1)
    assert_close(cached, per_batch, rtol=0, atol=2e-6, what="article cache vs per-batch")
    assert_close(cached, repeated, rtol=0, atol=2e-6, what="article cache vs repeated-history layout")


def test_evaluate_with_article_cache_equals_per_batch_encoding(nrms):
    """model.evaluate(train-layout loader): loss / AUC from cached news vectors == the per-batch forward."""
    import pandas as pd

    from ebrec.models.newsrec.dataloader import NRMSDataLoader

    hp = make_hp(history_size=6, title_size=8)
    rng = np.random.default_rng(47)
    V, n_art, n = 150, 40, 70
    art_ids = np.arange(500, 500 + n_art)
    mapping = {int(a): rng.integers(1, V, 8).tolist() for a in art_ids}
    df = pd.DataFrame({"user_id": rng.integers(0, 9, n), "article_id_fixed": [rng.choice(np.append(art_ids, 0), 6).tolist() for _ in range(n)],
                       "article_ids_inview": [rng.choice(art_ids, 5).tolist() for _ in range(n)],
                       "labels": [np.eye(5, dtype=int)[rng.integers(0, 5)].tolist() for _ in range(n)]})
    loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros", batch_size=16)
    for loss in ("cross_entropy_loss", "log_loss"):
        m = nrms(make_hp(history_size=6, title_size=8, loss=loss), word2vec_embedding=rng.standard_normal((V, 32)).astype(np.float32), seed=3)
        m.model.compile(optimizer="adam", loss=m.model.loss, metrics=["AUC"])
        a = m.model.evaluate(loader, return_dict=True)
        m.model.cache_articles = False
        b = m.model.evaluate(loader, return_dict=True)
        assert abs(a["loss"] - b["loss"]) <= 2e-6 * max(1.0, abs(b["loss"])) and abs(a["auc"] - b["auc"]) <= 1e-6, (a, b)


def test_device_resident_batches_equal_host_batches(nrms):
    """Batches handed over as device tensors in the step's dtypes go through the one-launch prologue copy (ebn_copy3);
    the step is the same as with numpy batches, bit for bit (graph replay and kernel-by-kernel)."""
    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(77)
    for use_graph in (True, False):
        a = nrms(hp, word_emb_dim=16, vocab_size=90, seed=5)
        b = nrms(hp, word_emb_dim=16, vocab_size=90, seed=5)
        a._engine.use_graph = b._engine.use_graph = use_graph
        for _ in range(3):
            his, pred, y = batch(rng, 4, hp.history_size, 5, hp.title_size, 90)
            la = float(a.train_step(his, pred, y).item())
            dev = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x)).to(device="cuda", dtype=dt)
            lb = float(b.train_step(dev(his, torch.int32), dev(pred, torch.int32), dev(y, torch.float32)).item())
            assert la == lb
        assert all(np.array_equal(x, z) for x, z in zip(a.model.get_weights(), b.model.get_weights()))


def test_indexed_batches_equal_token_batches(nrms):
    """Device-side batch assembly: article-row numbers + the loader's token matrix in HBM give the same step as the
    host-gathered token batches (dataloader.py:169-179), bit for bit with a frozen table."""
    import pandas as pd

    from ebrec.models.newsrec.dataloader import NRMSDataLoader

    hp = make_hp(dropout=0.2, learning_rate=1e-3, history_size=6, title_size=8)
    rng = np.random.default_rng(41)
    V, n_art, n = 120, 50, 64
    art_ids = np.arange(1000, 1000 + n_art)
    mapping = {int(a): rng.integers(1, V, 8).tolist() for a in art_ids}
    df = pd.DataFrame({"user_id": rng.integers(0, 9, n), "article_id_fixed": [rng.choice(np.append(art_ids, 0), 6).tolist() for _ in range(n)],
                       "article_ids_inview": [rng.choice(np.append(art_ids, 7), 5).tolist() for _ in range(n)],
                       "labels": [np.eye(5, dtype=int)[rng.integers(0, 5)].tolist() for _ in range(n)]})
    loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros", batch_size=16)
    emb = rng.standard_normal((V, 32)).astype(np.float32)
    a = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=False)
    b = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=False)
    b._engine.set_article_matrix(loader.lookup_article_matrix)
    for i in range(len(loader)):
        (his, pred), y = loader[i]
        (hi, pi), yi = loader.index_batch(i)
        assert np.array_equal(loader.lookup_article_matrix[hi], his) and np.array_equal(loader.lookup_article_matrix[pi], pred) and np.array_equal(y, yi)
        la = float(a.train_step(his, pred, y).item())
        lb = float(b._engine.train_step(hi, pi, yi, indexed=True).item())
        assert la == lb
    assert all(np.array_equal(x, z) for x, z in zip(a.model.get_weights(), b.model.get_weights()))
    c = nrms(hp, word2vec_embedding=emb, seed=3, train_embedding=False)
    c.model.fit(loader, epochs=1, verbose=0, shuffle=False)  # fit() picks the indexed path by itself
    assert c._engine._article_matrix_src is loader.lookup_article_matrix
    assert all(np.array_equal(x, z) for x, z in zip(a.model.get_weights(), c.model.get_weights()))
    with pytest.raises(IndexError):
        b._engine.set_article_matrix(np.full((3, 8), V))


def test_trainable_table_training_is_bitwise_reproducible(nrms):
    """deterministic=True (default): the embedding gradient goes through the order-independent fixed-point
    accumulator, so two runs give identical bits even with thousands of duplicate tokens per step."""
    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(51)
    V = 40  # tiny vocabulary: every row is hit ~600 times per step
    emb = rng.standard_normal((V, 32)).astype(np.float32)
    batches = [batch(rng, 8, hp.history_size, 5, hp.title_size, V) for _ in range(3)]
    runs = []
    for _ in range(2):
        m = nrms(hp, word2vec_embedding=emb, seed=5)
        losses = [float(m.train_step(*b).item()) for b in batches]
        runs.append((losses, m.model.get_weights()))
    assert runs[0][0] == runs[1][0]
    assert all(np.array_equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    m2 = nrms(hp, word2vec_embedding=emb, seed=5, deterministic=False)  # fp32 atomics: same to rounding
    l2 = [float(m2.train_step(*b).item()) for b in batches]
    assert np.allclose(l2, runs[0][0], rtol=1e-5)


@pytest.mark.parametrize("B,H,C,T", [(1, 1, 1, 1), (1, 20, 1, 30), (3, 2, 250, 5), (33, 5, 2, 7)])
def test_edge_shapes_forward_and_one_step(nrms, B, H, C, T):
    """Degenerate and ragged sizes: single impression / history slot / candidate / token, a 250-wide in-view list
    (beyond-accuracy rows), a batch that is not a multiple of anything."""
    hp = make_hp(history_size=H, title_size=T, head_num=4, head_dim=20, attention_hidden_dim=11, dropout=0.1)
    V, D = 97, 24
    rng = np.random.default_rng(B * 1000 + C)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=8)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=4).from_keras_weight_list(weight_list(P))
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    his, pred, y = batch(rng, B, H, C, T, V)
    probs, _, _ = on.nrms_forward(his, pred, P, hp.head_num, hp.head_dim)
    assert_close(m.model.predict((his, pred)), probs, rtol=0, atol=1e-5, what="edge forward")
    L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, "cross_entropy_loss", on.Drop(0.1, 4, 1))
    got = float(m.train_step(his, pred, y).item())
    assert abs(got - L) <= 2e-5 * max(1.0, abs(L))
    want = np.concatenate([g["u_WQ"], g["u_WK"], g["u_WV"]], 1)
    assert_close(m._engine.params.g("u_Wqkv").cpu().numpy(), want, rtol=1e-4, atol=1e-7 + 1e-4 * np.abs(want).max(), what="edge du_Wqkv")


def test_varying_batch_sizes_empty_inputs_and_large_eval_batches(nrms):
    hp = make_hp(history_size=4, title_size=6, head_num=2, head_dim=16, attention_hidden_dim=7, dropout=0.0)
    rng = np.random.default_rng(61)
    V = 60
    P = on.random_nrms_params(V, 16, 2, 16, 7, seed=2)
    m = nrms(hp, word2vec_embedding=P["emb"], seed=1).from_keras_weight_list(weight_list(P))
    m._engine.enable_graphs()
    his, pred, y = batch(rng, 70, 4, 3, 6, V)
    h = m.model.fit((his, pred), y, batch_size=32, epochs=2, verbose=0)  # batches of 32, 32, 6 -> three graph shapes
    assert len(h.history["loss"]) == 2 and np.isfinite(h.history["loss"]).all()
    assert m.model.predict((his[:0], pred[:0])).shape[0] == 0
    assert m.scorer.predict((his[:0], pred[:0, :1])).shape == (0, 1)
    # more titles than one encode chunk (8192): chunked path == oracle
    P2 = dict(zip(on.PARAM_ORDER, [w.astype(np.float64).reshape(P[k].shape) for k, w in zip(on.PARAM_ORDER, m.model.get_weights())]))
    ids = rng.integers(0, V, (9000, 6))
    want, _ = on.news_encoder_fwd(ids, P2, 2, 16)
    assert_close(m.newsencoder.predict(ids), want, rtol=2e-5, atol=2e-5, what="chunked news encoding")


# ---------------------------------------------------------------- optional per-token Dense/BN stack (nrms.py:142-152)
def _mlp_weight_list(P, units):
    out = [P["emb"], P["n_WQ"], P["n_WK"], P["n_WV"]]
    for l in range(len(units)):
        out += [P[f"n_d{l}_W"], P[f"n_d{l}_b"], P[f"n_bn{l}_g"], P[f"n_bn{l}_b"], P[f"n_bn{l}_mean"], P[f"n_bn{l}_var"]]
    return out + [P[k] for k in ("n_W", "n_b", "n_q", "u_WQ", "u_WK", "u_WV", "u_W", "u_b", "u_q")]


@pytest.mark.parametrize("p,l2", [(0.0, 0.0), (0.2, 1e-4)])
def test_news_encoder_units_per_layer_branch_matches_oracle(nrms, p, l2):
    units = [48, 32]  # must end with head_num*head_dim
    hp = make_hp(head_num=4, head_dim=8, attention_hidden_dim=10, history_size=5, title_size=6, dropout=p,
                 newsencoder_units_per_layer=units, newsencoder_l2_regularization=l2)
    V, D, seed = 80, 20, 6
    rng = np.random.default_rng(71)
    P = on.random_nrms_params(V, D, 4, 8, 10, seed=3)
    on.add_mlp_params(P, units, 32, 10, seed=4)
    P = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in P.items()}
    m = nrms(hp, word2vec_embedding=P["emb"], seed=seed)
    assert len(m.model.get_weights()) == 13 + 12 and "news.bn1.moving_variance" in [v.name for v in m.model.variables]
    m.model.set_weights(_mlp_weight_list(P, units))
    his, pred, y = batch(rng, 6, 5, 4, 6, V)
    probs, _, _ = on.nrms_mlp_forward(his, pred, P, 4, 8, training=False)
    assert_close(m.model.predict((his, pred)), probs, rtol=0, atol=2e-5, what="units branch, inference (moving statistics)")
    L, _, g, stats = on.nrms_mlp_loss_and_grads(his, pred, y, P, 4, 8, l2=l2, training=True, drop=on.Drop(p, seed, 1) if p > 0 else None)
    m._engine.keep_table_grad = True
    got = float(m.train_step(his, pred, y).item())
    assert abs(got - L) <= 3e-5 * max(1.0, abs(L)), (got, L)
    eng = m._engine
    for k in ("n_d0_W", "n_d0_b", "n_bn0_g", "n_bn1_b", "n_d1_W", "n_W", "n_b", "n_q", "u_W"):
        want = g[k].reshape(eng.params.shapes[k])
        assert_close(eng.params.g(k).cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what=f"d{k}")
    want = np.concatenate([g["n_WQ"], g["n_WK"], g["n_WV"]], 1)
    assert_close(eng.params.g("n_Wqkv").cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what="dn_Wqkv")
    assert_close(eng.table_grad.cpu().numpy(), g["emb"], rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(g["emb"]).max(), what="dEmb")
    Pn = dict(P)
    on.bn_update_moving(Pn, stats, prefix="n_")  # history call site, then candidates
    for l in range(2):
        assert_close(eng.mlp.bn_mean[l].cpu().numpy(), Pn[f"n_bn{l}_mean"], rtol=1e-5, atol=1e-6, what=f"moving mean {l}")
        assert_close(eng.mlp.bn_var[l].cpu().numpy(), Pn[f"n_bn{l}_var"], rtol=1e-5, atol=1e-6, what=f"moving var {l}")
    with pytest.raises(ValueError, match="must end with"):
        nrms(make_hp(head_num=4, head_dim=8, newsencoder_units_per_layer=[48, 30]), word_emb_dim=8, vocab_size=10)


@pytest.mark.gpu
@pytest.mark.parametrize("train_embedding,graph", [(False, True), (True, False)])
def test_one_finishing_launch_of_the_backward_changes_no_bit_of_a_training_run(hip, train_embedding, graph):
    """engine.defer_finish (default): the split-K sums of dW / dWqkv, the AttLayer2 d(q) / d(b) column sums and the head's d(q) / d(b) /
    loss sums run as ONE launch at the end of the backward instead of four launches of the dependent chain.  Same summation orders:
    three steps with it are bit-identical -- losses, every gradient, every weight -- to three steps with the stand-alone passes."""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(3)
    V, D = 700, 128
    emb = rng.standard_normal((V, D)).astype(np.float32)
    a = NRMSModel(hp, word2vec_embedding=emb, seed=5, train_embedding=train_embedding)
    b = NRMSModel(hp, word2vec_embedding=emb, seed=5, train_embedding=train_embedding)
    assert a._engine.defer_finish and a._engine._deferred(5)
    a._engine.adam_in_finish = False  # (this test isolates the finishing launch; the optimizer riding in it has its own test below)
    b._engine.defer_finish = False
    assert not b._engine._deferred(5)
    if graph:
        a._engine.enable_graphs()
        b._engine.enable_graphs()
    for t in range(3):
        his, pred, y = batch(rng, 40, hp.history_size, 5, hp.title_size, V)  # 30 000 title tokens: the weight-gradient GEMMs split K
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert la == lb, (t, la, lb)
        assert torch.equal(a._engine.params.grad, b._engine.params.grad), t
    for wa, wb in zip(a.model.get_weights(), b.model.get_weights()):
        assert np.array_equal(wa, wb)


@pytest.mark.gpu
@pytest.mark.parametrize("train_embedding,graph", [(False, True), (True, False), (True, True)])
def test_adam_inside_the_finishing_launch_equals_the_separate_optimizer_launch(hip, train_embedding, graph):
    """engine.adam_in_finish (default, one rank): ebn_grad_finish_adam_f32 applies Keras-form Adam to each dense gradient element in the
    thread that has just summed it (and to the user encoder's kernels in blocks behind) instead of a launch of its own; with a trainable
    table the finishing launch moves behind the input-gradient GEMM, which reads Wqkv.  Same arithmetic element by element: the first
    step's loss and every gradient are bit-identical, weights and moments agree to fp32 rounding over three steps."""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(3)
    V, D = 700, 128
    emb = rng.standard_normal((V, D)).astype(np.float32)
    a = NRMSModel(hp, word2vec_embedding=emb, seed=5, train_embedding=train_embedding)
    b = NRMSModel(hp, word2vec_embedding=emb, seed=5, train_embedding=train_embedding)
    assert a._engine.adam_in_finish and a._engine._adam_fused(5)
    b._engine.adam_in_finish = False
    if graph:
        a._engine.enable_graphs()
        b._engine.enable_graphs()
    for t in range(3):
        his, pred, y = batch(rng, 40, hp.history_size, 5, hp.title_size, V)
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert a._engine._adam_done_in_finish and not b._engine._adam_done_in_finish
        if t == 0:
            assert la == lb and torch.equal(a._engine.params.grad, b._engine.params.grad)
        assert abs(la - lb) <= 2e-6 * abs(lb), (t, la, lb)
        assert_close(a._engine.params.grad.cpu().numpy(), b._engine.params.grad.cpu().numpy(), rtol=1e-4, atol=1e-7, what=f"gradients, step {t}")
    for what in ("data", "m", "v"):
        assert_close(getattr(a._engine.params, what).cpu().numpy(), getattr(b._engine.params, what).cpu().numpy(), rtol=2e-5, atol=5e-8, what=f"dense {what}")  # (atol: 5e-5 of one Adam step of lr = 1e-3 -- a weight that three steps moved through zero)
    for wa, wb in zip(a.model.get_weights(), b.model.get_weights()):
        assert_close(wa, wb, rtol=2e-5, atol=1e-8, what="weights after three steps")
    assert int(a._engine.read_state().step) == int(b._engine.read_state().step) == 3

