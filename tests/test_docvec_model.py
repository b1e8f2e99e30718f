"""NRMSDocVec through the C ABI against the float64 oracle (gpu-marked)."""
import numpy as np
import pytest

from oracle import nrms_numpy as on
from tests.hip_testutil import ReluTieGate, assert_close

pytestmark = pytest.mark.gpu


def make_hp(**kw):
    base = dict(title_size=768, history_size=20, head_num=16, head_dim=16, attention_hidden_dim=200, optimizer="adam",
                loss="cross_entropy_loss", dropout=0.2, learning_rate=1e-3, newsencoder_units_per_layer=[512, 512, 512],
                newsencoder_l2_regularization=1e-4)
    base.update(kw)
    return type("hp", (), base)


def oracle_params(hp, seed):
    P = on.init_docvec_params(hp.title_size, hp.newsencoder_units_per_layer, hp.head_num, hp.head_dim, hp.attention_hidden_dim,
                              seed=seed, randomize_bn=True)
    rng = np.random.default_rng(seed + 1)
    for l, u in enumerate(P["units"]):
        P[f"bn{l}_mean"] = 0.1 * rng.standard_normal(u)
        P[f"bn{l}_var"] = 1 + 0.2 * rng.random(u)
    return {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in P.items()}


def weight_list(P):
    out = []
    for l in range(len(P["units"])):
        out += [P[f"d{l}_W"], P[f"d{l}_b"], P[f"bn{l}_g"], P[f"bn{l}_b"], P[f"bn{l}_mean"], P[f"bn{l}_var"]]
    return out + [P["out_W"], P["out_b"], P["u_WQ"], P["u_WK"], P["u_WV"], P["u_W"], P["u_b"], P["u_q"]]


def data(rng, B, H, C, Din):
    his = rng.standard_normal((B, H, Din)).astype(np.float32)
    his[0, :3] = 0  # padded history slots = the zero "unknown" document vector
    pred = rng.standard_normal((B, C, Din)).astype(np.float32)
    y = np.zeros((B, C), np.int8)
    y[np.arange(B), rng.integers(0, C, B)] = 1
    return his, pred, y


@pytest.fixture(scope="module")
def docvec(hip):
    from ebrec.models.newsrec import NRMSDocVec

    return NRMSDocVec


@pytest.mark.parametrize("cfg", [dict(), dict(title_size=40, newsencoder_units_per_layer=[24, 20], head_num=4, head_dim=8, attention_hidden_dim=9, history_size=6)])
def test_forward_eval_mode_uses_moving_statistics(docvec, cfg):
    hp = make_hp(**cfg)
    P = oracle_params(hp, 3)
    m = docvec(hp, seed=1)
    m.model.set_weights(weight_list(P))
    rng = np.random.default_rng(0)
    his, pred, y = data(rng, 6, hp.history_size, 5, hp.title_size)
    probs, s, _ = on.docvec_forward(his.astype(np.float64), pred.astype(np.float64), P, hp.head_num, hp.head_dim, training=False)
    assert_close(m.model.predict((his, pred)), probs, rtol=0, atol=2e-5, what="docvec probabilities")
    one = pred[:, :1]
    _, s1, _ = on.docvec_forward(his.astype(np.float64), one.astype(np.float64), P, hp.head_num, hp.head_dim, training=False)
    assert_close(m.scorer.predict((his, one)), on.sigmoid(s1), rtol=0, atol=2e-5, what="docvec scorer")
    ne = m.newsencoder.predict(pred[0])
    want, _, _ = on.docvec_news_encoder_fwd(pred[0].astype(np.float64), P, training=False)
    assert_close(ne, want, rtol=2e-5, atol=2e-5, what="docvec newsencoder")


@pytest.mark.parametrize("fused", [True, False], ids=["fused_launches", "separate_passes"])
@pytest.mark.parametrize("units,din,B", [([48, 40], 64, 8), ([200, 136, 72], 300, 24), ([36], 20, 3)])
@pytest.mark.parametrize("p,l2", [(0.0, 0.0), (0.2, 1e-4)])
def test_train_step_gradients_loss_and_moving_stats(docvec, p, l2, units, din, B, fused):
    """Both forms of the news encoder's training step -- BatchNormalization / Dropout / ReLU-backward inside the Dense launches
    (csrc/ebn_docvec.hip; partial row tiles, partial and multiple 128-deep slabs, 1-3 hidden layers) and as separate passes."""
    hp = make_hp(title_size=din, newsencoder_units_per_layer=units, head_num=4, head_dim=8, attention_hidden_dim=12,
                 history_size=7, dropout=p, newsencoder_l2_regularization=l2)
    seed = 5
    P = oracle_params(hp, 9)
    m = docvec(hp, seed=seed)
    m._engine.fuse_news_mlp = fused
    m.model.set_weights(weight_list(P))
    rng = np.random.default_rng(2)
    his, pred, y = data(rng, B, hp.history_size, 5, hp.title_size)
    got = float(m.train_step(his, pred, y).item())
    eng = m._engine
    gate = ReluTieGate(eng, B * hp.history_size, B * 5)  # ReLU inputs within rounding of 0: the engine's side (order-independent comparison)
    L, _, g, stats = on.docvec_loss_and_grads(his.astype(np.float64), pred.astype(np.float64), y, P, hp.head_num, hp.head_dim,
                                              l2=l2, training=True, drop=on.Drop(p, seed, 1) if p > 0 else None, relu_gate=gate)
    assert gate.fraction() < 1e-3, (gate.n_ambiguous, gate.n_total)  # (small shapes: one element of 3 x 12 x 36 would already be 8e-4)
    assert abs(got - L) <= 3e-5 * max(1.0, abs(L)), (got, L)
    E = eng.E
    assert (eng._bufs["mlp"].get("dvn_live") is not None) == fused
    nl = len(units)
    for k in [f"d{l}_{s}" for l in range(nl) for s in ("W", "b")] + [f"bn{l}_{s}" for l in range(nl) for s in ("g", "b")] + ["out_W", "out_b", "u_W", "u_b"]:
        want = g[k].reshape(eng.params.shapes[k])
        assert_close(eng.params.g(k).cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what=f"d{k}")
    want = np.concatenate([g["u_WQ"], g["u_WK"], g["u_WV"]], 1)
    assert_close(eng.params.g("u_Wqkv").cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what="du_Wqkv")
    Pn = dict(P)
    on.bn_update_moving(Pn, stats)  # history call site first, then candidates: two updates per step
    for l in range(nl):
        assert_close(eng.bn_mean[l].cpu().numpy(), Pn[f"bn{l}_mean"], rtol=1e-5, atol=1e-6, what=f"moving mean {l}")
        assert_close(eng.bn_var[l].cpu().numpy(), Pn[f"bn{l}_var"], rtol=1e-5, atol=1e-6, what=f"moving var {l}")


def test_docvec_fit_surface_and_weights_roundtrip(docvec, tmp_path):
    hp = make_hp(title_size=32, newsencoder_units_per_layer=[16], head_num=2, head_dim=8, attention_hidden_dim=6, history_size=4)
    m = docvec(hp, seed=3)
    rng = np.random.default_rng(4)
    his, pred, y = data(rng, 40, 4, 5, 32)
    m.model.compile(optimizer=m.model.optimizer, loss=m.model.loss, metrics=["AUC"])
    h = m.model.fit((his, pred), y, batch_size=16, epochs=4, verbose=0, validation_data=((his, pred), y))
    assert h.history["loss"][-1] < h.history["loss"][0] and "val_auc" in h.history
    w = m.model.get_weights()
    assert len(w) == 6 * 1 + 8 and w[4].shape == (16,) and not np.allclose(w[4], 0)  # moving mean moved
    path = tmp_path / "docvec.weights"
    m.model.save_weights(str(path))
    m2 = docvec(hp, seed=77)
    m2.model.load_weights(str(path))
    assert np.array_equal(m2.model.predict((his, pred)), m.model.predict((his, pred)))
    assert m.model.count_params() == sum(a.size for a in w)


def test_docvec_graph_replay_and_indexed_batches_match_eager_host_batches(docvec):
    hp = make_hp(title_size=32, newsencoder_units_per_layer=[24, 16], head_num=2, head_dim=8, attention_hidden_dim=6, history_size=4,
                 learning_rate=1e-3)
    rng = np.random.default_rng(9)
    matrix = rng.standard_normal((50, 32)).astype(np.float32)
    matrix[0] = 0
    ms = [docvec(hp, seed=3) for _ in range(3)]
    ms[1]._engine.enable_graphs()
    ms[2]._engine.enable_graphs()
    ms[2]._engine.set_article_matrix(matrix)
    for step in range(4):
        hi, pi = rng.integers(0, 50, (6, 4)), rng.integers(0, 50, (6, 5))
        y = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 6)]
        l0 = float(ms[0].train_step(matrix[hi], matrix[pi], y).item())
        l1 = float(ms[1].train_step(matrix[hi], matrix[pi], y).item())
        l2 = float(ms[2]._engine.train_step(hi, pi, y, indexed=True).item())
        assert l0 == l1 == l2, (step, l0, l1, l2)
    w = [m.model.get_weights() for m in ms]
    assert all(np.array_equal(a, b) and np.array_equal(a, c) for a, b, c in zip(*w))


def test_docvec_graphs_survive_an_eval_pass_that_grows_the_buffers(docvec):
    """fit(train, validation_data=val) with hipGraphs: the validation pass encodes the whole article matrix, which
    re-allocates the MLP buffers a captured training graph points into -- the graph must be re-captured, not replayed."""
    hp = make_hp(title_size=32, newsencoder_units_per_layer=[24, 16], head_num=2, head_dim=8, attention_hidden_dim=6, history_size=4,
                 learning_rate=1e-3)
    rng = np.random.default_rng(11)
    matrix = rng.standard_normal((400, 32)).astype(np.float32)  # 400 rows > B*(H+C) = 54: encode_news grows the buffers
    eager, graph = docvec(hp, seed=3), docvec(hp, seed=3)
    graph._engine.enable_graphs()
    for step in range(4):
        hi, pi = rng.integers(0, 400, (6, 4)), rng.integers(0, 400, (6, 5))
        y = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 6)]
        l0 = float(eager.train_step(matrix[hi], matrix[pi], y).item())
        l1 = float(graph.train_step(matrix[hi], matrix[pi], y).item())
        assert l0 == l1, (step, l0, l1)
        if step == 1:
            a, b = eager._engine.encode_news(matrix), graph._engine.encode_news(matrix)
            assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    for a, b in zip(eager.model.get_weights(), graph.model.get_weights()):
        assert np.array_equal(a, b)


def test_docvec_out_of_range_article_rows_raise_at_the_epoch_check_and_val_loss_carries_the_l2_term(docvec):
    """Device-resident indexed batches are not range-checked on the host: the gather flags a bad row and check_oob()
    (called by fit / evaluate once per epoch) raises, as TF-CPU's gather would.  evaluate() adds the kernel_regularizer
    penalty to the loss like Keras does, so `val_loss` is comparable to `loss`."""
    import torch

    hp = make_hp(title_size=32, newsencoder_units_per_layer=[24, 16], head_num=2, head_dim=8, attention_hidden_dim=6, history_size=4,
                 dropout=0.0, newsencoder_l2_regularization=1e-2)
    rng = np.random.default_rng(13)
    matrix = rng.standard_normal((50, 32)).astype(np.float32)
    m = docvec(hp, seed=3)
    eng = m._engine
    eng.set_article_matrix(matrix)
    hi = torch.from_numpy(rng.integers(0, 50, (6, 4)).astype(np.int32)).cuda()
    pi = torch.from_numpy(rng.integers(0, 50, (6, 5)).astype(np.int32)).cuda()
    y = torch.from_numpy(np.eye(5, dtype=np.float32)[rng.integers(0, 5, 6)]).cuda()
    eng.train_step(hi, pi, y, indexed=True)
    eng.check_oob()  # clean batch: nothing raised
    hi[0, 0] = 50
    eng.train_step(hi, pi, y, indexed=True)
    with pytest.raises(IndexError):
        eng.check_oob()
    eng.check_oob()  # the flag is cleared by the raise
    his, pred = matrix[rng.integers(0, 50, (6, 4))], matrix[rng.integers(0, 50, (6, 5))]
    yy = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 6)]
    plain = float(eng.eval_loss(his, pred, yy)[0].item())
    w = m.model.get_weights()
    penalty = 1e-2 * (float((w[0].astype(np.float64) ** 2).sum()) + float((w[6].astype(np.float64) ** 2).sum()))  # the two hidden Dense kernels
    got = m.model.evaluate((his, pred), yy, batch_size=6, return_dict=True)["loss"]
    assert abs(got - (plain + penalty)) <= 1e-5 * max(1.0, abs(got)), (got, plain, penalty)


def test_a_step_that_stopped_between_forward_and_backward_does_not_poison_the_next_one(docvec):
    """The fused launches keep their column sums in accumulators that the step itself re-zeroes (backward ones by the first forward launch,
    forward ones by the last backward launch).  A step that ran only its forward -- an exception in between -- leaves them dirty: the engine
    notices (`_dvn_dirty`) and clears the scratch, so the next step is the step a fresh model would take."""
    hp = make_hp(title_size=64, newsencoder_units_per_layer=[48, 40], head_num=4, head_dim=8, attention_hidden_dim=12, history_size=7, dropout=0.0)
    P = oracle_params(hp, 9)
    rng = np.random.default_rng(2)
    his, pred, y = data(rng, 8, hp.history_size, 5, hp.title_size)
    grads = []
    for interrupted in (False, True):
        m = docvec(hp, seed=5)
        m.model.set_weights(weight_list(P))
        eng = m._engine
        if interrupted:  # forward of a training step on other data, no backward
            h2, p2, _ = data(np.random.default_rng(7), 8, hp.history_size, 5, hp.title_size)
            mb = eng._mlp_bufs(8 * (hp.history_size + 5))
            eng._dvn(mb, 8 * hp.history_size, 8 * 5)
            eng._upload(mb, h2, p2)
            eng._news_forward(mb, 8 * hp.history_size, 8 * 5, True)
            assert eng._dvn_dirty
            m.model.set_weights(weight_list(P))  # (the moving statistics the lone forward pass updated)
        eng.train_step(his, pred, y)
        assert not eng._dvn_dirty
        grads.append(eng.params.grad.cpu().numpy().copy())
    assert np.array_equal(grads[0], grads[1])


@pytest.mark.parametrize("units,din,B,graphs", [([48, 40], 64, 8, False), ([512, 512, 512], 768, 32, True), ([36], 20, 3, True)])
def test_the_one_launch_finale_equals_the_three_launches_it_replaces(docvec, units, din, B, graphs):
    """ebn_dvn_finale_f32 (the weight-gradient group with Adam in its epilogue + Adam over every other parameter + the user head's
    finishing sums + the batch loss incl. the L2 term, ONE launch) against tn_group | user_head_finish | adam_keras: the same loss
    bits, the same gradients, the same weights and Adam moments after three steps (dropout on, l2 on), eager and as graph replays;
    64 x 64 tiles of 1024 threads (the c3 shape) and 32 x 32 tiles of 256."""
    import torch

    full = units == [512, 512, 512]
    hp = make_hp(title_size=din, newsencoder_units_per_layer=units, history_size=20 if full else 7, **({} if full else dict(head_num=2, head_dim=16, attention_hidden_dim=12)))  # (head_dim 16 / 20 / 32: the attention core's MFMA path, which the one-launch head needs)
    P = oracle_params(hp, 9)
    rng = np.random.default_rng(2)
    batches = [data(rng, B, hp.history_size, 5, hp.title_size) for _ in range(3)]
    out = []
    for finale in (True, False):
        m = docvec(hp, seed=5)
        m.model.set_weights(weight_list(P))
        eng = m._engine
        eng.fuse_finale = finale
        eng.enable_graphs(graphs)
        losses, g1 = [], None
        for i, bt in enumerate(batches):
            losses.append(float(m.train_step(*bt).item()))
            if i == 0:
                g1 = eng.params.grad.cpu().numpy().copy()
        torch.cuda.synchronize()
        assert eng._step_applied_adam(eng._bufs["mlp"]) == finale
        out.append((losses, g1, eng.params.data.cpu().numpy().copy(), eng.params.m.cpu().numpy().copy(),
                    eng.params.v.cpu().numpy().copy(), [t.cpu().numpy().copy() for t in eng.bn_mean + eng.bn_var], int(eng.read_state().step)))
    a, b = out
    # the loss: same rows, same fixed-order sums, same L2 term -- bit-identical on the first step; later steps start from weights that
    # agree to an ulp (the compiler contracts the two Adam kernels' arithmetic differently), so their losses agree to ~1e-7
    assert a[0][0] == b[0][0] and np.allclose(a[0], b[0], rtol=2e-6, atol=0), (a[0], b[0])
    assert a[6] == b[6] == 3
    assert np.array_equal(a[1], b[1])  # every gradient of the first step (same weights going in): the same bits
    for x, y, what in zip(a[2:5], b[2:5], ("weights", "Adam m", "Adam v")):  # (steps 2 and 3 start from weights an ulp apart: their gradients agree to ~1e-4)
        assert_close(x, y, rtol=1e-4, atol=1e-8 + 2e-4 * np.abs(y).max(), what=f"finale vs separate launches after three steps: {what}")
    for x, y in zip(a[5], b[5]):
        assert_close(x, y, rtol=1e-5, atol=1e-7, what="moving statistics")


def test_fused_column_sums_outside_the_fixed_point_range_raise_instead_of_wrapping(docvec):
    """Round-5 ADVICE: the fused launches keep their BatchNormalization column sums in 64-bit fixed-point accumulators; a tile sum that is
    not finite or could wrap the total is not added -- it raises a sticky flag and check_oob() raises FloatingPointError -- instead of
    silently yielding garbage statistics.  Large but in-range activations (x 30) agree with the separate-pass form; a diverged input
    (x 1e6, then an Inf) raises, and the engine trains normally afterwards (the accumulators are re-zeroed)."""
    hp = make_hp(title_size=64, newsencoder_units_per_layer=[48, 40], head_num=2, head_dim=16, attention_hidden_dim=12, history_size=7, dropout=0.0,
                 newsencoder_l2_regularization=0.0)
    P = oracle_params(hp, 9)
    rng = np.random.default_rng(2)
    his, pred, y = data(rng, 8, hp.history_size, 5, hp.title_size)
    grads = []
    for fused in (True, False):
        m = docvec(hp, seed=5)
        m.model.set_weights(weight_list(P))
        m._engine.fuse_news_mlp = fused
        m.train_step(30.0 * his, 30.0 * pred, y)
        m._engine.check_oob()  # in range: nothing raised
        grads.append(m._engine.params.grad.cpu().numpy().astype(np.float64))
    assert_close(grads[0], grads[1], rtol=2e-3, atol=1e-6 + 2e-4 * np.abs(grads[1]).max(), what="x30 activations, fused vs separate passes")
    m = docvec(hp, seed=5)
    m.model.set_weights(weight_list(P))
    eng = m._engine
    # (an Inf, not a NaN: Dense(relu) is fmaxf(z, 0) in both forms, which returns 0 for a NaN z -- a NaN input never reaches the column sums)
    for bad in (1e6 * his, np.where(np.arange(his.size).reshape(his.shape) == 5, np.inf, his).astype(np.float32)):
        m.train_step(bad, pred, y)
        with pytest.raises(FloatingPointError, match="fixed-point accumulator"):
            eng.check_oob()
        eng.check_oob()  # the flag is cleared by the raise
        m.model.set_weights(weight_list(P))  # (the diverged step has moved / poisoned the weights)
    loss = float(m.train_step(his, pred, y).item())
    eng.check_oob()
    fresh = docvec(hp, seed=5)
    fresh.model.set_weights(weight_list(P))
    assert loss == float(fresh.train_step(his, pred, y).item())  # the accumulators were re-zeroed: the same statistics as a fresh engine's
