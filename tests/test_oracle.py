"""CPU tests of the oracle (oracle/nrms_numpy.py): committed golden fixtures, finite differences, the
independent torch-autograd restatement, and one assertion per reference quirk."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import nrms_numpy as on
from oracle import nrms_torch as ot

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def g():
    z = np.load(GOLD / "nrms_oracle_small.npz")
    V, D, h, d, A, B, H, C, T = (int(x) for x in z["dims"])
    P = {k: z[f"w_{k}"] for k in on.PARAM_ORDER}
    return z, P, (V, D, h, d, A, B, H, C, T)


def test_forward_reproduces_golden(g):
    z, P, (V, D, h, d, A, B, H, C, T) = g
    probs, scores, _ = on.nrms_forward(z["his"], z["pred"], P, h, d)
    np.testing.assert_allclose(probs, z["probs"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(scores, z["scores"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(on.scorer_forward(z["his"], z["pred"][:, :1], P, h, d), z["scorer"], rtol=1e-12)
    ne, c = on.news_encoder_fwd(z["pred"].reshape(-1, T), P, h, d)
    np.testing.assert_allclose(ne, z["news_encoding_of_pred"], rtol=1e-12)
    np.testing.assert_allclose(c[4][4], z["att_weights"], rtol=1e-12)
    assert np.allclose(probs.sum(1), 1)


@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss"])
def test_loss_and_gradients_reproduce_golden(g, loss):
    z, P, (V, D, h, d, A, B, H, C, T) = g
    L, _, grads = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss)
    assert L == pytest.approx(float(z[f"{loss}_value"]), rel=1e-12)
    for k in on.PARAM_ORDER:
        np.testing.assert_allclose(grads[k], z[f"{loss}_grad_{k}"], rtol=1e-10, atol=1e-14, err_msg=k)


def test_dropout_stream_and_adam_reproduce_golden(g):
    z, P, (V, D, h, d, A, B, H, C, T) = g
    L, _, grads = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, "cross_entropy_loss", on.Drop(0.2, 42, 3))
    assert L == pytest.approx(float(z["dropout_loss_p0.2_seed42_step3"]), rel=1e-12)
    np.testing.assert_allclose(grads["n_WQ"], z["dropout_grad_n_WQ"], rtol=1e-10, atol=1e-14)
    assert np.array_equal(on.dropout_keep_mask(on.dropout_key(42, 3, 0), 64, 0.2), z["dropout_keep_site0_first64"])
    th, m, v = P["n_W"].copy(), np.zeros_like(P["n_W"]), np.zeros_like(P["n_W"])
    for t in range(1, 4):
        on.adam_keras_step(th, grads["n_W"] * t, m, v, t, lr=1e-3)
    np.testing.assert_allclose(th, z["adam_theta_after3"], rtol=1e-12)
    np.testing.assert_allclose(v, z["adam_v_after3"], rtol=1e-12)


def test_known_hash_values_pin_the_dropout_stream():
    """Fixed points of the counter-based stream shared with csrc/ebn_common.h."""
    assert [int(x) for x in on.lowbias32(np.arange(4, dtype=np.uint32))] == [0, 1753845952, 3507691905, 1408362973]
    assert on.dropout_key(42, 3, 0) == 1591691161 and on.dropout_key(0, 1, 1) == 1901086856
    assert on.dropout_threshold(0.2) == 13107 and on.dropout_threshold(0.0) == 0 and on.dropout_threshold(1.0) == 65535
    k = on.dropout_key(42, 3, 0)
    assert k == on.dropout_key(42, 3, 0) and k != on.dropout_key(42, 3, 1) and k != on.dropout_key(42, 4, 0)
    keep = on.dropout_keep_mask(k, 200000, 0.2)
    assert abs(keep.mean() - 0.8) < 0.005
    big = on.dropout_keep_mask(k, 8, 0.5, start=2 ** 33 - 4)  # crosses the 32-bit pair-index boundary
    assert big.shape == (8,)


@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss"])
def test_backward_against_finite_differences(g, loss):
    z, P, (V, D, h, d, A, B, H, C, T) = g
    P = {k: v.copy() for k, v in P.items()}
    drop = on.Drop(0.25, 5, 2)
    L0, _, grads = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss, drop)
    rng = np.random.default_rng(0)
    eps = 1e-6
    for k in on.PARAM_ORDER:
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in P[k].shape)
            if k == "emb":
                idx = (int(z["his"].reshape(-1)[rng.integers(0, z["his"].size)]), idx[1])  # a row that is used
            P[k][idx] += eps
            Lp, _, _ = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss, drop)
            P[k][idx] -= 2 * eps
            Lm, _, _ = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss, drop)
            P[k][idx] += eps
            fd = (Lp - Lm) / (2 * eps)
            assert fd == pytest.approx(grads[k][idx], rel=2e-5, abs=1e-9), (k, idx)


def test_torch_autograd_restatement_agrees(g):
    z, P, (V, D, h, d, A, B, H, C, T) = g
    for loss in ("cross_entropy_loss", "log_loss"):
        L, _, grads = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss)
        Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
        s = ot.nrms_scores(torch.tensor(z["his"]), torch.tensor(z["pred"]), Pt, h, d)
        Lt = ot.loss_from_scores(s, torch.tensor(z["y"]), loss)
        Lt.backward()
        assert float(Lt.detach()) == pytest.approx(L, rel=1e-12)
        for k in on.PARAM_ORDER:
            np.testing.assert_allclose(Pt[k].grad.numpy(), grads[k], rtol=1e-9, atol=1e-13, err_msg=k)


# ---------------- one assertion per quirk -----------------------------------------
def test_quirk_attention_multiplies_by_the_transposed_matrix():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((2, 5, 6))
    WQ, WK, WV = (rng.standard_normal((6, 8)) for _ in range(3))
    O, cache = on.self_attention_fwd(X, WQ, WK, WV, 2, 4)
    Pm, Vh = cache[7], cache[6]
    want = (Pm.transpose(0, 1, 3, 2) @ Vh).transpose(0, 2, 1, 3).reshape(2, 5, 8)  # layers.py:249 adjoint_a=True
    std = (Pm @ Vh).transpose(0, 2, 1, 3).reshape(2, 5, 8)
    np.testing.assert_allclose(O, want)
    assert np.abs(O - std).max() > 1e-2
    np.testing.assert_allclose(Pm.sum(-1), 1.0)  # softmax is over the key axis (layers.py:247)


def test_quirk_no_bias_no_output_projection_no_mask():
    rng = np.random.default_rng(2)
    V = 30
    P = on.random_nrms_params(V, 8, 2, 4, 5, seed=0)
    ids = rng.integers(1, V, (3, 6))
    ids_pad = ids.copy()
    ids_pad[:, 4:] = 0  # "padding" tokens are ordinary rows: output changes with the content of row 0
    a, _ = on.news_encoder_fwd(ids_pad, P, 2, 4)
    P2 = dict(P)
    P2["emb"] = P["emb"].copy()
    P2["emb"][0] += 1.0
    b, _ = on.news_encoder_fwd(ids_pad, P2, 2, 4)
    assert np.abs(a - b).max() > 1e-3
    zero = {k: (np.zeros_like(v) if k == "emb" else v) for k, v in P.items()}
    O, _ = on.self_attention_fwd(zero["emb"][ids], P["n_WQ"], P["n_WK"], P["n_WV"], 2, 4)
    assert np.all(O == 0)  # no bias anywhere in SelfAttention (layers.py:155-172)


def test_quirk_additive_attention_is_unstabilised_with_epsilon():
    X = np.ones((1, 3, 4))
    W = np.zeros((4, 2))
    b = np.full(2, -30.0)  # tanh -> -1
    q = np.full((2, 1), 10.0)  # e = -20
    out, cache = on.att_layer2_fwd(X, W, b, q)
    a = np.exp(-20.0)
    assert cache[4][0, 0] == pytest.approx(a / (3 * a + 1e-7), rel=1e-12)  # layers.py:75-77
    assert cache[4].sum() < 0.2  # a stabilised softmax would sum to 1
    # and it overflows where a stabilised one would not (layers.py:71 has no max-subtraction)
    with np.errstate(over="ignore", invalid="ignore"):
        big, _ = on.att_layer2_fwd(X, W, np.full(2, 30.0), np.full((2, 1), 400.0))
    assert not np.isfinite(big).all()


def test_quirk_seeded_init_shares_one_draw_for_q_k_v():
    P = on.init_nrms_params(20, 6, 2, 3, 4, seed=7)
    assert np.array_equal(P["n_WQ"], P["n_WK"]) and np.array_equal(P["n_WK"], P["n_WV"])
    assert np.array_equal(P["u_WQ"], P["u_WV"]) and np.all(P["n_b"] == 0)
    assert [k for k in on.PARAM_ORDER] == ["emb", "n_WQ", "n_WK", "n_WV", "n_W", "n_b", "n_q", "u_WQ", "u_WK", "u_WV", "u_W", "u_b", "u_q"]


def test_quirk_keras_adam_form_and_dense_decay():
    th, m, v = np.ones(3), np.zeros(3), np.zeros(3)
    g = np.array([0.5, 0.0, -2.0])
    on.adam_keras_step(th, g, m, v, 1, lr=0.1)
    alpha = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(m, 0.1 * g)
    np.testing.assert_allclose(v, 0.001 * g * g)
    np.testing.assert_allclose(th, 1 - alpha * m / (np.sqrt(v) + 1e-7))  # eps next to sqrt(v), before bias correction
    m0, v0 = m.copy(), v.copy()
    on.adam_keras_step(th, np.zeros(3), m, v, 2, lr=0.1)  # a row with zero gradient still decays and moves
    np.testing.assert_allclose(m, 0.9 * m0)
    np.testing.assert_allclose(v, 0.999 * v0)


def test_losses_follow_keras_logits_path():
    s = np.array([[2.0, -1.0, 0.5]])
    y = np.array([[0, 1, 0]])
    L, ds = on.loss_fwd_bwd(s, y, "cross_entropy_loss")
    assert L == pytest.approx(np.log(np.exp(s).sum()) + 1.0)
    np.testing.assert_allclose(ds, on.softmax_rows(s) - y)
    L2, ds2 = on.loss_fwd_bwd(s, y, "log_loss")
    assert L2 == pytest.approx(np.mean(np.log1p(np.exp(s)) - s * y))
    np.testing.assert_allclose(ds2, (on.sigmoid(s) - y) / 3)
    with pytest.raises(ValueError):
        on.loss_fwd_bwd(s, y, "mse")


def test_log_loss_on_clipped_probabilities_is_the_other_keras_reading():
    """binary_crossentropy on the softmax OUTPUTS (clip to [1e-7, 1-1e-7], +1e-7 inside the logs) -- kept next
    to the logits reading until the TF dump decides.  Value by hand, gradient against torch autograd and finite differences,
    and the clip passes no gradient where it is active."""
    import torch
    from oracle import nrms_torch as ot

    rng = np.random.default_rng(3)
    s = rng.standard_normal((6, 5)) * 2
    y = np.eye(5)[rng.integers(0, 5, 6)]
    L, ds = on.loss_fwd_bwd(s, y, "log_loss_probs")
    p = on.softmax_rows(s)
    assert L == pytest.approx(float(-(y * np.log(p + 1e-7) + (1 - y) * np.log(1 - p + 1e-7)).mean()), rel=1e-12)
    ts = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    Lt = ot.loss_from_scores(ts, torch.tensor(y), "log_loss_probs")
    Lt.backward()
    assert float(Lt) == pytest.approx(L, rel=1e-12)
    np.testing.assert_allclose(ds, ts.grad.numpy(), rtol=1e-9, atol=1e-14)
    num = np.zeros_like(s)
    for i in np.ndindex(*s.shape):
        sp, sm = s.copy(), s.copy()
        sp[i] += 1e-6
        sm[i] -= 1e-6
        num[i] = (on.loss_fwd_bwd(sp, y, "log_loss_probs")[0] - on.loss_fwd_bwd(sm, y, "log_loss_probs")[0]) / 2e-6
    np.testing.assert_allclose(ds, num, rtol=1e-6, atol=1e-9)
    # the two readings are different functions (this is what the TF dump decides between)
    assert abs(L - on.loss_fwd_bwd(s, y, "log_loss")[0]) > 1e-2
    # saturated row: p = (1, 0, 0) to within 1e-7 -> every element sits on the clip, the gradient vanishes
    s_sat = np.array([[60.0, 0.0, 0.0]])
    L_sat, ds_sat = on.loss_fwd_bwd(s_sat, np.array([[0.0, 1.0, 0.0]]), "log_loss_probs")
    assert np.all(ds_sat == 0) and L_sat == pytest.approx(-(np.log(2e-7) + np.log(2e-7) + np.log(1.0)) / 3, rel=1e-6)


# ---------------- NRMSDocVec ---------------------------------------------------------------------------
def test_docvec_golden_and_batchnorm_call_site_statistics():
    z = np.load(GOLD / "docvec_oracle_small.npz")
    dims = [int(x) for x in z["dims"]]
    Din, units, (h, d, A) = dims[0], dims[1:-3], dims[-3:]
    P = {k[2:]: z[k] for k in z.files if k.startswith("w_")}
    P["units"] = units
    p, s, _ = on.docvec_forward(z["his"], z["pred"], P, h, d, training=False)
    np.testing.assert_allclose(p, z["probs_eval"], rtol=1e-12)
    L, p_tr, g, stats = on.docvec_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, l2=1e-4, training=True,
                                                 drop=on.Drop(0.2, 7, 1))
    assert L == pytest.approx(float(z["train_loss"]), rel=1e-12)
    for k in g:
        np.testing.assert_allclose(g[k], z[f"grad_{k}"], rtol=1e-10, atol=1e-14, err_msg=k)
    # history and candidate call sites use their OWN batch statistics; two moving-average updates per step
    st_h, st_c = stats
    assert not np.allclose(st_h[0][0], st_c[0][0])
    Pn = dict(P)
    on.bn_update_moving(Pn, stats)
    np.testing.assert_allclose(Pn["bn0_mean"], z["moving_mean_after_0"], rtol=1e-12)
    want = (P["bn0_mean"] * 0.99 + st_h[0][0] * 0.01) * 0.99 + st_c[0][0] * 0.01
    np.testing.assert_allclose(Pn["bn0_mean"], want, rtol=1e-12)


def test_docvec_backward_against_finite_differences():
    rng = np.random.default_rng(4)
    h, d, A = 2, 3, 4
    P = on.init_docvec_params(7, [6, 5], h, d, A, seed=1, randomize_bn=True)
    his, pred = rng.standard_normal((3, 4, 7)), rng.standard_normal((3, 2, 7))
    y = np.eye(2)[rng.integers(0, 2, 3)]
    drop = on.Drop(0.2, 3, 1)
    L, _, g, _ = on.docvec_loss_and_grads(his, pred, y, P, h, d, l2=1e-3, training=True, drop=drop)
    eps = 1e-6
    for k in ("d0_W", "d0_b", "bn0_g", "bn1_b", "d1_W", "out_W", "out_b", "u_WK", "u_q"):
        idx = tuple(rng.integers(0, s) for s in P[k].shape)
        P[k][idx] += eps
        Lp = on.docvec_loss_and_grads(his, pred, y, P, h, d, l2=1e-3, training=True, drop=drop)[0]
        P[k][idx] -= 2 * eps
        Lm = on.docvec_loss_and_grads(his, pred, y, P, h, d, l2=1e-3, training=True, drop=drop)[0]
        P[k][idx] += eps
        assert (Lp - Lm) / (2 * eps) == pytest.approx(g[k][idx], rel=5e-5, abs=1e-9), k


def test_torch_docvec_port_agrees_with_the_numpy_oracle():
    """oracle/nrms_torch.py:CpuDocVecTrainer (autograd; the cpu_baseline port of configs[2]) against the hand-derived float64
    restatement: loss, every gradient, and the batch statistics both call sites feed into the moving averages."""
    import torch

    from oracle import nrms_numpy as on
    from oracle.nrms_torch import CpuDocVecTrainer

    units, h, d, A, Din, B, H, C, l2 = [24, 16], 2, 4, 6, 12, 5, 3, 4, 1e-3
    P = on.init_docvec_params(Din, units, h, d, A, seed=3, randomize_bn=True)
    P = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in P.items()}  # fp32-exact values
    rng = np.random.default_rng(0)
    his, pred = rng.standard_normal((B, H, Din)), rng.standard_normal((B, C, Din))
    y = np.zeros((B, C)); y[np.arange(B), rng.integers(0, C, B)] = 1
    L, _, g, (st_h, st_c) = on.docvec_loss_and_grads(his, pred, y, P, h, d, l2=l2, training=True, drop=None)
    tr = CpuDocVecTrainer(P, units, h, d, dropout=0.0, l2=l2)
    tr.P = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in tr.P.items()}
    Lt, gt, stats = tr.loss_and_grads(torch.tensor(his), torch.tensor(pred), torch.tensor(y))
    assert float(Lt) == pytest.approx(L, rel=1e-10)
    for k in ("d0_W", "d1_b", "bn0_g", "bn1_b", "out_W", "out_b", "u_WQ", "u_W", "u_q"):
        np.testing.assert_allclose(gt[k].numpy().reshape(g[k].shape), g[k], rtol=1e-8, atol=1e-10)
    for (l, mu, var), (mu_o, var_o) in zip(stats, list(st_h) + list(st_c)):
        np.testing.assert_allclose(mu.numpy(), mu_o, rtol=1e-10)
        np.testing.assert_allclose(var.numpy(), var_o, rtol=1e-10)


def test_relu_gate_hook_only_changes_what_it_is_told_to():
    """oracle._relu_gate (the parity tests' order-independence hook): None / the oracle's own [pre > 0] change nothing; the hook sees every
    ReLU of both call sites (hidden layers 0..L-1 and the output Dense L), history site first."""
    rng = np.random.default_rng(0)
    P = on.init_docvec_params(20, [12, 8], 2, 4, 6, seed=1)
    his, pred = rng.standard_normal((3, 4, 20)), rng.standard_normal((3, 5, 20))
    y = np.eye(5)[rng.integers(0, 5, 3)]
    a = on.docvec_loss_and_grads(his, pred, y, P, 2, 4, l2=1e-3)
    seen = []

    def same(site, layer, pre):
        seen.append((site, layer, pre.shape))
        return pre > 0

    b = on.docvec_loss_and_grads(his, pred, y, P, 2, 4, l2=1e-3, relu_gate=same)
    c = on.docvec_loss_and_grads(his, pred, y, P, 2, 4, l2=1e-3, relu_gate=lambda s, l, pre: None)
    for other in (b, c):
        assert a[0] == other[0] and all(np.array_equal(a[2][k], other[2][k]) for k in a[2])
    assert seen == [(0, 2, (12, 8)), (0, 1, (12, 8)), (0, 0, (12, 12)), (1, 2, (15, 8)), (1, 1, (15, 8)), (1, 0, (15, 12))]
    d = on.docvec_loss_and_grads(his, pred, y, P, 2, 4, l2=1e-3, relu_gate=lambda s, l, pre: np.ones_like(pre, bool))
    assert max(np.abs(a[2][k] - d[2][k]).max() for k in a[2]) > 1e-3  # an all-pass gate IS a different function
