"""Parity against a golden file produced by the REAL TensorFlow reference (tools/dump_tf_golden.py).
The file cannot be generated in the build container (no TensorFlow), so these tests skip until someone drops
tests/golden/nrms_tf_golden.npz in; they are the hook that turns "parity unpinned" into pinned.  One dump pins the
forward pass, both compiled losses (incl. the BCE-on-logits-vs-clipped-probabilities question of SURVEY A.5), Adam's
update form over three steps, the Keras weight order of all three model variants and NRMSDocVec's per-call-site
BatchNorm with its moving statistics."""
from pathlib import Path

import numpy as np
import pytest

from oracle import nrms_numpy as on

GOLD = Path(__file__).parent / "golden" / "nrms_tf_golden.npz"
needs_gold = pytest.mark.skipif(not GOLD.exists(), reason="no TF golden file (run tools/dump_tf_golden.py where TensorFlow exists)")
TOL = 1e-4  # BASELINE.json north_star: forward scores to 1e-4 fp32


def _load():
    z = np.load(GOLD, allow_pickle=False)
    V, D, h, d, A = (int(x) for x in z["dims"])
    P = {k: z[f"w{i:02d}"].astype(np.float64) for i, k in enumerate(on.PARAM_ORDER)}
    return z, P, h, d


def _docvec_params(z, prefix="docvec_w"):
    """Map the dumped Keras variables to the oracle's names BY NAME (dense*/batch_normalization*/kernel/bias/gamma/...)."""
    names = [str(n) for n in z["docvec_weight_names"]]
    arrs = [z[f"{prefix}{i:02d}"].astype(np.float64) for i in range(len(names))]
    units = [int(u) for u in z["docvec_units"]]
    P = {"units": units}
    dense = [(n, a) for n, a in zip(names, arrs) if "kernel" in n and a.ndim == 2 and "dense" in n.lower()]
    # creation order: hidden Dense layers, then the output Dense; BatchNorm variables per hidden layer
    kernels = [a for n, a in zip(names, arrs) if n.endswith("kernel:0") and "dense" in n.lower()]
    biases = [a for n, a in zip(names, arrs) if n.endswith("bias:0") and "dense" in n.lower()]
    gam = [a for n, a in zip(names, arrs) if "gamma" in n]
    bet = [a for n, a in zip(names, arrs) if "beta" in n]
    mme = [a for n, a in zip(names, arrs) if "moving_mean" in n]
    mva = [a for n, a in zip(names, arrs) if "moving_variance" in n]
    assert len(kernels) == len(units) + 1 and len(gam) == len(units), (names, len(dense))
    for l in range(len(units)):
        P[f"d{l}_W"], P[f"d{l}_b"], P[f"bn{l}_g"], P[f"bn{l}_b"] = kernels[l], biases[l], gam[l], bet[l]
        P[f"bn{l}_mean"], P[f"bn{l}_var"] = mme[l], mva[l]
    P["out_W"], P["out_b"] = kernels[-1], biases[-1]
    rest = [a for n, a in zip(names, arrs) if "dense" not in n.lower() and "batch_normalization" not in n.lower()]
    assert len(rest) == 6, names  # WQ, WK, WV, W, b, q of the user encoder
    P["u_WQ"], P["u_WK"], P["u_WV"], P["u_W"], P["u_b"], P["u_q"] = rest
    return P


@needs_gold
def test_oracle_matches_tensorflow_forward():
    z, P, h, d = _load()
    probs, _, _ = on.nrms_forward(z["his"], z["pred"], P, h, d)
    np.testing.assert_allclose(probs, z["probs"], atol=TOL, rtol=0)
    np.testing.assert_allclose(on.scorer_forward(z["his"], z["pred"][:, :1], P, h, d), z["scorer"], atol=TOL, rtol=0)
    ne, _ = on.news_encoder_fwd(z["pred"][0], P, h, d)
    np.testing.assert_allclose(ne, z["newsencoder"], atol=TOL, rtol=0)


DEFAULT_BCE_ON = "logits"  # must equal NRMSEngine's default `bce_on`; the test below says when to flip both


def _bce_reading(z, P, h, d):
    """Which of the two implemented readings of `log_loss` (nrms.py:54,61-62) TensorFlow's evaluate() agrees with:
    "logits" (sigmoid-CE on the softmax's cached logits) or "probs" (BCE on the clipped softmax outputs)."""
    _, s, _ = on.nrms_forward(z["his"], z["pred"], P, h, d)
    want = float(np.ravel(z["loss_log_loss"])[0])
    err = {"logits": abs(on.loss_fwd_bwd(s, z["y"], "log_loss")[0] - want), "probs": abs(on.loss_fwd_bwd(s, z["y"], "log_loss_probs")[0] - want)}
    return min(err, key=err.get), err


@needs_gold
def test_tensorflow_decides_the_log_loss_reading_and_the_default_follows_it():
    """Both readings are implemented end to end (oracle kind "log_loss" / "log_loss_probs", C-ABI loss_kind 1 / 2, model
    argument bce_on).  Exactly one must match TF; if it is not the default, flip `bce_on` in _engine.py and DEFAULT_BCE_ON."""
    z, P, h, d = _load()
    if "loss_log_loss" not in z.files:
        pytest.skip("golden file predates the widened dump")
    which, err = _bce_reading(z, P, h, d)
    assert err[which] < TOL and max(err.values()) > TOL, err
    assert which == DEFAULT_BCE_ON, f"TensorFlow runs log_loss on the {which}: make bce_on='{which}' the default"


@needs_gold
@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss"])
def test_oracle_matches_tensorflow_losses_and_three_adam_steps(loss):
    """Pins every [KERAS-SEMANTICS] choice of training: the compiled loss (log_loss is where SURVEY A.5 and the oracle
    disagree on paper -- the reading TF agrees with is used from here on), the gradient of the batch mean, Adam's
    sqrt(v)+eps placement and the dense decay of untouched rows."""
    z, P, h, d = _load()
    if f"loss_{loss}" not in z.files:
        pytest.skip("golden file predates the widened dump")
    if loss == "log_loss" and _bce_reading(z, P, h, d)[0] == "probs":
        loss_kind = "log_loss_probs"
    else:
        loss_kind = loss
    _, s, _ = on.nrms_forward(z["his"], z["pred"], P, h, d)
    L, _ = on.loss_fwd_bwd(s, z["y"], loss_kind)
    assert abs(L - float(np.ravel(z[f"loss_{loss}"])[0])) < TOL
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    lr = float(z["learning_rate"])
    for t in range(1, 4):
        Lt, _, g = on.nrms_loss_and_grads(z["his"], z["pred"], z["y"], P, h, d, loss_kind, None)
        assert abs(Lt - float(z[f"train3_losses_{loss}"][t - 1])) < TOL, (t, Lt)
        for k in P:
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=lr)
    for i, k in enumerate(on.PARAM_ORDER):
        want = z[f"train3_{loss}_w{i:02d}"]
        np.testing.assert_allclose(P[k].reshape(want.shape), want, atol=3 * lr * 0.02 + 1e-6, rtol=0, err_msg=k)  # 3 steps of <= lr each


@needs_gold
def test_keras_weight_orders_are_what_from_keras_weight_list_assumes():
    z, _, _, _ = _load()
    if "weight_names" not in z.files:
        pytest.skip("golden file predates the widened dump")
    names = [str(n) for n in z["weight_names"]]
    assert len(names) == 13 and "embedding" in names[0].lower()
    shapes = [z[f"w{i:02d}"].shape for i in range(13)]
    V, D, h, d, A = (int(x) for x in z["dims"])
    E = h * d
    assert shapes == [(V, D), (D, E), (D, E), (D, E), (E, A), (A,), (A, 1), (E, E), (E, E), (E, E), (E, A), (A,), (A, 1)]
    un = [str(n) for n in z["units_weight_names"]]
    assert len(un) == 13 + 6  # one Dense + BatchNormalization block between the news attention and AttLayer2
    kinds = [("kernel" in n, "gamma" in n, "moving" in n) for n in un[4:10]]
    assert kinds[0][0] and kinds[2][1] and kinds[4][2] and kinds[5][2], un


@needs_gold
def test_oracle_matches_tensorflow_docvec_forward_and_one_training_step():
    """Per-call-site batch statistics, two moving-average updates per step, L2 on the hidden kernels only."""
    z, _, _, _ = _load()
    if "docvec_probs" not in z.files:
        pytest.skip("golden file predates the widened dump")
    Din, h, d, A, H = (int(x) for x in z["docvec_dims"])
    P = _docvec_params(z)
    his, pred = z["docvec_his"].astype(np.float64), z["docvec_pred"].astype(np.float64)
    probs, _, _ = on.docvec_forward(his, pred, P, h, d, training=False)
    np.testing.assert_allclose(probs, z["docvec_probs"], atol=TOL, rtol=0)
    L, _, g, stats = on.docvec_loss_and_grads(his, pred, z["y"], P, h, d, l2=float(z["docvec_l2"]), training=True, drop=None)
    assert abs(L - float(z["docvec_train1_loss"])) < TOL
    after = _docvec_params(z, "docvec_train1_w")
    Pn = dict(P)
    on.bn_update_moving(Pn, stats)
    for l in range(len(P["units"])):
        np.testing.assert_allclose(Pn[f"bn{l}_mean"], after[f"bn{l}_mean"], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(Pn[f"bn{l}_var"], after[f"bn{l}_var"], atol=1e-5, rtol=1e-5)
    lr = float(z["docvec_learning_rate"])
    for k in g:
        m, v = np.zeros_like(P[k]), np.zeros_like(P[k])
        th = P[k].copy()
        on.adam_keras_step(th, g[k].reshape(th.shape), m, v, 1, lr=lr)
        np.testing.assert_allclose(th, after[k].reshape(th.shape), atol=lr * 0.02 + 1e-6, rtol=0, err_msg=k)


@needs_gold
def test_oracle_two_replica_docvec_step_follows_tensorflows_batchnorm_contract():
    """Round 5: the data-parallel BatchNormalization contract (per-replica batch statistics, moving statistics = the MEAN over
    the replicas' updates, gradients averaged) against a tf.distribute.MirroredStrategy step on two logical devices."""
    z, _, _, _ = _load()
    if "docvec_dp2_loss" not in z.files:
        pytest.skip("golden file has no two-replica NRMSDocVec step")
    P = _docvec_params(z, "docvec_w")
    _, h, d, _, _ = (int(v) for v in z["docvec_dims"])
    his, pred, y = z["docvec_dp2_his"].astype(np.float64), z["docvec_dp2_pred"].astype(np.float64), z["docvec_dp2_y"]
    B = his.shape[0] // 2
    after = _docvec_params(z, "docvec_dp2_w")
    losses, moved = [], []
    for r in range(2):
        sl = slice(r * B, (r + 1) * B)
        L, _, g, stats = on.docvec_loss_and_grads(his[sl], pred[sl], y[sl], P, h, d, l2=float(z["docvec_l2"]), training=True, drop=None)
        Pn = dict(P)
        on.bn_update_moving(Pn, stats)
        losses.append(L)
        moved.append(Pn)
    assert abs(np.mean(losses) - float(z["docvec_dp2_loss"])) < TOL
    for l in range(len(P["units"])):
        for nm in ("mean", "var"):
            want = after[f"bn{l}_{nm}"]
            got = (moved[0][f"bn{l}_{nm}"] + moved[1][f"bn{l}_{nm}"]) / 2
            np.testing.assert_allclose(got, want, atol=1e-5, rtol=1e-5, err_msg=f"bn{l}_{nm}: mean over the replicas")


@needs_gold
@pytest.mark.gpu
def test_hip_path_matches_tensorflow_forward(hip):
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.model_config import hparams_nrms

    z, P, h, d = _load()
    m = NRMSModel(hparams_nrms, word2vec_embedding=P["emb"]).from_keras_weight_list([z[f"w{i:02d}"] for i in range(13)])
    np.testing.assert_allclose(m.model.predict((z["his"], z["pred"])), z["probs"], atol=TOL, rtol=0)
    np.testing.assert_allclose(m.scorer.predict((z["his"], z["pred"][:, :1])), z["scorer"], atol=TOL, rtol=0)
    np.testing.assert_allclose(m.userencoder.predict(z["his"]), z["userencoder"], atol=TOL, rtol=0)


@needs_gold
@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss"])
def test_hip_path_matches_tensorflow_training_steps(hip, loss):
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.model_config import hparams_nrms

    z, P, h, d = _load()
    if f"train3_losses_{loss}" not in z.files:
        pytest.skip("golden file predates the widened dump")
    hp = type("hp", (hparams_nrms,), dict(loss=loss, dropout=0.0, learning_rate=float(z["learning_rate"])))
    bce_on = _bce_reading(z, P, h, d)[0] if "loss_log_loss" in z.files else DEFAULT_BCE_ON  # the reading TF agrees with
    m = NRMSModel(hp, word2vec_embedding=P["emb"], seed=42, bce_on=bce_on).from_keras_weight_list([z[f"w{i:02d}"] for i in range(13)])
    for t in range(3):
        got = float(m.train_step(z["his"], z["pred"], z["y"]).item())
        assert abs(got - float(z[f"train3_losses_{loss}"][t])) < TOL, (t, got)
    lr = float(z["learning_rate"])
    for i, w in enumerate(m.model.get_weights()):
        want = z[f"train3_{loss}_w{i:02d}"]
        np.testing.assert_allclose(w.reshape(want.shape), want, atol=3 * lr * 0.02 + 1e-6, rtol=0)
