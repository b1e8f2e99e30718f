"""Generates tests/golden/nrms_oracle_small.npz and docvec_oracle_small.npz from the float64 oracle.

PARITY UNPINNED: these vectors come from the build's own restatement (oracle/nrms_numpy.py), not from
TensorFlow (not installable here; the reference has no model tests).  They pin the oracle against
regressions and define the fixture FORMAT: anyone with TF 2.12-2.15 can regenerate the same keys from
``NRMSModel(...).model`` (weights in SURVEY.md A.6 order, dropout off) and drop the file in to pin parity.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import nrms_numpy as on  # noqa: E402

rng = np.random.default_rng(2024)
V, D, h, d, A = 40, 12, 3, 4, 7
B, H, C, T = 3, 4, 3, 5
P = on.random_nrms_params(V, D, h, d, A, seed=1)
his = rng.integers(0, V, (B, H, T))
his[0, 0] = 0  # a padded history slot (token-0 title)
pred = rng.integers(0, V, (B, C, T))
y = np.zeros((B, C))
y[np.arange(B), rng.integers(0, C, B)] = 1
out = {"his": his, "pred": pred, "y": y, "dims": np.array([V, D, h, d, A, B, H, C, T])}
for k in on.PARAM_ORDER:
    out[f"w_{k}"] = P[k]
probs, scores, cache = on.nrms_forward(his, pred, P, h, d)
out["probs"], out["scores"] = probs, scores
out["scorer"] = on.scorer_forward(his, pred[:, :1], P, h, d)
ne, c = on.news_encoder_fwd(pred.reshape(-1, T), P, h, d)
out["news_encoding_of_pred"] = ne
out["self_attention_out"] = c[5]  # O before dropout, (N,T,E)
out["att_weights"] = c[4][4]
for loss in ("cross_entropy_loss", "log_loss"):
    L, _, g = on.nrms_loss_and_grads(his, pred, y, P, h, d, loss)
    out[f"{loss}_value"] = np.array(L)
    for k in on.PARAM_ORDER:
        out[f"{loss}_grad_{k}"] = g[k]
L, _, g = on.nrms_loss_and_grads(his, pred, y, P, h, d, "cross_entropy_loss", on.Drop(0.2, 42, 3))
out["dropout_loss_p0.2_seed42_step3"] = np.array(L)
out["dropout_grad_n_WQ"] = g["n_WQ"]
out["dropout_keep_site0_first64"] = on.dropout_keep_mask(on.dropout_key(42, 3, 0), 64, 0.2)
# Keras-form Adam, 3 steps on one tensor
th, m, v = P["n_W"].copy(), np.zeros_like(P["n_W"]), np.zeros_like(P["n_W"])
for t in range(1, 4):
    on.adam_keras_step(th, g["n_W"] * t, m, v, t, lr=1e-3)
out["adam_theta_after3"], out["adam_m_after3"], out["adam_v_after3"] = th, m, v
np.savez_compressed(Path(__file__).with_name("nrms_oracle_small.npz"), **out)

# NRMSDocVec
Din, units = 10, [8, 6]
Pd = on.init_docvec_params(Din, units, h, d, A, seed=3, randomize_bn=True)
hv, pv = rng.standard_normal((B, H, Din)), rng.standard_normal((B, C, Din))
dv = {"his": hv, "pred": pv, "y": y, "dims": np.array([Din, *units, h, d, A])}
for k, val in Pd.items():
    if k != "units":
        dv[f"w_{k}"] = val
p_eval, s_eval, _ = on.docvec_forward(hv, pv, Pd, h, d, training=False)
dv["probs_eval"], dv["scores_eval"] = p_eval, s_eval
L, p_tr, g, stats = on.docvec_loss_and_grads(hv, pv, y, Pd, h, d, l2=1e-4, training=True, drop=on.Drop(0.2, 7, 1))
dv["train_loss"], dv["probs_train"] = np.array(L), p_tr
for k, val in g.items():
    dv[f"grad_{k}"] = val
Pn = dict(Pd)
on.bn_update_moving(Pn, stats)
for l in range(len(units)):
    dv[f"moving_mean_after_{l}"], dv[f"moving_var_after_{l}"] = Pn[f"bn{l}_mean"], Pn[f"bn{l}_var"]
np.savez_compressed(Path(__file__).with_name("docvec_oracle_small.npz"), **dv)
print("written", sorted(out)[:5], "...", len(out), "arrays;", len(dv), "docvec arrays")
