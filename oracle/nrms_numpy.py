"""ORACLE -- test infrastructure, NOT product code.

A numpy (float64 by default) restatement of the reference's NRMS / NRMSDocVec
math, written from SURVEY.md Appendix A.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``ebnerd-benchmark_amd/``) never does and
fails loudly when the HIP library is missing.

PARITY UNPINNED (model math): the reference delegates every arithmetic op of
this path to TensorFlow/Keras (``tensorflow>=2.12,<2.16``, pyproject.toml:11),
which is neither vendored under /root/reference nor installable here, and the
reference has no test, golden vector or known-answer value for any model
output.  This file therefore restates the published
semantics of the Keras ops at the reference's call sites; statements that
depend on Keras behaviour are tagged [KERAS-SEMANTICS].  The evaluation
metrics, by contrast, ARE pinned against the reference run in this container
(tests/golden/make_metrics_golden.py -> tests/golden/metrics_golden.json).

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/ebrec/models/newsrec/).

Layout conventions (same as the HIP path):
  ids   (N, T) int          token ids of N titles
  X     (N, L, Din)         a batch of N sequences of length L
  heads are the column blocks [a*d, (a+1)*d) of the E = h*d projection output
"""
from __future__ import annotations

import numpy as np

EPS_KERAS = 1e-7  # K.epsilon()  (layers.py:75-77)

# --------------------------------------------------------------------------
# Counter-based dropout stream shared bit-for-bit with the HIP kernels
# (csrc/ebn_common.h: ebn_lowbias32 / ebn_dropout_keep).  TF's RNG stream is
# not reproducible outside TF ("Dropout RNG cannot match
# TF"), so the build defines its own and tests training-mode parity with it.
# --------------------------------------------------------------------------
_U32 = np.uint32


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> _U32(16)
        x *= _U32(0x7FEB352D)
        x ^= x >> _U32(15)
        x *= _U32(0x846CA68B)
        x ^= x >> _U32(16)
    return x


def dropout_key(seed: int, step: int, site: int) -> int:
    """One 32-bit key per (model seed, optimizer step, dropout call site)."""
    with np.errstate(over="ignore"):
        k = lowbias32(_U32(seed & 0xFFFFFFFF) ^ _U32(0x9E3779B9))
        k = k + _U32(step & 0xFFFFFFFF) * _U32(0x85EBCA6B) + _U32(site & 0xFFFFFFFF) * _U32(0xC2B2AE35)
        k = lowbias32(k)
    return int(k)


def dropout_threshold(p: float) -> int:
    """16-bit threshold: an element is dropped iff its 16-bit hash half is < threshold (probability t/65536)."""
    return int(min(max(int(p * 65536.0), 0), 65535))


def dropout_keep_mask(key: int, n_elem: int, p: float, start: int = 0) -> np.ndarray:
    """keep[i] for flat element index start+i.  One lowbias32 hash decides two consecutive elements: element idx
    uses the low (even idx) / high (odd idx) 16 bits of lowbias32(pair ^ key ^ pair_hi * 0x27D4EB2F), pair = idx >> 1."""
    idx = np.arange(start, start + n_elem, dtype=np.uint64)
    pair = idx >> np.uint64(1)
    lo = (pair & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (pair >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = lowbias32(lo ^ _U32(key) ^ (hi * _U32(0x27D4EB2F)))
    half = np.where((idx & np.uint64(1)).astype(bool), h >> _U32(16), h & _U32(0xFFFF))
    return half >= _U32(dropout_threshold(p))


def dropout_apply(x: np.ndarray, key: int, p: float):
    """Inverted dropout (nrms.py:136,154) [KERAS-SEMANTICS: scale 1/(1-p)]."""
    if p <= 0.0:
        return x, None
    keep = dropout_keep_mask(key, x.size, p).reshape(x.shape)
    scale = x.dtype.type(np.float32(1.0) / np.float32(1.0 - p)) if x.dtype == np.float32 else 1.0 / (1.0 - p)
    m = keep.astype(x.dtype) * scale
    return x * m, m


# --------------------------------------------------------------------------
# initialisers  [KERAS-SEMANTICS]
# --------------------------------------------------------------------------
def glorot_uniform(shape, rng: np.random.Generator, dtype=np.float64):
    fan_in, fan_out = shape[0], shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_nrms_params(V, D, h, d, A, seed=0, dtype=np.float64, table=None):
    """13 arrays in the A.6 interchange order.  Quirk 5 (layers.py:155-172,38,50):
    with a seed, WQ/WK/WV of one layer share one GlorotUniform(seed) draw."""
    E = h * d
    r = lambda: np.random.default_rng(seed)
    P = {}
    P["emb"] = glorot_uniform((V, D), np.random.default_rng(seed + 1), dtype) if table is None else np.asarray(table, dtype)
    w = glorot_uniform((D, E), r(), dtype)
    P["n_WQ"], P["n_WK"], P["n_WV"] = w.copy(), w.copy(), w.copy()
    P["n_W"] = glorot_uniform((E, A), r(), dtype)
    P["n_b"] = np.zeros((A,), dtype)
    P["n_q"] = glorot_uniform((A, 1), r(), dtype)
    w = glorot_uniform((E, E), r(), dtype)
    P["u_WQ"], P["u_WK"], P["u_WV"] = w.copy(), w.copy(), w.copy()
    P["u_W"] = glorot_uniform((E, A), r(), dtype)
    P["u_b"] = np.zeros((A,), dtype)
    P["u_q"] = glorot_uniform((A, 1), r(), dtype)
    return P


def random_nrms_params(V, D, h, d, A, seed=0, dtype=np.float64, scale=1.0):
    """Independent random weights (breaks the WQ=WK=WV symmetry so that a
    Q/K/V mix-up in a kernel cannot hide)."""
    rng = np.random.default_rng(seed)
    E = h * d
    P = {"emb": glorot_uniform((V, D), rng, dtype) * (3.0 * scale)}
    for pre, din in (("n", D), ("u", E)):
        for nm in ("WQ", "WK", "WV"):
            P[f"{pre}_{nm}"] = glorot_uniform((din, E), rng, dtype) * (2.0 * scale)
        P[f"{pre}_W"] = glorot_uniform((E, A), rng, dtype) * (2.0 * scale)
        P[f"{pre}_b"] = (rng.standard_normal(A) * 0.1).astype(dtype)
        P[f"{pre}_q"] = glorot_uniform((A, 1), rng, dtype) * (2.0 * scale)
    return P


PARAM_ORDER = ["emb", "n_WQ", "n_WK", "n_WV", "n_W", "n_b", "n_q",
               "u_WQ", "u_WK", "u_WV", "u_W", "u_b", "u_q"]


# --------------------------------------------------------------------------
# A.1 Embedding (nrms.py:125-134): plain row gather, no mask_zero.
# --------------------------------------------------------------------------
def embedding_fwd(ids, table):
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= table.shape[0]):
        raise IndexError("token id out of range for the embedding table")
    return table[ids]


def embedding_bwd(ids, dX, V):
    dT = np.zeros((V, dX.shape[-1]), dX.dtype)
    np.add.at(dT, np.asarray(ids).reshape(-1), dX.reshape(-1, dX.shape[-1]))
    return dT


# --------------------------------------------------------------------------
# A.2 SelfAttention (layers.py:200-254).  Quirks: no bias, no output
# projection (155-172); O = softmax(QK^T/sqrt(d))^T . V (249, adjoint_a=True);
# no mask (209-211, 186-187).
# --------------------------------------------------------------------------
def self_attention_fwd(X, WQ, WK, WV, h, d):
    N, L, _ = X.shape
    Q = X @ WQ  # layers.py:214
    K = X @ WK  # layers.py:220
    V = X @ WV  # layers.py:226
    Qh = Q.reshape(N, L, h, d).transpose(0, 2, 1, 3)  # layers.py:215-218
    Kh = K.reshape(N, L, h, d).transpose(0, 2, 1, 3)
    Vh = V.reshape(N, L, h, d).transpose(0, 2, 1, 3)
    inv = X.dtype.type(1.0) / np.sqrt(X.dtype.type(d))
    S = (Qh @ Kh.transpose(0, 1, 3, 2)) * inv  # layers.py:231-233
    S = S - S.max(axis=-1, keepdims=True)
    P = np.exp(S)
    P = P / P.sum(axis=-1, keepdims=True)  # K.softmax, last axis; layers.py:247
    Oh = P.transpose(0, 1, 3, 2) @ Vh  # P^T V; layers.py:249
    O = Oh.transpose(0, 2, 1, 3).reshape(N, L, h * d)  # layers.py:250-252
    return O, (X, WQ, WK, WV, Qh, Kh, Vh, P, h, d)


def self_attention_bwd(dO, cache):
    X, WQ, WK, WV, Qh, Kh, Vh, P, h, d = cache
    N, L, _ = X.shape
    inv = X.dtype.type(1.0) / np.sqrt(X.dtype.type(d))
    dOh = dO.reshape(N, L, h, d).transpose(0, 2, 1, 3)  # (N,h,L(j),d)
    dVh = P @ dOh  # dV[i] = sum_j P[i,j] dO[j]
    dP = Vh @ dOh.transpose(0, 1, 3, 2)  # dP[i,j] = V[i].dO[j]
    dS = P * (dP - (P * dP).sum(axis=-1, keepdims=True))
    dQh = (dS @ Kh) * inv
    dKh = (dS.transpose(0, 1, 3, 2) @ Qh) * inv
    back = lambda Z: Z.transpose(0, 2, 1, 3).reshape(N * L, h * d)
    dQ, dK, dV = back(dQh), back(dKh), back(dVh)
    X2 = X.reshape(N * L, -1)
    dWQ, dWK, dWV = X2.T @ dQ, X2.T @ dK, X2.T @ dV
    dX = (dQ @ WQ.T + dK @ WK.T + dV @ WV.T).reshape(X.shape)
    return dX, dWQ, dWK, dWV


# --------------------------------------------------------------------------
# A.3 AttLayer2 (layers.py:55-81).  Quirk 4: exp without max-subtraction and
# a +1e-7 in the denominator (layers.py:71-77).
# --------------------------------------------------------------------------
def att_layer2_fwd(X, W, b, q):
    U = np.tanh(X @ W + b)  # layers.py:65
    e = (U @ q)[..., 0]  # layers.py:66-68
    a = np.exp(e)  # layers.py:71
    w = a / (a.sum(axis=-1, keepdims=True) + X.dtype.type(EPS_KERAS))  # layers.py:75-77
    out = (X * w[..., None]).sum(axis=1)  # layers.py:79-81
    return out, (X, W, q, U, w)


def att_layer2_bwd(dout, cache):
    X, W, q, U, w = cache
    N, L, E = X.shape
    dw = (X * dout[:, None, :]).sum(-1)  # (N,L)
    de = w * (dw - (w * dw).sum(-1, keepdims=True))
    dU = de[..., None] * q[:, 0]
    dq = (U * de[..., None]).sum((0, 1))[:, None]
    dpre = dU * (1 - U * U)
    db = dpre.sum((0, 1))
    dW = X.reshape(N * L, E).T @ dpre.reshape(N * L, -1)
    dX = w[..., None] * dout[:, None, :] + dpre @ W.T
    return dX, dW, db, dq


# --------------------------------------------------------------------------
# A.4 wiring (nrms.py:92-210)
# --------------------------------------------------------------------------
class Drop:
    """Dropout spec for one optimizer step: p, model seed, step."""

    def __init__(self, p, seed, step):
        self.p, self.seed, self.step = float(p), int(seed), int(step)

    def key(self, site):
        return dropout_key(self.seed, self.step, site)


SITE_NEWS_IN, SITE_NEWS_ATT = 0, 1
SITE_MLP0 = 8  # DocVec / optional NRMS MLP: site 8 + layer index


def news_encoder_fwd(ids, P, h, d, drop: Drop | None = None):
    """nrms.py:116-159 (units_per_layer=None branch)."""
    X = embedding_fwd(ids, P["emb"])  # nrms.py:134
    m0 = m1 = None
    if drop is not None and drop.p > 0:
        X, m0 = dropout_apply(X, drop.key(SITE_NEWS_IN), drop.p)  # nrms.py:136
    O, c_sa = self_attention_fwd(X, P["n_WQ"], P["n_WK"], P["n_WV"], h, d)  # nrms.py:137-139
    Y = O
    if drop is not None and drop.p > 0:
        Y, m1 = dropout_apply(O, drop.key(SITE_NEWS_ATT), drop.p)  # nrms.py:154
    out, c_al = att_layer2_fwd(Y, P["n_W"], P["n_b"], P["n_q"])  # nrms.py:156
    return out, (ids, m0, m1, c_sa, c_al, O, Y)


def news_encoder_bwd(dout, cache, V, need_emb_grad=True):
    ids, m0, m1, c_sa, c_al, _, _ = cache
    dY, dW, db, dq = att_layer2_bwd(dout, c_al)
    dO = dY if m1 is None else dY * m1
    dX, dWQ, dWK, dWV = self_attention_bwd(dO, c_sa)
    if m0 is not None:
        dX = dX * m0
    g = {"n_WQ": dWQ, "n_WK": dWK, "n_WV": dWV, "n_W": dW, "n_b": db, "n_q": dq}
    if need_emb_grad:
        g["emb"] = embedding_bwd(ids, dX, V)
    return g


def add_mlp_params(P, units, E, A, seed=0, randomize_bn=True):
    """Parameters of the optional per-token stack of the NRMS news encoder (nrms.py:142-152), prefix "n_";
    the additive-attention kernel n_W then takes the last layer's width."""
    rng = np.random.default_rng(seed)
    prev = E
    for l, u in enumerate(units):
        P[f"n_d{l}_W"] = glorot_uniform((prev, u), rng) * 1.5
        P[f"n_d{l}_b"] = rng.standard_normal(u) * 0.1
        P[f"n_bn{l}_g"] = 1 + 0.1 * rng.standard_normal(u) * randomize_bn
        P[f"n_bn{l}_b"] = 0.1 * rng.standard_normal(u) * randomize_bn
        P[f"n_bn{l}_mean"] = 0.1 * rng.standard_normal(u) * randomize_bn
        P[f"n_bn{l}_var"] = 1 + 0.2 * rng.random(u) * randomize_bn
        prev = u
    P["n_W"] = glorot_uniform((prev, A), rng) * 2
    P["n_units"] = list(units)
    return P


def nrms_mlp_forward(his, pred, P, h, d, training=False, drop: Drop | None = None):
    """NRMS with hparams.newsencoder_units_per_layer (nrms.py:142-152): embedding -> Dropout -> SelfAttention ->
    [Dense-ReLU -> BatchNorm -> Dropout] per token -> AttLayer2; NO dropout straight after the self-attention in
    this branch.  The two TimeDistributed call sites (history, candidates) have their own batch statistics."""
    B, H, T = his.shape
    C = pred.shape[1]
    units = P["n_units"]
    outs, caches, stats = [], [], []
    row_off = 0
    for ids in (his.reshape(B * H, T), pred.reshape(B * C, T)):
        N = ids.shape[0]
        X = embedding_fwd(ids, P["emb"])
        m0 = None
        if training and drop is not None and drop.p > 0:
            keep = dropout_keep_mask(drop.key(SITE_NEWS_IN), X.size, drop.p, start=row_off * T * X.shape[-1]).reshape(X.shape)
            m0 = keep.astype(X.dtype) / (1.0 - drop.p)
            X = X * m0
        O, c_sa = self_attention_fwd(X, P["n_WQ"], P["n_WK"], P["n_WV"], h, d)
        Z, c_stack, st = dense_bn_stack_fwd(O.reshape(N * T, -1), P, units, "n_", training, drop, row_off * T)
        out, c_al = att_layer2_fwd(Z.reshape(N, T, -1), P["n_W"], P["n_b"], P["n_q"])
        outs.append(out)
        caches.append((ids, m0, c_sa, c_stack, c_al, O.shape))
        stats.append(st)
        row_off += N
    NEh, NEc = outs[0].reshape(B, H, -1), outs[1].reshape(B, C, -1)
    user, c_user = user_encoder_from_news_fwd(NEh, P, h, d)
    s = np.einsum("bce,be->bc", NEc, user)
    return softmax_rows(s), s, (B, H, C, caches, c_user, NEc, user, stats)


def nrms_mlp_loss_and_grads(his, pred, y, P, h, d, loss="cross_entropy_loss", l2=0.0, training=True,
                            drop: Drop | None = None):
    probs, s, cache = nrms_mlp_forward(his, pred, P, h, d, training, drop)
    B, H, C, caches, c_user, NEc, user, stats = cache
    units = P["n_units"]
    L, ds = loss_fwd_bwd(s, y, loss)
    dNEc = ds[..., None] * user[:, None, :]
    duser = np.einsum("bc,bce->be", ds, NEc)
    dNEh, g = user_encoder_from_news_bwd(duser, c_user)
    V = P["emb"].shape[0]
    for dout, (ids, m0, c_sa, c_stack, c_al, oshape) in zip((dNEh.reshape(B * H, -1), dNEc.reshape(B * C, -1)), caches):
        dZ, dW, db, dq = att_layer2_bwd(dout, c_al)
        g_stack, dO = dense_bn_stack_bwd(dZ.reshape(-1, dZ.shape[-1]), c_stack, P, units, "n_")
        dX, dWQ, dWK, dWV = self_attention_bwd(dO.reshape(oshape), c_sa)
        if m0 is not None:
            dX = dX * m0
        site = {"n_WQ": dWQ, "n_WK": dWK, "n_WV": dWV, "n_W": dW, "n_b": db, "n_q": dq, "emb": embedding_bwd(ids, dX, V)}
        site.update(g_stack)
        for k, v in site.items():
            g[k] = g.get(k, 0) + v
    for l in range(len(units)):  # kernel_regularizer=l2(lambda), once per kernel
        W = P[f"n_d{l}_W"]
        L = L + l2 * (W * W).sum()
        g[f"n_d{l}_W"] = g[f"n_d{l}_W"] + 2 * l2 * W
    return L, probs, g, stats


def user_encoder_from_news_fwd(NEh, P, h, d):
    """nrms.py:108-111 on already-encoded history (B,H,E)."""
    O, c_sa = self_attention_fwd(NEh, P["u_WQ"], P["u_WK"], P["u_WV"], h, d)
    out, c_al = att_layer2_fwd(O, P["u_W"], P["u_b"], P["u_q"])
    return out, (c_sa, c_al)


def user_encoder_from_news_bwd(duser, cache):
    c_sa, c_al = cache
    dO, dW, db, dq = att_layer2_bwd(duser, c_al)
    dNEh, dWQ, dWK, dWV = self_attention_bwd(dO, c_sa)
    return dNEh, {"u_WQ": dWQ, "u_WK": dWK, "u_WV": dWV, "u_W": dW, "u_b": db, "u_q": dq}


def softmax_rows(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def nrms_forward(his, pred, P, h, d, drop: Drop | None = None):
    """model: softmax_c(news(pred) . user(his))  (nrms.py:195-202).  Returns
    (probs (B,C), scores (B,C), cache)."""
    B, H, T = his.shape
    C = pred.shape[1]
    ids = np.concatenate([his.reshape(B * H, T), pred.reshape(B * C, T)], 0)
    NE, c_news = news_encoder_fwd(ids, P, h, d, drop)
    NEh = NE[: B * H].reshape(B, H, -1)
    NEc = NE[B * H:].reshape(B, C, -1)
    user, c_user = user_encoder_from_news_fwd(NEh, P, h, d)
    s = np.einsum("bce,be->bc", NEc, user)  # Dot(axes=-1), nrms.py:201
    return softmax_rows(s), s, (B, H, C, c_news, c_user, NEc, user)


def scorer_forward(his, pred_one, P, h, d):
    """scorer: sigmoid(news(pred_one) . user(his)) -> (B,1)  (nrms.py:204-205)."""
    _, s, _ = nrms_forward(his, pred_one, P, h, d, None)
    return sigmoid(s)


# A.5 losses [KERAS-SEMANTICS, unverified]: both Keras cross-entropies recover
# the cached logits of the softmax Activation (output._keras_logits).  For
# log_loss the OTHER reading -- SURVEY.md A.5: binary cross-entropy on the
# clipped softmax outputs, what Keras' backend.binary_crossentropy does for a
# tensor without cached logits -- is kind="log_loss_probs"; the TF dump
# (tools/dump_tf_golden.py) decides between the two.
def loss_fwd_bwd(s, y, kind="cross_entropy_loss"):
    """Returns (loss, dL/ds).  nrms.py:56-67."""
    y = np.asarray(y, dtype=s.dtype)
    B, C = s.shape
    if kind == "cross_entropy_loss":  # categorical_crossentropy on logits
        m = s.max(-1, keepdims=True)
        lse = m + np.log(np.exp(s - m).sum(-1, keepdims=True))
        logp = s - lse
        L = -(y * logp).sum(-1).mean()
        ds = (np.exp(logp) * y.sum(-1, keepdims=True) - y) / B
        return L, ds
    if kind == "log_loss":  # binary_crossentropy -> sigmoid CE on the logits
        L = (np.maximum(s, 0) - s * y + np.log1p(np.exp(-np.abs(s)))).mean()
        ds = (sigmoid(s) - y) / (B * C)
        return L, ds
    if kind == "log_loss_probs":  # binary_crossentropy(from_logits=False) on the softmax outputs (nrms.py:54,61-62 per SURVEY A.5)
        # Keras 2.12-2.15 backend.binary_crossentropy: output = clip(output, eps, 1-eps);
        # bce = -(target*log(output+eps) + (1-target)*log(1-output+eps)); mean over the last axis, then over the batch
        eps = s.dtype.type(EPS_KERAS)
        p = softmax_rows(s)
        pc = np.clip(p, eps, 1 - eps)
        L = -(y * np.log(pc + eps) + (1 - y) * np.log(1 - pc + eps)).mean()
        dLdp = np.where((p >= eps) & (p <= 1 - eps), -(y / (pc + eps)) + (1 - y) / (1 - pc + eps), 0.0) / (B * C)
        ds = p * (dLdp - (p * dLdp).sum(-1, keepdims=True))  # softmax backward
        return L, ds
    raise ValueError(f"this loss not defined {kind}")


def nrms_loss_and_grads(his, pred, y, P, h, d, loss="cross_entropy_loss",
                        drop: Drop | None = None, need_emb_grad=True):
    probs, s, cache = nrms_forward(his, pred, P, h, d, drop)
    B, H, C, c_news, c_user, NEc, user = cache
    L, ds = loss_fwd_bwd(s, y, loss)
    dNEc = ds[..., None] * user[:, None, :]
    duser = np.einsum("bc,bce->be", ds, NEc)
    dNEh, g_user = user_encoder_from_news_bwd(duser, c_user)
    dNE = np.concatenate([dNEh.reshape(B * H, -1), dNEc.reshape(B * C, -1)], 0)
    g = news_encoder_bwd(dNE, c_news, P["emb"].shape[0], need_emb_grad)
    g.update(g_user)
    return L, probs, g


def adam_keras_step(theta, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """Keras>=2.11 Adam (A.5) [KERAS-SEMANTICS]: eps added to sqrt(v) before the
    bias correction is folded into the step size.  t starts at 1.  In place."""
    dt = theta.dtype.type
    m += (g - m) * dt(1 - b1)
    v += (g * g - v) * dt(1 - b2)
    alpha = dt(lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t))
    theta -= alpha * m / (np.sqrt(v) + dt(eps))


# --------------------------------------------------------------------------
# NRMSDocVec (nrms_docvec.py:99-188)
# --------------------------------------------------------------------------
BN_EPS, BN_MOM = 1e-3, 0.99  # Keras BatchNormalization defaults [KERAS-SEMANTICS]


def init_docvec_params(Din, units, h, d, A, seed=0, dtype=np.float64, randomize_bn=False):
    rng = np.random.default_rng(seed)
    E = h * d
    P = {"units": list(units)}
    prev = Din
    for l, u in enumerate(units):
        P[f"d{l}_W"] = glorot_uniform((prev, u), rng, dtype)
        P[f"d{l}_b"] = (rng.standard_normal(u) * (0.1 if randomize_bn else 0.0)).astype(dtype)
        P[f"bn{l}_g"] = (1 + 0.1 * rng.standard_normal(u) * randomize_bn).astype(dtype)
        P[f"bn{l}_b"] = (0.1 * rng.standard_normal(u) * randomize_bn).astype(dtype)
        P[f"bn{l}_mean"] = np.zeros(u, dtype)
        P[f"bn{l}_var"] = np.ones(u, dtype)
        prev = u
    P["out_W"] = glorot_uniform((prev, E), rng, dtype)
    P["out_b"] = (rng.standard_normal(E) * (0.1 if randomize_bn else 0.0)).astype(dtype)
    for nm in ("WQ", "WK", "WV"):
        P[f"u_{nm}"] = glorot_uniform((E, E), rng, dtype)
    P["u_W"] = glorot_uniform((E, A), rng, dtype)
    P["u_b"] = np.zeros(A, dtype)
    P["u_q"] = glorot_uniform((A, 1), rng, dtype)
    return P


def dense_bn_stack_fwd(X, P, units, prefix="", training=False, drop: Drop | None = None, row_offset=0):
    """[Dense(u, relu, l2) -> BatchNormalization -> Dropout] x len(units) on the rows of ONE call site
    (nrms_docvec.py:116-124; nrms.py:143-152 applies the same stack per token).  Batch statistics are per call
    site [KERAS-SEMANTICS]; the dropout stream is indexed by (row_offset + row, column)."""
    caches, new_stats = [], []
    x = X
    for l, u in enumerate(units):
        pre = x @ P[f"{prefix}d{l}_W"] + P[f"{prefix}d{l}_b"]
        r = np.maximum(pre, 0)
        if training:
            mu, var = r.mean(0), r.var(0)  # biased batch variance
            new_stats.append((mu, var))
        else:
            mu, var = P[f"{prefix}bn{l}_mean"], P[f"{prefix}bn{l}_var"]
        istd = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
        xh = (r - mu) * istd
        bn = xh * P[f"{prefix}bn{l}_g"] + P[f"{prefix}bn{l}_b"]
        msk = None
        y = bn
        if training and drop is not None and drop.p > 0:
            keep = dropout_keep_mask(drop.key(SITE_MLP0 + l), bn.size, drop.p, start=row_offset * u).reshape(bn.shape)
            msk = keep.astype(bn.dtype) * (1.0 / (1.0 - drop.p))
            y = bn * msk
        caches.append((x, pre, xh, istd, msk))
        x = y
    return x, (caches, training), new_stats


def _relu_gate(pre, layer, gate):
    """d relu(pre) / d pre = [pre > 0] (Keras' relu gradient, nrms_docvec.py:113-130 `activation="relu"`).  `gate(layer, pre)` -- a TEST hook,
    None in every other use -- may return a replacement boolean array: at |pre| within rounding of 0 the derivative is discontinuous and an
    fp32 implementation's choice depends on its summation order; a parity test hands the implementation's own choice back for exactly
    those elements (tests/test_full_size_parity.py) so that the comparison does not hinge on which side of 0 a rounding fell."""
    g = pre > 0
    if gate is not None:
        alt = gate(layer, pre)
        if alt is not None:
            g = alt
    return g


def dense_bn_stack_bwd(dx, cache, P, units, prefix="", gate=None):
    caches, training = cache
    g = {}
    for l in reversed(range(len(units))):
        x, pre_l, xh, istd, msk = caches[l]
        if msk is not None:
            dx = dx * msk
        g[f"{prefix}bn{l}_g"] = (dx * xh).sum(0)
        g[f"{prefix}bn{l}_b"] = dx.sum(0)
        dxh = dx * P[f"{prefix}bn{l}_g"]
        if training:
            R = x.shape[0]
            dr = istd / R * (R * dxh - dxh.sum(0) - xh * (dxh * xh).sum(0))
        else:
            dr = dxh * istd
        dpre_l = dr * _relu_gate(pre_l, l, gate)
        g[f"{prefix}d{l}_W"] = x.T @ dpre_l
        g[f"{prefix}d{l}_b"] = dpre_l.sum(0)
        dx = dpre_l @ P[f"{prefix}d{l}_W"].T
    return g, dx


def docvec_news_encoder_fwd(X, P, training=False, drop: Drop | None = None, site_offset=0, row_offset=0):
    """nrms_docvec.py:113-135.  X (R, Din) = all rows of ONE call site.  Returns out, cache, and the new moving
    statistics (list of (mean, var)) when training."""
    x, c_stack, new_stats = dense_bn_stack_fwd(X, P, P["units"], "", training, drop, row_offset)
    pre = x @ P["out_W"] + P["out_b"]
    out = np.maximum(pre, 0)  # nrms_docvec.py:130
    return out, (c_stack, x, pre), new_stats


def docvec_news_encoder_bwd(dout, cache, P, gate=None):
    c_stack, xl, pre = cache
    dpre = dout * _relu_gate(pre, len(P["units"]), gate)
    g = {"out_W": xl.T @ dpre, "out_b": dpre.sum(0)}
    g_stack, dx = dense_bn_stack_bwd(dpre @ P["out_W"].T, c_stack, P, P["units"], "", gate)
    g.update(g_stack)
    return g, dx


def docvec_forward(his, pred, P, h, d, training=False, drop: Drop | None = None):
    """nrms_docvec.py:139-188: his (B,H,Din), pred (B,C,Din) float."""
    B, H, Din = his.shape
    C = pred.shape[1]
    # two call sites -> two sets of batch statistics (TimeDistributed at 88-90 and 176-178)
    NEh, ch, st_h = docvec_news_encoder_fwd(his.reshape(B * H, Din), P, training, drop, 0, 0)
    NEc, cc, st_c = docvec_news_encoder_fwd(pred.reshape(B * C, Din), P, training, drop, 0, B * H)
    user, c_user = user_encoder_from_news_fwd(NEh.reshape(B, H, -1), P, h, d)
    NEc3 = NEc.reshape(B, C, -1)
    s = np.einsum("bce,be->bc", NEc3, user)
    return softmax_rows(s), s, (B, H, C, ch, cc, c_user, NEc3, user, st_h, st_c)


def docvec_loss_and_grads(his, pred, y, P, h, d, loss="cross_entropy_loss", l2=0.0,
                          training=True, drop: Drop | None = None, relu_gate=None):
    """relu_gate(site, layer, pre) -> bool array | None: test hook, see _relu_gate (site 0 = history rows, 1 = candidate rows; layer
    len(units) = the output Dense)."""
    probs, s, cache = docvec_forward(his, pred, P, h, d, training, drop)
    B, H, C, ch, cc, c_user, NEc3, user, st_h, st_c = cache
    L, ds = loss_fwd_bwd(s, y, loss)
    dNEc = (ds[..., None] * user[:, None, :]).reshape(B * C, -1)
    duser = np.einsum("bc,bce->be", ds, NEc3)
    dNEh, g = user_encoder_from_news_bwd(duser, c_user)
    g_h, _ = docvec_news_encoder_bwd(dNEh.reshape(B * H, -1), ch, P, None if relu_gate is None else (lambda l, pre: relu_gate(0, l, pre)))
    g_c, _ = docvec_news_encoder_bwd(dNEc, cc, P, None if relu_gate is None else (lambda l, pre: relu_gate(1, l, pre)))
    for k in g_h:
        g[k] = g_h[k] + g_c[k]
    # kernel_regularizer=l2(lambda) on the hidden Dense kernels only (116-122; not 130)
    for l in range(len(P["units"])):
        W = P[f"d{l}_W"]
        L = L + l2 * (W * W).sum()
        g[f"d{l}_W"] = g[f"d{l}_W"] + 2 * l2 * W
    return L, probs, g, (st_h, st_c)


def bn_update_moving(P, stats_seq, prefix=""):
    """moving = moving*0.99 + batch*0.01, one update per call site, in call order."""
    for stats in stats_seq:
        for l, (mu, var) in enumerate(stats):
            P[f"{prefix}bn{l}_mean"] = P[f"{prefix}bn{l}_mean"] * BN_MOM + mu * (1 - BN_MOM)
            P[f"{prefix}bn{l}_var"] = P[f"{prefix}bn{l}_var"] * BN_MOM + var * (1 - BN_MOM)
