"""ORACLE -- test infrastructure, NOT product code.

An independent PyTorch-eager (CPU, autograd) restatement of the same reference
math as oracle/nrms_numpy.py.  Two uses only:
  * tests/: cross-checks the numpy oracle's hand-derived backward with autograd;
  * bench.py ``cpu_baseline`` leg (kind "port"): the reference's TF-CPU train
    step cannot run here or on the GPU box (TensorFlow is not installable,
    SURVEY.md section 8c), so this fp32 port -- forward, loss, backward and the
    Keras-form Adam with dense moment decay over the embedding table -- is what
    gets timed on the host cores.

PARITY UNPINNED for the same reason as nrms_numpy.py (no TF, no reference
model tests).  Citations: /root/reference/src/ebrec/models/newsrec/.
"""
from __future__ import annotations

import math

import torch


def self_attention(X, WQ, WK, WV, h, d):
    """layers.py:200-254: no bias, no mask, O = softmax(QK^T/sqrt d)^T V (line 249)."""
    N, L, _ = X.shape
    Q = (X @ WQ).view(N, L, h, d).permute(0, 2, 1, 3)
    K = (X @ WK).view(N, L, h, d).permute(0, 2, 1, 3)
    V = (X @ WV).view(N, L, h, d).permute(0, 2, 1, 3)
    A = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(float(d))
    A = torch.softmax(A, dim=-1)
    O = torch.matmul(A.transpose(-1, -2), V)  # adjoint_a=True
    return O.permute(0, 2, 1, 3).reshape(N, L, h * d)


def att_layer2(X, W, b, q):
    """layers.py:55-81: exp without max-subtraction, +1e-7 in the denominator."""
    e = (torch.tanh(X @ W + b) @ q).squeeze(-1)
    a = torch.exp(e)
    w = a / (a.sum(-1, keepdim=True) + 1e-7)
    return (X * w.unsqueeze(-1)).sum(1)


def news_encoder(ids, P, h, d, masks=None):
    """nrms.py:116-159.  masks = (m0, m1) pre-scaled dropout multipliers or None."""
    X = P["emb"][ids]
    if masks is not None and masks[0] is not None:
        X = X * masks[0]
    Y = self_attention(X, P["n_WQ"], P["n_WK"], P["n_WV"], h, d)
    if masks is not None and masks[1] is not None:
        Y = Y * masks[1]
    return att_layer2(Y, P["n_W"], P["n_b"], P["n_q"])


def user_from_news(NEh, P, h, d):
    """nrms.py:108-111."""
    Y = self_attention(NEh, P["u_WQ"], P["u_WK"], P["u_WV"], h, d)
    return att_layer2(Y, P["u_W"], P["u_b"], P["u_q"])


def nrms_scores(his, pred, P, h, d, masks=None):
    """nrms.py:195-201: raw dot scores (B,C)."""
    B, H, T = his.shape
    C = pred.shape[1]
    ids = torch.cat([his.reshape(B * H, T), pred.reshape(B * C, T)], 0)
    NE = news_encoder(ids, P, h, d, masks)
    user = user_from_news(NE[: B * H].view(B, H, -1), P, h, d)
    return torch.einsum("bce,be->bc", NE[B * H:].view(B, C, -1), user)


def loss_from_scores(s, y, kind="cross_entropy_loss"):
    """nrms.py:56-67 [KERAS-SEMANTICS: both losses run on the softmax's cached logits]."""
    y = y.to(s.dtype)
    if kind == "cross_entropy_loss":
        return -(y * torch.log_softmax(s, -1)).sum(-1).mean()
    if kind == "log_loss":
        return torch.nn.functional.binary_cross_entropy_with_logits(s, y)
    if kind == "log_loss_probs":  # SURVEY.md A.5's reading: Keras backend.binary_crossentropy on the clipped softmax outputs
        eps = 1e-7
        p = torch.softmax(s, -1).clamp(eps, 1 - eps)
        return -(y * torch.log(p + eps) + (1 - y) * torch.log(1 - p + eps)).mean()
    raise ValueError(f"this loss not defined {kind}")


def adam_keras_(theta, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam form, dense over every row."""
    m.add_((g - m) * (1 - b1))
    v.add_((g * g - v) * (1 - b2))
    alpha = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    theta.sub_(alpha * m / (v.sqrt() + eps))


class CpuNRMSTrainer:
    """fp32 CPU train step used as the cpu_baseline 'port'."""

    def __init__(self, P_np: dict, h, d, loss="cross_entropy_loss", lr=1e-4, dropout=0.2,
                 train_embedding=True, seed=0, mask_threads=1):
        self.h, self.d, self.loss, self.lr, self.p = h, d, loss, lr, dropout
        self.mask_threads, self.seed = int(mask_threads), seed
        self.P = {k: torch.tensor(v, dtype=torch.float32) for k, v in P_np.items()}
        self.train_names = [k for k in self.P if train_embedding or k != "emb"]
        for k in self.train_names:
            self.P[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.t = 0
        self.gen = torch.Generator().manual_seed(seed)

    def _rand(self, N, T, W):
        """(N, T, W) uniform draws.  torch's CPU generator is serial (43 M draws per c2 step on one thread); with mask_threads > 1
        the rows are drawn in slabs by a thread pool, one generator per slab (torch releases the GIL inside the fill)."""
        if self.mask_threads <= 1:
            return torch.rand(N, T, W, generator=self.gen)
        from concurrent.futures import ThreadPoolExecutor

        out = torch.empty(N, T, W)
        k = self.mask_threads
        if getattr(self, "_pool", None) is None:
            self._pool = ThreadPoolExecutor(k)
            self._gens = [torch.Generator().manual_seed(int(self.seed) * 1000 + i) for i in range(k)]
        bounds = [N * i // k for i in range(k + 1)]

        def fill(i):
            if bounds[i + 1] > bounds[i]:
                out[bounds[i]: bounds[i + 1]].uniform_(generator=self._gens[i])

        list(self._pool.map(fill, range(k)))
        return out

    def step(self, his, pred, y):
        his, pred, y = (torch.as_tensor(a) for a in (his, pred, y))
        B, H, T = his.shape
        C = pred.shape[1]
        N = B * (H + C)
        masks = None
        if self.p > 0:
            D = self.P["emb"].shape[1]
            E = self.h * self.d
            sc = 1.0 / (1.0 - self.p)
            m0 = (self._rand(N, T, D) >= self.p).float() * sc
            m1 = (self._rand(N, T, E) >= self.p).float() * sc
            masks = (m0, m1)
        s = nrms_scores(his.long(), pred.long(), self.P, self.h, self.d, masks)
        L = loss_from_scores(s, y, self.loss)
        grads = torch.autograd.grad(L, [self.P[k] for k in self.train_names])
        self.t += 1
        with torch.no_grad():
            for k, g in zip(self.train_names, grads):
                adam_keras_(self.P[k], g, self.m[k], self.v[k], self.t, self.lr)
        return float(L.detach())


def docvec_news_encoder(X, P, units, training=True, masks=None, stats_out=None):
    """nrms_docvec.py:113-135 on the rows of ONE TimeDistributed call site: [Dense(u, relu) -> BatchNormalization (batch statistics
    of this call site when training, eps 1e-3 [KERAS-SEMANTICS]) -> Dropout] x len(units) -> Dense(E, relu)."""
    x = X
    for l in range(len(units)):
        r = torch.relu(x @ P[f"d{l}_W"] + P[f"d{l}_b"])
        if training:
            mu, var = r.mean(0), r.var(0, unbiased=False)
            if stats_out is not None:
                stats_out.append((l, mu.detach(), var.detach()))
        else:
            mu, var = P[f"bn{l}_mean"], P[f"bn{l}_var"]
        x = (r - mu) / torch.sqrt(var + 1e-3) * P[f"bn{l}_g"] + P[f"bn{l}_b"]
        if masks is not None:
            x = x * masks[l]
    return torch.relu(x @ P["out_W"] + P["out_b"])


class CpuDocVecTrainer:
    """fp32 CPU train step of NRMSDocVec (nrms_docvec.py:75-188) -- the cpu_baseline 'port' of configs[2]: two call sites with their
    own batch statistics and moving-average updates (history first), dropout after every BatchNormalization, l2 on the hidden Dense
    kernels, the NRMS user encoder and scorer, Keras-form Adam."""

    def __init__(self, P_np: dict, units, h, d, loss="cross_entropy_loss", lr=1e-4, dropout=0.2, l2=1e-4, seed=0):
        self.units, self.h, self.d, self.loss, self.lr, self.p, self.l2 = list(units), h, d, loss, lr, dropout, l2
        self.P = {k: torch.tensor(v, dtype=torch.float32) for k, v in P_np.items() if k != "units"}
        self.moving = [k for k in self.P if k.endswith("_mean") or k.endswith("_var")]
        self.train_names = [k for k in self.P if k not in self.moving]
        for k in self.train_names:
            self.P[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.v = {k: torch.zeros_like(self.P[k]) for k in self.train_names}
        self.t = 0
        self.gen = torch.Generator().manual_seed(seed)

    def loss_and_grads(self, his, pred, y, masks_h=None, masks_c=None):
        his, pred, y = (torch.as_tensor(a) for a in (his, pred, y))
        B, H, Din = his.shape
        C = pred.shape[1]
        stats, dt = [], self.P["out_W"].dtype
        NEh = docvec_news_encoder(his.reshape(B * H, Din).to(dt), self.P, self.units, True, masks_h, stats)
        NEc = docvec_news_encoder(pred.reshape(B * C, Din).to(dt), self.P, self.units, True, masks_c, stats)
        user = user_from_news(NEh.view(B, H, -1), self.P, self.h, self.d)
        s = torch.einsum("bce,be->bc", NEc.view(B, C, -1), user)
        L = loss_from_scores(s, y, self.loss)
        for l in range(len(self.units)):
            L = L + self.l2 * (self.P[f"d{l}_W"] ** 2).sum()
        grads = torch.autograd.grad(L, [self.P[k] for k in self.train_names])
        return L, dict(zip(self.train_names, grads)), stats

    def step(self, his, pred, y):
        B, H, C = his.shape[0], his.shape[1], pred.shape[1]
        masks_h = masks_c = None
        if self.p > 0:
            sc = 1.0 / (1.0 - self.p)
            masks_h = [(torch.rand(B * H, u, generator=self.gen) >= self.p).float() * sc for u in self.units]
            masks_c = [(torch.rand(B * C, u, generator=self.gen) >= self.p).float() * sc for u in self.units]
        L, grads, stats = self.loss_and_grads(his, pred, y, masks_h, masks_c)
        self.t += 1
        with torch.no_grad():
            for l, mu, var in stats:  # history call site first, then the candidates: two moving-average updates per layer and step
                self.P[f"bn{l}_mean"].mul_(0.99).add_(0.01 * mu)
                self.P[f"bn{l}_var"].mul_(0.99).add_(0.01 * var)
            for k in self.train_names:
                adam_keras_(self.P[k], grads[k], self.m[k], self.v[k], self.t, self.lr)
        return float(L.detach())
