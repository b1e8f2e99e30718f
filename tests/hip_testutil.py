"""Helpers shared by the gpu-marked parity tests (all calls go through the C ABI)."""
import ctypes

import numpy as np
import torch

from ebrec import _hip
from oracle import nrms_numpy as on


_KEEP = []  # device tensors stay alive until the test ends: `P(dev(x))` hands a raw pointer to an
# asynchronous launch, and a temporary freed right after P() could be recycled by the next dev().


def dev(a, dtype=torch.float32):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()
    _KEEP.append(t)
    return t


def release_kept():
    torch.cuda.synchronize()
    _KEEP.clear()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def make_state(seed=0, step=0, lr=1e-4, alpha=0.0, advance=False):
    """Device ebn_step_state with the oracle's keys for (seed, step)."""
    st = _hip.StepState()
    st.step, st.seed, st.lr, st.adam_alpha = step, seed & 0xFFFFFFFF, lr, alpha
    for s in range(_hip.binding.EBN_N_SITES):
        st.drop_key[s] = on.dropout_key(seed, step, s)
    t = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
    return t


def read_state(t):
    return _hip.StepState.from_buffer_copy(bytes(t.cpu().numpy().tobytes()))


def P(t):
    return _hip.ptr(t)


def S():
    return _hip.stream_handle()


def i64(x):
    return ctypes.c_int64(int(x))


def gemm(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ws=None):
    if ws is None:
        _hip.call("ebn_gemm_f32", transA, transB, M, N, K, alpha, P(A), lda, P(B), ldb, beta, P(C), ldc, S())
    else:
        _hip.call("ebn_gemm_f32_ws", transA, transB, M, N, K, alpha, P(A), lda, P(B), ldb, beta, P(C), ldc,
                  P(ws), ws.numel(), S())


def assert_close(got, want, rtol=1e-5, atol=1e-6, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} elements off; worst at {i}: got {got[i]!r} want "
                             f"{want[i]!r} (abs err {err[i]:.3e}, tol {tol[i]:.3e}); max abs err {err.max():.3e}")


class ReluTieGate:
    """Order-independence of the NRMSDocVec gradient comparisons (verdict r5 item 3).

    relu'(z) is discontinuous at z = 0.  A pre-activation that the float64 oracle puts within rounding of 0 can come out on
    either side in an fp32 GEMM, depending on its summation order (tile shape, k-split): both are correct roundings of the same
    function, but the gradient element they gate -- and every gradient upstream of it -- differs by a finite amount (measured:
    one weight-gradient column off by 1.3e-3 of the largest gradient about one step in ten at the c3 size,
    profiles/r05_tuning_notes.md).  This object is the oracle's `relu_gate` hook: for the AMBIGUOUS elements only (|z| <= rel x
    max|z| of that layer and call site in float64; rel = 3e-6 is > 10 standard deviations of the fp32 rounding noise of a 768-deep dot product
    relative to the layer's largest pre-activation) it hands the oracle the ENGINE's own choice (whether the engine's stored ReLU
    output is > 0); every other element keeps the oracle's [z > 0].  `fraction()` = the share of ReLU inputs that were treated
    as ambiguous (exact zeros -- zero-padded rows -- are not rounding cases and are not counted); the tests bound it."""

    def __init__(self, eng, n_hist, n_cand, rel=3e-6):
        mb = eng._bufs["mlp"]
        rb = eng.mlp.bufs(mb["N"])
        L = len(eng.units)
        n = n_hist + n_cand
        self.on = [rb["R"][l][:n].cpu().numpy() > 0 for l in range(L)] + [mb["NE"][:n].cpu().numpy() > 0]
        self.rows = {0: slice(0, n_hist), 1: slice(n_hist, n)}
        self.rel, self.n_ambiguous, self.n_total, self.n_flipped = rel, 0, 0, 0

    def __call__(self, site, layer, pre):
        tie = (np.abs(pre) <= self.rel * np.abs(pre).max()) & (pre != 0)
        self.n_total += pre.size
        if not tie.any():
            return None
        gate = pre > 0
        mine = self.on[layer][self.rows[site]]
        assert mine.shape == pre.shape, (mine.shape, pre.shape)
        self.n_ambiguous += int(tie.sum())
        self.n_flipped += int((gate != mine)[tie].sum())
        return np.where(tie, mine, gate)

    def fraction(self):
        return self.n_ambiguous / max(self.n_total, 1)
