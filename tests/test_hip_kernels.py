"""Parity of every low-level C-ABI entry point against the float64 oracle (gpu-marked).

Tolerances: the kernels compute in fp32 (exact-fp32 MFMA / VALU); the oracle is float64.
rtol 2e-5 / atol scaled to the magnitude of each quantity -- an order tighter than the 1e-4
forward-score budget of BASELINE.json's north_star.  Gather/scatter-without-dropout, dropout
masks and the step-state keys are bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nrms_numpy as on
from pathlib import Path

from tests.hip_testutil import P, S, assert_close, dev, gemm, host, make_state, read_state

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


# ---------------------------------------------------------------- step state
def test_step_advance_matches_oracle_keys_and_alpha(hip):
    st = make_state(seed=42, step=0, lr=1e-4)
    for t in range(1, 4):
        hip.call("ebn_step_advance", P(st), 0.9, 0.999, S())
        got = read_state(st)
        assert got.step == t and got.seed == 42
        want_alpha = 1e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        assert abs(got.adam_alpha - want_alpha) <= 1e-6 * want_alpha
        for s in range(12):
            assert got.drop_key[s] == on.dropout_key(42, t, s)


# ---------------------------------------------------------------- a1 gather / scatter
@pytest.mark.parametrize("V,D,n_tok", [(1000, 300, 750), (5000, 1024, 1500), (37, 7, 129), (64, 4, 1), (100, 768, 25)])
def test_gather_bit_exact(hip, V, D, n_tok):
    rng = np.random.default_rng(0)
    table = rng.standard_normal((V, D)).astype(np.float32)
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    ids[0] = 0
    ids[-1] = V - 1
    out = torch.empty(n_tok, D, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("ebn_gather_rows_f32", P(dev(ids, torch.int32)), P(dev(table)), P(out), n_tok, D, V, None, -1, 0.0,
             P(flag), S())
    assert np.array_equal(out.cpu().numpy(), table[ids])
    assert int(flag.item()) == 0


def test_gather_oob_sets_flag_and_writes_zero_row(hip):
    V, D = 10, 8
    table = np.ones((V, D), np.float32)
    ids = np.array([1, 10, -1, 3], np.int32)
    out = torch.full((4, D), 7.0, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("ebn_gather_rows_f32", P(dev(ids, torch.int32)), P(dev(table)), P(out), 4, D, V, None, -1, 0.0,
             P(flag), S())
    o = out.cpu().numpy()
    assert int(flag.item()) == 1
    assert (o[0] == 1).all() and (o[3] == 1).all() and (o[1] == 0).all() and (o[2] == 0).all()


@pytest.mark.parametrize("D", [300, 10])
def test_gather_dropout_mask_bit_exact(hip, D):
    rng = np.random.default_rng(1)
    V, n_tok, p, seed, step = 200, 333, 0.2, 7, 5
    table = rng.standard_normal((V, D)).astype(np.float32)
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    st = make_state(seed=seed, step=step)
    out = torch.empty(n_tok, D, device="cuda")
    hip.call("ebn_gather_rows_f32", P(dev(ids, torch.int32)), P(dev(table)), P(out), n_tok, D, V, P(st), 0,
             ctypes.c_float(p), None, S())
    keep = on.dropout_keep_mask(on.dropout_key(seed, step, 0), n_tok * D, p).reshape(n_tok, D)
    want = table[ids] * keep.astype(np.float32) * (np.float32(1.0) / np.float32(1.0 - p))
    assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
    assert 0.75 < keep.mean() < 0.85


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_embedding_grad_scatter(hip, p):
    rng = np.random.default_rng(2)
    V, D, n_tok, seed, step = 50, 12, 400, 3, 9
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    ids[:100] = 0  # hot row (padded history)
    dX = rng.standard_normal((n_tok, D)).astype(np.float32)
    st = make_state(seed=seed, step=step)
    dT = torch.zeros(V, D, device="cuda")
    hip.call("ebn_embedding_grad_scatter_f32", P(dev(ids, torch.int32)), P(dev(dX)), P(dT), n_tok, D, V, P(st), 0,
             ctypes.c_float(p), S())
    m = np.ones((n_tok, D))
    if p > 0:
        m = on.dropout_keep_mask(on.dropout_key(seed, step, 0), n_tok * D, p).reshape(n_tok, D) / (1 - p)
    want = on.embedding_bwd(ids, dX.astype(np.float64) * m, V)
    assert_close(host(dT), want, rtol=1e-5, atol=1e-4, what="dTable")  # atomics: order-dependent rounding


# ---------------------------------------------------------------- GEMM
GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (64, 64, 16), (130, 70, 33), (257, 129, 300), (750, 1200, 300),
               (640, 200, 400), (300, 1200, 2000), (96, 100, 4096), (1, 400, 400), (513, 5, 64),
               # shapes whose plan picks the 256x64 tile family (ragged M and N, split-K)
               (6000, 1200, 256), (5000, 300, 128), (1030, 1210, 3000),
               # split-K whose last K range ends in a partial slab (direct-to-LDS fetch for the full slabs, guarded loader for the tail)
               (256, 256, 2004), (512, 64, 8200),
               # the small-output kernel with two slabs of loads in flight: 6 full slabs; 3 full + a 16-wide tail; 5 full, ragged M
               (800, 512, 768), (640, 1200, 400), (250, 768, 640),
               # the edges of the planner's small-tile envelope (K = 1536 | 1537, tiles64 = 256 at K = 4096)
               (640, 1200, 1536), (640, 1200, 1537), (1024, 1024, 4096),
               # tall outputs with an N the 64-wide tiles pad by > 10 %: the LDS-free 16x16-block kernels (ebn_gemm_direct.hip) for A not
               # transposed and beta = 0 -- AttLayer2's two shapes (short), ragged M / N / K with a partial last slab, two column panels
               (4096, 200, 400), (4100, 400, 200), (4099, 68, 72), (4500, 416, 100), (5000, 100, 64),
               # small, awkward outputs under a long contraction (AttLayer2's weight gradient and ragged relatives): the transposed-A layout
               # takes the 16x16-block K-chunked kernel of ebn_gemm_direct.hip -- partial last group (K % 16 != 0), partial last chunk,
               # M / N not multiples of 16, every (R, CW) instantiation the plan can pick
               (400, 200, 24000), (416, 208, 9000), (100, 500, 5003), (500, 60, 4100), (72, 72, 30001), (330, 330, 7777), (300, 1200, 9000), (512, 1280, 4100)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_all_layouts(hip, M, N, K, tA, tB):
    rng = np.random.default_rng(M * 131 + N * 17 + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    opA = A.T if tA else A
    opB = B.T if tB else B
    ref = opA.astype(np.float64) @ opB.astype(np.float64)
    for alpha, beta, use_ws in ((1.0, 0.0, False), (0.5, 1.0, False), (1.0, 0.0, True), (2.0, -0.5, True)):
        C = dev(C0)
        ws = None
        if use_ws:
            n = hip.lib().ebn_gemm_workspace_floats(M, N, K)
            ws = torch.empty(max(int(n), 1), device="cuda")
        gemm(tA, tB, M, N, K, alpha, dev(A), A.shape[1], dev(B), B.shape[1], beta, C, N, ws)
        want = alpha * ref + beta * C0
        assert_close(host(C), want, rtol=2e-6, atol=1e-5 + 3e-7 * K, what=f"gemm {M}x{N}x{K} tA={tA} tB={tB} a={alpha} b={beta} ws={use_ws}")


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_split_precision_gemm_holds_the_exact_path_tolerances(hip, M, N, K, tA, tB):
    """ebn_gemm_f32_prec(precision = 1): every fp32 operand element split exactly into three bf16 values, six bf16 MFMA
    products with fp32 accumulation.  The SAME shapes, layouts, alpha / beta cases and tolerances (rtol 2e-6 against float64)
    as test_gemm_all_layouts asserts for the exact-fp32 kernels -- an opt-in precision that is not a reduced one."""
    rng = np.random.default_rng(M * 131 + N * 17 + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64)
    nbytes = int(hip.lib().ebn_gemm_prec_workspace_bytes(M, N, K, 1))
    ws = torch.empty(nbytes // 4 + 64, device="cuda")
    for alpha, beta in ((1.0, 0.0), (0.5, 1.0), (2.0, -0.5)):
        C = dev(C0)
        hip.call("ebn_gemm_f32_prec", tA, tB, M, N, K, ctypes.c_float(alpha), P(dev(A)), A.shape[1], P(dev(B)), B.shape[1], ctypes.c_float(beta),
                 P(C), N, P(ws), nbytes, 1, S())
        assert_close(host(C), alpha * ref + beta * C0, rtol=2e-6, atol=1e-5 + 3e-7 * K, what=f"split gemm {M}x{N}x{K} tA={tA} tB={tB} a={alpha} b={beta}")
    # precision 0 through the same entry point is the exact path, bit for bit
    C1, C2 = dev(C0), dev(C0)
    n0 = int(hip.lib().ebn_gemm_prec_workspace_bytes(M, N, K, 0))
    ws0 = torch.empty(max(n0 // 4, 1), device="cuda")
    hip.call("ebn_gemm_f32_prec", tA, tB, M, N, K, ctypes.c_float(1.0), P(dev(A)), A.shape[1], P(dev(B)), B.shape[1], ctypes.c_float(0.0), P(C1), N,
             P(ws0), n0, 0, S())
    gemm(tA, tB, M, N, K, 1.0, dev(A), A.shape[1], dev(B), B.shape[1], 0.0, C2, N, ws0)
    assert torch.equal(C1, C2)


def test_split_precision_is_fp32_accurate_not_bf16_accurate(hip):
    """The six-product split against the exact-fp32 kernel on the projection's contraction length: the two differ from float64
    by the same order (fp32 accumulation noise); a plain bf16 product would be off by ~3e-3 relative.  Asymmetric operands with
    padded leading dimensions (an operand or output swap cannot hide), ragged M / N / K (zero padding of the planes)."""
    M, N, K = 777, 333, 1030
    rng = np.random.default_rng(3)
    A = (rng.standard_normal((M, K + 6)) * np.exp(rng.standard_normal((M, 1)))).astype(np.float32)  # rows of very different scale
    B = rng.standard_normal((K, N + 2)).astype(np.float32)
    ref = A[:, :K].astype(np.float64) @ B[:, :N].astype(np.float64)
    nbytes = int(hip.lib().ebn_gemm_prec_workspace_bytes(M, N, K, 1))
    ws = torch.empty(nbytes // 4 + 64, device="cuda")
    Cs, Ce = torch.full((M, N + 3), 9.0, device="cuda"), torch.full((M, N + 3), 9.0, device="cuda")
    hip.call("ebn_gemm_f32_prec", 0, 0, M, N, K, ctypes.c_float(1.0), P(dev(A)), K + 6, P(dev(B)), N + 2, ctypes.c_float(0.0), P(Cs), N + 3, P(ws), nbytes, 1, S())
    gemm(0, 0, M, N, K, 1.0, dev(A), K + 6, dev(B), N + 2, 0.0, Ce, N + 3)
    assert torch.all(Cs[:, N:] == 9.0)  # the padding columns of C are untouched
    scale = np.abs(A[:, :K]).astype(np.float64) @ np.abs(B[:, :N]).astype(np.float64)  # sum |a||b| per output: the error scale
    err_s = np.abs(host(Cs)[:, :N] - ref) / scale
    err_e = np.abs(host(Ce)[:, :N] - ref) / scale
    assert err_s.max() < 3e-7 and err_e.max() < 3e-7, (err_s.max(), err_e.max())  # both: a few fp32 ulps of the error scale
    assert err_s.max() < 4 * err_e.max() + 1e-8  # ... and of the same order


def test_gemm_is_asymmetric_safe_and_respects_ld(hip):
    """A = I against an asymmetric B, with padded leading dimensions (catches C^T / operand swaps)."""
    M = N = K = 96
    lda, ldb, ldc = 100, 104, 128
    A = np.zeros((M, lda), np.float32)
    A[:, :K] = np.eye(M, dtype=np.float32)
    B = np.zeros((K, ldb), np.float32)
    B[:, :N] = np.arange(K * N, dtype=np.float32).reshape(K, N) / 100.0
    C = torch.full((M, ldc), -1.0, device="cuda")
    gemm(0, 0, M, N, K, 1.0, dev(A), lda, dev(B), ldb, 0.0, C, ldc)
    c = C.cpu().numpy()
    assert np.array_equal(c[:, :N], B[:, :N])
    assert (c[:, N:] == -1).all()


def test_gemm_unaligned_pointers_take_the_scalar_path(hip):
    rng = np.random.default_rng(5)
    M, N, K = 33, 45, 27  # ld not multiples of 4
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    C = torch.zeros(M, N, device="cuda")
    gemm(0, 0, M, N, K, 1.0, dev(A), K, dev(B), N, 0.0, C, N)
    assert_close(host(C), A.astype(np.float64) @ B.astype(np.float64), rtol=2e-6, atol=2e-5, what="odd gemm")


@pytest.mark.parametrize("R,K_in,N_out,beta", [(640, 400, 200, 0.0), (640, 400, 1200, 1.0), (800, 768, 512, 0.0), (250, 76, 36, 0.5),
                                               (3000, 1200, 300, 0.0), (33, 30, 18, 1.0), (0, 64, 32, 1.0)])
def test_dense_backward_pair_equals_the_two_gemms(hip, R, K_in, N_out, beta):
    """ebn_dense_bwd_pair_f32: dW = X^T.dY + beta*dW and dX = dY.W^T, grouped into one launch for small-output aligned
    shapes, two plain GEMMs otherwise (big, unaligned, empty) -- all against float64."""
    rng = np.random.default_rng(R * 7 + K_in * 3 + N_out)
    X = rng.standard_normal((R, K_in)).astype(np.float32)
    dY = rng.standard_normal((R, N_out)).astype(np.float32)
    W = rng.standard_normal((K_in, N_out)).astype(np.float32)
    dW0 = rng.standard_normal((K_in, N_out)).astype(np.float32)
    dW, dX = dev(dW0), torch.full((max(R, 1), K_in), float("nan"), device="cuda")
    n = max(int(hip.lib().ebn_gemm_workspace_floats(K_in, N_out, max(R, 1))), int(hip.lib().ebn_gemm_workspace_floats(max(R, 1), K_in, N_out)), 1)
    ws = torch.empty(n, device="cuda")
    hip.call("ebn_dense_bwd_pair_f32", R, K_in, N_out, P(dev(X) if R else torch.zeros(1, device="cuda")), K_in,
             P(dev(dY) if R else torch.zeros(1, device="cuda")), N_out, P(dev(W)), N_out, ctypes.c_float(beta), P(dW), N_out, P(dX), K_in,
             P(ws), ws.numel(), S())
    assert_close(host(dW), X.astype(np.float64).T @ dY.astype(np.float64) + beta * dW0, rtol=2e-6, atol=1e-5 + 3e-7 * max(R, 1), what="pair dW")
    if R:
        assert_close(host(dX)[:R], dY.astype(np.float64) @ W.astype(np.float64).T, rtol=2e-6, atol=1e-5 + 3e-7 * N_out, what="pair dX")


# ---------------------------------------------------------------- a3/a6 attention core
ATTN_CASES = [(4, 30, 20, 20), (3, 20, 20, 20), (2, 50, 16, 16), (5, 1, 2, 4), (2, 33, 3, 32), (3, 32, 2, 5),
              (1, 64, 2, 8),
              # the other instantiations of the MFMA path (L <= 32, d in {16, 20, 32}), incl. its largest LDS footprint
              (2, 32, 4, 32), (3, 17, 2, 16), (2, 32, 2, 16), (1, 5, 3, 32), (9, 1, 1, 20),
              # group-form backward (d = 20, head_num a multiple of 4): short and full tiles, one group per sequence
              (1030, 7, 4, 20), (520, 32, 8, 20),
              # long sequences (64 < L <= 256): recompute-based kernels
              (2, 100, 2, 20), (1, 256, 1, 32), (3, 65, 3, 16),
              # the backward image of the one-wave kernel exceeds 64 KB here (forward fits): must route to the long kernels
              (2, 64, 2, 32), (2, 50, 20, 20)]


def _qkv_case(n_seq, L, h, d, seed):
    rng = np.random.default_rng(seed)
    E = h * d
    qkv = (rng.standard_normal((n_seq, L, 3 * E)) * 0.8).astype(np.float32)
    Qh = qkv[..., :E].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    Kh = qkv[..., E:2 * E].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    Vh = qkv[..., 2 * E:].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    S_ = Qh @ Kh.transpose(0, 1, 3, 2) / np.sqrt(d)
    Pm = on.softmax_rows(S_)
    return qkv, Qh, Kh, Vh, Pm


@pytest.mark.parametrize("n_seq,L,h,d", ATTN_CASES)
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attention_forward_is_transposed_softmax_times_v(hip, n_seq, L, h, d, p):
    E = h * d
    qkv, Qh, Kh, Vh, Pm = _qkv_case(n_seq, L, h, d, 11)
    O = (Pm.transpose(0, 1, 3, 2) @ Vh).transpose(0, 2, 1, 3).reshape(n_seq, L, E)  # P^T V (layers.py:249)
    wrong = (Pm @ Vh).transpose(0, 2, 1, 3).reshape(n_seq, L, E)  # standard attention
    seed, step = 5, 2
    if p > 0:
        O = O * on.dropout_keep_mask(on.dropout_key(seed, step, 1), O.size, p).reshape(O.shape) / (1 - p)
    st = make_state(seed=seed, step=step)
    out = torch.empty(n_seq * L, E, device="cuda")
    hip.call("ebn_attn_fwd_f32", P(dev(qkv.reshape(n_seq * L, 3 * E))), 3 * E, P(out), E, n_seq, L, h, d, P(st), 1,
             ctypes.c_float(p), S())
    assert_close(host(out).reshape(n_seq, L, E), O, rtol=2e-5, atol=2e-6, what="attn fwd")
    if L > 2 and p == 0:
        assert np.abs(host(out).reshape(n_seq, L, E) - wrong).max() > 1e-3  # the quirk is observable


@pytest.mark.parametrize("n_seq,L,h,d", [(3, 30, 20, 20), (2, 50, 4, 16)])  # MFMA path and the fallback
def test_attention_forward_with_saturated_softmax(hip, n_seq, L, h, d):
    """Logits of magnitude ~1e2 (one-hot attention rows): the exp2-domain softmax must neither overflow nor lose the
    winner; finite everywhere and equal to the float64 oracle."""
    E = h * d
    rng = np.random.default_rng(3)
    qkv = (rng.standard_normal((n_seq, L, 3 * E))).astype(np.float32)
    qkv[..., :2 * E] *= 7.0  # |q.k|/sqrt(d) up to a few hundred
    Qh = qkv[..., :E].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    Kh = qkv[..., E:2 * E].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    Vh = qkv[..., 2 * E:].reshape(n_seq, L, h, d).transpose(0, 2, 1, 3).astype(np.float64)
    Pm = on.softmax_rows(Qh @ Kh.transpose(0, 1, 3, 2) / np.sqrt(d))
    O = (Pm.transpose(0, 1, 3, 2) @ Vh).transpose(0, 2, 1, 3).reshape(n_seq, L, E)
    out = torch.empty(n_seq * L, E, device="cuda")
    hip.call("ebn_attn_fwd_f32", P(dev(qkv.reshape(n_seq * L, 3 * E))), 3 * E, P(out), E, n_seq, L, h, d, None, -1,
             ctypes.c_float(0.0), S())
    got = host(out).reshape(n_seq, L, E)
    assert np.isfinite(got).all()
    assert_close(got, O, rtol=2e-4, atol=2e-4, what="saturated attn fwd")  # logits ~3e2: fp32 resolution of the exponent


@pytest.mark.parametrize("n_seq,L,h,d", ATTN_CASES)
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attention_backward(hip, n_seq, L, h, d, p):
    E = h * d
    qkv, Qh, Kh, Vh, Pm = _qkv_case(n_seq, L, h, d, 12)
    rng = np.random.default_rng(13)
    dY = rng.standard_normal((n_seq, L, E)).astype(np.float32)
    seed, step = 8, 3
    dO = dY.astype(np.float64)
    if p > 0:
        dO = dO * on.dropout_keep_mask(on.dropout_key(seed, step, 1), dO.size, p).reshape(dO.shape) / (1 - p)
    dOh = dO.reshape(n_seq, L, h, d).transpose(0, 2, 1, 3)
    dVh = Pm @ dOh
    dP = Vh @ dOh.transpose(0, 1, 3, 2)
    dS = Pm * (dP - (Pm * dP).sum(-1, keepdims=True))
    dQh = dS @ Kh / np.sqrt(d)
    dKh = dS.transpose(0, 1, 3, 2) @ Qh / np.sqrt(d)
    back = lambda Z: Z.transpose(0, 2, 1, 3).reshape(n_seq * L, E)
    want = np.concatenate([back(dQh), back(dKh), back(dVh)], 1)
    st = make_state(seed=seed, step=step)
    dqkv = torch.full((n_seq * L, 3 * E), float("nan"), device="cuda")
    hip.call("ebn_attn_bwd_f32", P(dev(qkv.reshape(n_seq * L, 3 * E))), 3 * E, P(dev(dY.reshape(n_seq * L, E))), E,
             P(dqkv), 3 * E, n_seq, L, h, d, P(st), 1, ctypes.c_float(p), S())
    assert_close(host(dqkv), want, rtol=3e-5, atol=5e-6, what="attn bwd")


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("n_seq,L,h,d", [(4, 30, 20, 20), (3, 20, 20, 20), (2, 50, 20, 20), (2, 64, 2, 32), (3, 17, 2, 16), (2, 33, 3, 32), (5, 1, 1, 20)])
def test_attention_backward_with_the_pooling_term_folded_in(hip, n_seq, L, h, d, p):
    """ebn_attn_bwd_pooled_f32(dout, w, dpool) == ebn_attn_bwd_f32(dout + w (x) dpool): the AttLayer2 pooling half of d(Y)
    is added while dO is staged (to rounding: one fma per element instead of a separately rounded sum)."""
    E = h * d
    qkv, *_ = _qkv_case(n_seq, L, h, d, 21)
    rng = np.random.default_rng(22)
    dY = rng.standard_normal((n_seq * L, E)).astype(np.float32)
    w = rng.random(n_seq * L).astype(np.float32)
    dpool = rng.standard_normal((n_seq, E)).astype(np.float32)
    full = (dY.astype(np.float64) + w[:, None].astype(np.float64) * np.repeat(dpool.astype(np.float64), L, axis=0)).astype(np.float32)
    assert hip.lib().ebn_attn_bwd_pooled_supported(L, d) == 1
    st = make_state(seed=5, step=2)
    q = dev(qkv.reshape(n_seq * L, 3 * E))
    a, b = torch.full((n_seq * L, 3 * E), float("nan"), device="cuda"), torch.full((n_seq * L, 3 * E), float("nan"), device="cuda")
    hip.call("ebn_attn_bwd_f32", P(q), 3 * E, P(dev(full)), E, P(a), 3 * E, n_seq, L, h, d, P(st), 1, ctypes.c_float(p), S())
    hip.call("ebn_attn_bwd_pooled_f32", P(q), 3 * E, P(dev(dY)), E, P(dev(w)), P(dev(dpool)), E, P(b), 3 * E, n_seq, L, h, d, P(st), 1,
             ctypes.c_float(p), S())
    assert_close(host(b), host(a), rtol=2e-5, atol=2e-5, what="pooled attn bwd")
    # shapes outside the MFMA path are refused (the caller keeps the rank-1 GEMM epilogue there)
    assert hip.lib().ebn_attn_bwd_pooled_supported(65, 20) == 0 and hip.lib().ebn_attn_bwd_pooled_supported(30, 8) == 0


_GROUP_SCRIPT = r'''
import ctypes, hashlib, sys
sys.path.insert(0, "{root}/ebnerd-benchmark_amd")
import torch
from ebrec import _hip
P, S = _hip.ptr, _hip.stream_handle
g = torch.Generator(device="cuda").manual_seed(5)
st = _hip.StepState(); st.step, st.seed, st.lr = 3, 7, 1e-3
for s_ in range(_hip.binding.EBN_N_SITES):
    st.drop_key[s_] = (0x9E3779B9 * (s_ + 1)) & 0xFFFFFFFF
st_dev = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
out = []
for n_seq, L, h in ((1400, 30, 20), (300, 20, 20), (1100, 11, 4), (2100, 20, 20)):
    d = 20; E = h * d; R = n_seq * L
    qkv = torch.randn(R, 3 * E, device="cuda", generator=g)
    dY = torch.randn(R, E, device="cuda", generator=g)
    w = torch.rand(R, device="cuda", generator=g)
    dpool = torch.randn(n_seq, E, device="cuda", generator=g)
    for p in (0.0, 0.2):
        a = torch.zeros(R, 3 * E, device="cuda"); b = torch.zeros(R, 3 * E, device="cuda")
        _hip.call("ebn_attn_bwd_f32", P(qkv), 3 * E, P(dY), E, P(a), 3 * E, n_seq, L, h, d, P(st_dev), 1, ctypes.c_float(p), S())
        _hip.call("ebn_attn_bwd_pooled_f32", P(qkv), 3 * E, P(dY), E, P(w), P(dpool), E, P(b), 3 * E, n_seq, L, h, d, P(st_dev), 1,
                  ctypes.c_float(p), S())
        y = torch.zeros(R, E, device="cuda")
        _hip.call("ebn_attn_fwd_f32", P(qkv), 3 * E, P(y), E, n_seq, L, h, d, P(st_dev), 1, ctypes.c_float(p), S())
        torch.cuda.synchronize()
        out.append(hashlib.sha256(a.cpu().numpy().tobytes() + b.cpu().numpy().tobytes() + y.cpu().numpy().tobytes()).hexdigest())
print("DIGESTS", " ".join(out))
'''


def test_group_form_attention_kernels_are_the_per_wave_kernels_bit_for_bit(hip, tmp_path):
    """Forward and backward of the title-level attention run four heads of a title per workgroup (operands in, results out through
    the whole workgroup); EBN_ATTN_BWD_PER_WAVE=1 selects the one-wave-per-head kernels they replace.  Same MFMA order, same fma for
    the pooling term, same mask: identical bytes, with and without dropout / pooling term."""
    import os
    import subprocess
    import sys
    script = tmp_path / "attn_group.py"
    script.write_text(_GROUP_SCRIPT.format(root=str(ROOT)))
    got = []
    for flag in ("0", "1"):
        out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, EBN_ATTN_BWD_PER_WAVE=flag), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "DIGESTS" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
        got.append(out.stdout.split("DIGESTS", 1)[1].split())
    assert len(got[0]) == 8 and got[0] == got[1]


def test_attention_rejects_unsupported_shapes(hip):
    x = torch.zeros(10, device="cuda")
    with pytest.raises(hip.HipError):
        hip.call("ebn_attn_fwd_f32", P(x), 300, P(x), 100, 1, 257, 2, 8, None, -1, ctypes.c_float(0), S())
    with pytest.raises(hip.HipError):
        hip.call("ebn_attn_fwd_f32", P(x), 300, P(x), 100, 1, 10, 2, 33, None, -1, ctypes.c_float(0), S())


# ---------------------------------------------------------------- a4/a7 AttLayer2 tail
@pytest.mark.parametrize("n_seq,L,E,A", [(6, 30, 400, 200), (3, 20, 256, 200), (2, 50, 40, 7), (4, 1, 8, 3), (1, 70, 64, 300)])
def test_attpool_forward_and_backward(hip, n_seq, L, E, A):
    rng = np.random.default_rng(21)
    X = rng.standard_normal((n_seq, L, E)) * 0.7
    W = on.glorot_uniform((E, A), rng) * 2
    b = rng.standard_normal(A) * 0.1
    q = on.glorot_uniform((A, 1), rng) * 2
    out, cache = on.att_layer2_fwd(X, W, b, q)
    R = n_seq * L
    Xd, Wd = dev(X.reshape(R, E)), dev(W)
    U = torch.empty(R, A, device="cuda")
    gemm(0, 0, R, A, E, 1.0, Xd, E, Wd, A, 0.0, U, A)
    o = torch.empty(n_seq, E, device="cuda")
    w = torch.empty(R, device="cuda")
    hip.call("ebn_attpool_fwd_f32", P(U), P(dev(b)), P(dev(q)), P(Xd), P(o), P(w), n_seq, L, E, A, S())
    assert_close(host(U).reshape(n_seq, L, A), cache[3], rtol=2e-5, atol=3e-6, what="tanh U")
    assert_close(host(w).reshape(n_seq, L), cache[4], rtol=3e-5, atol=1e-7, what="attention weights")
    assert_close(host(o), out, rtol=3e-5, atol=3e-6, what="pooled out")
    # the +1e-7 of layers.py:75-77 is there: weights sum to slightly less than one
    dout = rng.standard_normal((n_seq, E))
    dX, dW, db, dq = on.att_layer2_bwd(dout, cache)
    dXd = torch.empty(R, E, device="cuda")
    de = torch.empty(R, device="cuda")
    hip.call("ebn_attpool_bwd_pool_f32", P(Xd), P(w), P(dev(dout)), P(dXd), P(de), n_seq, L, E, S())
    part = torch.empty(int(hip.lib().ebn_attpool_partials_len(R, A)), device="cuda")
    dqd = torch.full((A,), 5.0, device="cuda")
    dbd = torch.full((A,), 5.0, device="cuda")
    hip.call("ebn_attpool_bwd_dpre_f32", P(U), P(dev(q)), P(de), P(dqd), P(dbd), P(part), R, A, 0, S())
    assert_close(host(dqd), dq[:, 0], rtol=3e-5, atol=2e-5, what="dq")
    assert_close(host(dbd), db, rtol=3e-5, atol=2e-5, what="db")
    dWd = torch.empty(E, A, device="cuda")
    gemm(1, 0, E, A, R, 1.0, Xd, E, U, A, 0.0, dWd, A)
    gemm(0, 1, R, E, A, 1.0, U, A, Wd, A, 1.0, dXd, E)
    assert_close(host(dWd), dW, rtol=3e-5, atol=3e-5, what="dW")
    assert_close(host(dXd).reshape(n_seq, L, E), dX, rtol=3e-5, atol=1e-5, what="dX")
    # the fused form the encoder stage uses: pool backward without its dX write + the rank-1 term in the GEMM epilogue
    de2 = torch.empty(R, device="cuda")
    hip.call("ebn_attpool_bwd_pool_f32", P(Xd), P(w), P(dev(dout)), None, P(de2), n_seq, L, E, S())
    assert np.array_equal(host(de2), host(de)), "de must not depend on whether dX is written"
    for use_ws in (False, True):
        ws = torch.empty(max(int(hip.lib().ebn_gemm_workspace_floats(R, E, A)), 1), device="cuda") if use_ws else None
        dX2 = torch.full((R, E), 7.0, device="cuda")
        hip.call("ebn_gemm_f32_rank1", R, E, A, ctypes.c_float(1.0), P(U), A, P(Wd), A, P(dX2), E, P(w), P(dev(dout)), E, L,
                 P(ws), 0 if ws is None else ws.numel(), S())
        assert_close(host(dX2).reshape(n_seq, L, E), dX, rtol=3e-5, atol=1e-5, what=f"dX via rank-1 epilogue ws={use_ws}")
    # accumulate flag adds on top
    hip.call("ebn_attpool_bwd_dpre_f32", P(U), P(dev(q)), P(dev(np.zeros(R))), P(dqd), P(dbd), P(part), R, A, 1, S())
    assert_close(host(dqd), dq[:, 0], rtol=3e-5, atol=2e-5, what="dq accumulate(0)")


def test_attpool_epsilon_bias_is_reproduced(hip):
    """Quirk 4: exp without max-subtraction and +1e-7 in the denominator (layers.py:71-77)."""
    n_seq, L, E, A = 1, 3, 4, 2
    U = torch.full((L, A), -30.0, device="cuda")  # tanh -> -1; e = -q.sum ~ -20 -> exp ~ 2e-9
    b = torch.zeros(A, device="cuda")
    q = torch.full((A,), 10.0, device="cuda")
    X = torch.ones(L, E, device="cuda")
    o = torch.empty(1, E, device="cuda")
    w = torch.empty(L, device="cuda")
    hip.call("ebn_attpool_fwd_f32", P(U), P(b), P(q), P(X), P(o), P(w), n_seq, L, E, A, S())
    a = np.exp(np.float64(-20.0))
    want = a / (3 * a + 1e-7)
    assert_close(host(w), np.full(L, want), rtol=1e-4, atol=0, what="eps-biased weights")
    assert host(w).sum() < 0.2  # far from 1: a stabilised softmax would give 1/3 each


# ---------------------------------------------------------------- stage: encoder fwd/bwd
def _run_encoder(hip, X, Wqkv, W, b, q, h, d, A, drop_p, st, dout=None, need_dx=True):
    n_seq, L, Din = X.shape
    E = h * d
    R = n_seq * L
    t = {"X": dev(X.reshape(R, Din)), "QKV": torch.empty(R, 3 * E, device="cuda"), "Y": torch.empty(R, E, device="cuda"),
         "U": torch.empty(R, A, device="cuda"), "w": torch.empty(R, device="cuda"),
         "out": torch.empty(n_seq, E, device="cuda")}
    prm = {"Wqkv": dev(Wqkv), "W": dev(W), "b": dev(b), "q": dev(q)}
    dims = hip.EncoderDims(n_seq, L, Din, h, d, A, 1 if drop_p > 0 else -1, drop_p)
    params = hip.EncoderParams(*[t_.data_ptr() for t_ in (prm["Wqkv"], prm["W"], prm["b"], prm["q"])])
    acts = hip.EncoderActs(*[t[k].data_ptr() for k in ("X", "QKV", "Y", "U", "w", "out")])
    hip.call("ebn_encoder_fwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), None, P(st), S())
    res = {"out": host(t["out"]), "Y": host(t["Y"])}
    if dout is not None:
        g = {"dWqkv": torch.zeros(Din, 3 * E, device="cuda"), "dW": torch.zeros(E, A, device="cuda"),
             "db": torch.zeros(A, device="cuda"), "dq": torch.zeros(A, device="cuda")}
        ws_n = max(int(hip.lib().ebn_gemm_workspace_floats(Din, 3 * E, R)), int(hip.lib().ebn_gemm_workspace_floats(E, A, R)), 1)
        sc = {"dY": torch.empty(R, E, device="cuda"), "dQKV": torch.empty(R, 3 * E, device="cuda"),
              "de": torch.empty(R, device="cuda"),
              "partials": torch.empty(int(hip.lib().ebn_attpool_partials_len(R, A)), device="cuda"),
              "ws": torch.empty(ws_n, device="cuda")}
        grads = hip.EncoderGrads(*[g[k].data_ptr() for k in ("dWqkv", "dW", "db", "dq")])
        scratch = hip.EncoderScratch(sc["dY"].data_ptr(), sc["dQKV"].data_ptr(), sc["de"].data_ptr(),
                                     sc["partials"].data_ptr(), sc["ws"].data_ptr(), ws_n)
        dX = torch.empty(R, Din, device="cuda") if need_dx else None
        hip.call("ebn_encoder_bwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), P(dev(dout)),
                 ctypes.byref(grads), ctypes.byref(scratch), P(dX), 0, P(st), S())
        res.update({k: host(v) for k, v in g.items()})
        if need_dx:
            res["dX"] = host(dX).reshape(n_seq, L, Din)
    return res


@pytest.mark.parametrize("n_seq,L,Din,h,d,A,p", [(8, 30, 300, 20, 20, 200, 0.0), (8, 30, 300, 20, 20, 200, 0.2),
                                                 (5, 20, 400, 20, 20, 200, 0.0), (3, 50, 256, 16, 16, 200, 0.0),
                                                 (2, 7, 12, 3, 4, 5, 0.3)])
def test_encoder_stage_forward_backward_matches_oracle(hip, n_seq, L, Din, h, d, A, p):
    rng = np.random.default_rng(31)
    E = h * d
    X = rng.standard_normal((n_seq, L, Din)) * 0.5
    WQ, WK, WV = (on.glorot_uniform((Din, E), rng) * 2 for _ in range(3))
    W = on.glorot_uniform((E, A), rng) * 2
    b = rng.standard_normal(A) * 0.1
    q = on.glorot_uniform((A, 1), rng) * 2
    seed, step = 4, 6
    O, c_sa = on.self_attention_fwd(X, WQ, WK, WV, h, d)
    m1 = None
    Y = O
    if p > 0:
        Y, m1 = on.dropout_apply(O, on.dropout_key(seed, step, 1), p)
    out, c_al = on.att_layer2_fwd(Y, W, b, q)
    dout = rng.standard_normal((n_seq, E))
    dY, dW, db, dq = on.att_layer2_bwd(dout, c_al)
    dO = dY if m1 is None else dY * m1
    dX, dWQ, dWK, dWV = on.self_attention_bwd(dO, c_sa)
    st = make_state(seed=seed, step=step)
    got = _run_encoder(hip, X, np.concatenate([WQ, WK, WV], 1), W, b, q, h, d, A, p, st, dout)
    assert_close(got["Y"].reshape(Y.shape), Y, rtol=3e-5, atol=3e-6, what="Y")
    assert_close(got["out"], out, rtol=3e-5, atol=3e-6, what="encoder out")
    assert_close(got["dWqkv"], np.concatenate([dWQ, dWK, dWV], 1), rtol=5e-5, atol=2e-5, what="dWqkv")
    assert_close(got["dW"], dW, rtol=5e-5, atol=2e-5, what="dW")
    assert_close(got["db"], db, rtol=5e-5, atol=2e-5, what="db")
    assert_close(got["dq"], dq[:, 0], rtol=5e-5, atol=2e-5, what="dq")
    assert_close(got["dX"], dX, rtol=5e-5, atol=5e-6, what="dX")


# ---------------------------------------------------------------- a8/a9 scorer + loss
@pytest.mark.parametrize("B,C,E", [(32, 5, 400), (3, 1, 16), (2, 250, 256), (1, 70, 33)])
def test_score_forward_softmax_and_sigmoid(hip, B, C, E):
    rng = np.random.default_rng(41)
    cand = rng.standard_normal((B, C, E)) * 0.3
    user = rng.standard_normal((B, E)) * 0.3
    s = np.einsum("bce,be->bc", cand, user)
    for mode, want in ((0, on.softmax_rows(s)), (1, on.sigmoid(s))):
        sc = torch.empty(B, C, device="cuda")
        pr = torch.empty(B, C, device="cuda")
        hip.call("ebn_score_fwd_f32", P(dev(cand)), P(dev(user)), P(sc), P(pr), B, C, E, mode, S())
        assert_close(host(sc), s, rtol=2e-5, atol=2e-6, what="scores")
        assert_close(host(pr), want, rtol=3e-5, atol=1e-7, what=f"probs mode {mode}")


LOSS_KINDS = {"cross_entropy_loss": 0, "log_loss": 1, "log_loss_probs": 2}  # the C ABI's loss_kind per oracle loss name


@pytest.mark.parametrize("kind", list(LOSS_KINDS))
@pytest.mark.parametrize("B,C,E", [(32, 5, 400), (4, 9, 20)])
def test_loss_and_backward_into_representations(hip, kind, B, C, E):
    rng = np.random.default_rng(43)
    cand = rng.standard_normal((B, C, E)) * 0.4
    user = rng.standard_normal((B, E)) * 0.4
    y = np.zeros((B, C))
    y[np.arange(B), rng.integers(0, C, B)] = 1
    s = np.einsum("bce,be->bc", cand, user)
    L, ds = on.loss_fwd_bwd(s, y, kind)
    dcand = ds[..., None] * user[:, None, :]
    duser = np.einsum("bc,bce->be", ds, cand)
    rows = torch.empty(B, device="cuda")
    dc = torch.empty(B, C, E, device="cuda")
    du = torch.empty(B, E, device="cuda")
    hip.call("ebn_score_loss_bwd_f32", P(dev(cand)), P(dev(user)), P(dev(s)), P(dev(y)), P(rows), P(dc), P(du), B, C, E,
             LOSS_KINDS[kind], ctypes.c_float(1.0 / B), S())
    assert abs(host(rows).sum() - L) <= 3e-6 * max(1, abs(L))
    # (the clipped-probabilities form differentiates through the softmax: ds = p (dl/dp - sum_k p_k dl/dp_k) cancels in fp32)
    atol = 5e-7 if kind == "log_loss_probs" else 1e-7
    assert_close(host(dc), dcand, rtol=3e-5, atol=atol, what="dcand")
    assert_close(host(du), duser, rtol=3e-5, atol=atol, what="duser")
    tot = torch.zeros(1, device="cuda")
    hip.call("ebn_sum_f32", P(rows), B, ctypes.c_float(1.0), P(tot), 0, S())
    assert abs(float(tot.item()) - L) <= 3e-6 * max(1, abs(L))


@pytest.mark.parametrize("kind", list(LOSS_KINDS))
@pytest.mark.parametrize("B,C,E", [(32, 5, 400), (64, 5, 256), (3, 64, 20), (1, 1, 8), (300, 5, 400), (4, 70, 16)])
def test_fused_train_scorer_equals_the_three_kernels_and_the_oracle(hip, kind, B, C, E):
    """ebn_score_loss_train_f32: scorer + loss + backward in one launch (+ the batch-loss reduction) -- bitwise the separate
    kernels on scores / probabilities / loss rows / gradients, oracle-close on the loss."""
    rng = np.random.default_rng(B * 7 + C)
    cand = (rng.standard_normal((B, C, E)) * 0.4).astype(np.float32)
    user = (rng.standard_normal((B, E)) * 0.4).astype(np.float32)
    y = np.zeros((B, C), np.float32)
    y[np.arange(B), rng.integers(0, C, B)] = 1
    lk = LOSS_KINDS[kind]
    f = lambda *shape: torch.empty(*shape, device="cuda")
    sc0, pr0, rows0, dc0, du0, tot0 = f(B, C), f(B, C), f(B), f(B, C, E), f(B, E), torch.zeros(1, device="cuda")
    hip.call("ebn_score_fwd_f32", P(dev(cand)), P(dev(user)), P(sc0), P(pr0), B, C, E, 0, S())
    hip.call("ebn_score_loss_bwd_f32", P(dev(cand)), P(dev(user)), P(sc0), P(dev(y)), P(rows0), P(dc0), P(du0), B, C, E, lk, ctypes.c_float(1.0 / B), S())
    hip.call("ebn_sum_f32", P(rows0), B, ctypes.c_float(1.0), P(tot0), 0, S())
    sc, pr, rows, dc, du, tot = f(B, C), f(B, C), f(B), f(B, C, E), f(B, E), torch.full((1,), 9.0, device="cuda")
    hip.call("ebn_score_loss_train_f32", P(dev(cand)), P(dev(user)), P(dev(y)), P(sc), P(pr), P(rows), P(tot), P(dc), P(du), B, C, E, lk,
             ctypes.c_float(1.0 / B), S())
    for a, b, what in ((sc, sc0, "scores"), (pr, pr0, "probs"), (rows, rows0, "loss rows"), (dc, dc0, "dcand"), (du, du0, "duser")):
        assert torch.equal(a, b), what
    s64 = np.einsum("bce,be->bc", cand.astype(np.float64), user.astype(np.float64))
    L, _ = on.loss_fwd_bwd(s64, y.astype(np.float64), kind)
    assert abs(float(tot.item()) - L) <= 3e-6 * max(1, abs(L)) and abs(float(tot.item()) - float(tot0.item())) <= 1e-6 * max(1, abs(L))


@pytest.mark.parametrize("kind", list(LOSS_KINDS))
@pytest.mark.parametrize("B,L,C,E,A", [(32, 20, 5, 400, 200), (3, 50, 5, 400, 200), (5, 7, 9, 20, 12), (2, 1, 1, 8, 4), (64, 20, 5, 256, 200)])
def test_user_head_is_the_six_kernels_it_replaces(hip, kind, B, L, C, E, A):
    """ebn_user_head_train_f32: user AttLayer2 after its matmul -> scorer -> loss -> their backward up to d(pre-tanh), one
    workgroup per impression out of LDS.  Against the float64 oracle (layers.py:65-81, nrms.py:201-202, 56-67) AND against the
    separate kernels it replaces (attpool_fwd, score_loss_train, attpool_bwd_pool, attpool_bwd_dpre) on the same inputs."""
    assert hip.lib().ebn_user_head_supported(L, C, E, A) == 1
    rng = np.random.default_rng(B * 1000 + L * 10 + C)
    Upre = (rng.standard_normal((B * L, A)) * 0.7).astype(np.float32)
    bb = (rng.standard_normal(A) * 0.2).astype(np.float32)
    q = (rng.standard_normal(A) * 0.3).astype(np.float32)
    X = (rng.standard_normal((B * L, E)) * 0.5).astype(np.float32)
    cand = (rng.standard_normal((B * C, E)) * 0.4).astype(np.float32)
    y = np.zeros((B, C), np.float32)
    y[np.arange(B), rng.integers(0, C, B)] = 1
    # ---- oracle, float64
    U64, X64, c64 = Upre.astype(np.float64).reshape(B, L, A), X.astype(np.float64).reshape(B, L, E), cand.astype(np.float64).reshape(B, C, E)
    t = np.tanh(U64 + bb)
    a_ = np.exp(t @ q.astype(np.float64))
    w = a_ / (a_.sum(-1, keepdims=True) + 1e-7)
    user = (w[..., None] * X64).sum(1)
    sc = np.einsum("bce,be->bc", c64, user)
    Lw, ds = on.loss_fwd_bwd(sc, y.astype(np.float64), kind)
    dcand = ds[..., None] * user[:, None, :]
    duser = np.einsum("bc,bce->be", ds, c64)
    dw = np.einsum("ble,be->bl", X64, duser)
    de = w * (dw - (w * dw).sum(-1, keepdims=True))
    dpre = de[..., None] * q * (1 - t * t)
    dq, db = (de[..., None] * t).sum((0, 1)), dpre.sum((0, 1))
    # ---- fused head
    f = lambda *shape: torch.full(shape, 7.0, device="cuda")
    U = dev(Upre)
    o = dict(w=f(B * L), user=f(B, E), scores=f(B * C), probs=f(B * C), rows=f(B), loss=f(1), dcand=f(B * C, E), duser=f(B, E), de=f(B * L),
             dq=f(A), db=f(A), part=f(int(hip.lib().ebn_user_head_partials_len(B, A))))
    hip.call("ebn_user_head_train_f32", P(U), P(dev(bb)), P(dev(q)), P(dev(X)), P(dev(cand)), P(dev(y)), P(o["w"]), P(o["user"]), P(o["scores"]),
             P(o["probs"]), P(o["rows"]), P(o["loss"]), P(o["dcand"]), P(o["duser"]), P(o["de"]), P(o["dq"]), P(o["db"]), P(o["part"]),
             B, L, C, E, A, LOSS_KINDS[kind], ctypes.c_float(1.0 / B), S())
    tol = dict(rtol=3e-5, atol=5e-7)
    assert_close(host(o["w"]).reshape(B, L), w, rtol=2e-5, atol=1e-7, what="w")
    assert_close(host(o["user"]), user, what="user", **tol)
    assert_close(host(o["scores"]).reshape(B, C), sc, rtol=3e-5, atol=3e-6, what="scores")
    assert_close(host(o["probs"]).reshape(B, C), on.softmax_rows(sc), rtol=5e-5, atol=1e-7, what="probs")
    assert abs(float(o["loss"].item()) - Lw) <= 5e-6 * max(1, abs(Lw)) and abs(host(o["rows"]).sum() - Lw) <= 5e-6 * max(1, abs(Lw))
    g_tol = dict(rtol=1e-4, atol=2e-6 if kind == "log_loss_probs" else 1e-6)
    assert_close(host(o["dcand"]).reshape(B, C, E), dcand, what="dcand", **g_tol)
    assert_close(host(o["duser"]), duser, what="duser", **g_tol)
    assert_close(host(o["de"]).reshape(B, L), de, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(de).max(), what="de")
    assert_close(host(U).reshape(B, L, A), dpre, rtol=1e-4, atol=1e-7 + 1e-4 * np.abs(dpre).max(), what="d(pre-tanh) written over U")
    assert_close(host(o["dq"]), dq, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(dq).max(), what="dq")
    assert_close(host(o["db"]), db, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(db).max(), what="db")
    # ---- the separate kernels on the same inputs: same values to fp32 summation-order noise
    U2, w2, out2 = dev(Upre), f(B * L), f(B, E)
    hip.call("ebn_attpool_fwd_f32", P(U2), P(dev(bb)), P(dev(q)), P(dev(X)), P(out2), P(w2), B, L, E, A, S())
    sc2, pr2, rows2, tot2, dc2, du2 = f(B * C), f(B * C), f(B), f(1), f(B * C, E), f(B, E)
    hip.call("ebn_score_loss_train_f32", P(dev(cand)), P(out2), P(dev(y)), P(sc2), P(pr2), P(rows2), P(tot2), P(dc2), P(du2), B, C, E,
             LOSS_KINDS[kind], ctypes.c_float(1.0 / B), S())
    de2, dq2, db2 = f(B * L), f(A), f(A)
    part2 = f(int(hip.lib().ebn_attpool_partials_len(B * L, A)))
    hip.call("ebn_attpool_bwd_pool_f32", P(dev(X)), P(w2), P(du2), None, P(de2), B, L, E, S())
    hip.call("ebn_attpool_bwd_dpre_f32", P(U2), P(dev(q)), P(de2), P(dq2), P(db2), P(part2), B * L, A, 0, S())
    for got, ref, what in ((o["w"], w2, "w"), (o["user"], out2, "user"), (o["scores"], sc2, "scores"), (o["probs"], pr2, "probs"),
                           (o["dcand"], dc2, "dcand"), (o["duser"], du2, "duser"), (o["de"], de2, "de"), (U, U2, "dpre"), (o["dq"], dq2, "dq"),
                           (o["db"], db2, "db")):
        r = host(ref)
        assert_close(host(got), r, rtol=2e-4, atol=2e-6 + 2e-5 * np.abs(r).max(), what=f"fused head vs separate kernels: {what}")
    assert abs(float(o["loss"].item()) - float(tot2.item())) <= 2e-6 * max(1, abs(Lw))


def test_user_head_supported_says_what_fits_one_workgroup(hip):
    sup = hip.lib().ebn_user_head_supported
    assert sup(20, 5, 400, 200) == 1 and sup(50, 5, 400, 200) == 1  # history_size 20 and 50 at the reference's sizes
    assert sup(100, 5, 400, 200) == 0  # 100 x (400 + 200) floats do not fit 160 KB: the stage runs its separate kernels
    assert sup(20, 5, 402, 200) == 0 and sup(20, 5, 400, 202) == 0 and sup(0, 5, 400, 200) == 0


def test_pair_score_ragged(hip):
    rng = np.random.default_rng(47)
    nu, nn, E, n_pairs = 7, 13, 400, 101
    user = rng.standard_normal((nu, E)) * 0.2
    news = rng.standard_normal((nn, E)) * 0.2
    ui = rng.integers(0, nu, n_pairs).astype(np.int32)
    ni = rng.integers(0, nn, n_pairs).astype(np.int32)
    want = (user[ui] * news[ni]).sum(-1)
    for mode, w in ((0, want), (1, on.sigmoid(want))):
        out = torch.empty(n_pairs, device="cuda")
        hip.call("ebn_pair_score_f32", P(dev(user)), P(dev(news)), P(dev(ui, torch.int32)), P(dev(ni, torch.int32)),
                 P(out), n_pairs, E, mode, S())
        assert_close(host(out), w, rtol=2e-5, atol=2e-6, what="pair score")


# ---------------------------------------------------------------- a10 Adam (Keras form)
@pytest.mark.parametrize("n", [1, 1000, 4097, 120007])
def test_adam_keras_multi_step(hip, n):
    rng = np.random.default_rng(51)
    theta = rng.standard_normal(n)
    m = np.zeros(n)
    v = np.zeros(n)
    th_d, m_d, v_d = dev(theta), dev(m), dev(v)
    st = make_state(seed=0, step=0, lr=1e-3)
    for t in range(1, 6):
        g = rng.standard_normal(n) * (0.1 if t % 2 else 3.0)
        g[: n // 3] = 0.0  # untouched embedding rows still get the dense moment decay
        hip.call("ebn_step_advance", P(st), 0.9, 0.999, S())
        hip.call("ebn_adam_keras_step_f32", P(th_d), P(dev(g * 2.0)), P(m_d), P(v_d), n, P(st), 0.9, 0.999,
                 1e-7, ctypes.c_float(0.5), S())
        on.adam_keras_step(theta, g, m, v, t, lr=1e-3)
    assert_close(host(th_d), theta, rtol=1e-5, atol=2e-6, what="theta")
    assert_close(host(m_d), m, rtol=1e-5, atol=1e-7, what="m")
    assert_close(host(v_d), v, rtol=1e-5, atol=1e-9, what="v")


# ---------------------------------------------------------------- DocVec dense helpers
@pytest.mark.parametrize("R,C", [(160, 512), (1025, 70), (3, 5)])  # both sides of the single-launch strip limit
def test_bias_relu_forward_backward(hip, R, C):
    rng = np.random.default_rng(61)
    X = rng.standard_normal((R, C))
    b = rng.standard_normal(C) * 0.2
    Y = torch.empty(R, C, device="cuda")
    hip.call("ebn_bias_relu_f32", P(dev(X)), P(dev(b)), P(Y), R, C, S())
    want = np.maximum(X.astype(np.float32) + b.astype(np.float32), 0)
    assert np.array_equal(Y.cpu().numpy(), want)
    dY = rng.standard_normal((R, C))
    dX = torch.empty(R, C, device="cuda")
    dbias = torch.zeros(C, device="cuda")
    part = torch.empty(int(hip.lib().ebn_colsum_partials_len(R, C)), device="cuda")
    hip.call("ebn_bias_relu_bwd_f32", P(Y), P(dev(dY)), P(dX), P(dbias), P(part), R, C, 0, S())
    wdx = dY * (want > 0)
    assert_close(host(dX), wdx, rtol=1e-6, atol=1e-7, what="relu dX")
    assert_close(host(dbias), wdx.sum(0), rtol=2e-5, atol=2e-5, what="dbias")


@pytest.mark.parametrize("R,C", [(640, 96), (37, 21), (1024, 512), (1500, 40)])  # <= 1024 rows: single-launch strip kernels
@pytest.mark.parametrize("training,p", [(1, 0.0), (1, 0.2), (0, 0.0)])
def test_batchnorm_forward_backward(hip, training, p, R, C):
    rng = np.random.default_rng(67)
    X = np.maximum(rng.standard_normal((R, C)) + 0.3, 0)
    gamma = 1 + 0.1 * rng.standard_normal(C)
    beta = 0.1 * rng.standard_normal(C)
    mm0 = 0.05 * rng.standard_normal(C)
    mv0 = 1 + 0.1 * rng.random(C)
    seed, step, site, off = 3, 4, 9, 1000
    if training:
        mu, var = X.mean(0), X.var(0)
    else:
        mu, var = mm0, mv0
    istd = 1 / np.sqrt(var + 1e-3)
    xh = (X - mu) * istd
    Yw = xh * gamma + beta
    msk = np.ones_like(Yw)
    if training and p > 0:
        msk = on.dropout_keep_mask(on.dropout_key(seed, step, site), Yw.size, p, start=off).reshape(Yw.shape) / (1 - p)
    Yw = Yw * msk
    st = make_state(seed=seed, step=step)
    mm, mv = dev(mm0), dev(mv0)
    Y = torch.empty(R, C, device="cuda")
    xhat = torch.empty(R, C, device="cuda")
    mean_o = torch.empty(C, device="cuda")
    istd_o = torch.empty(C, device="cuda")
    part = torch.empty(int(hip.lib().ebn_colsum_partials_len(R, C)), device="cuda")
    hip.call("ebn_batchnorm_fwd_f32", P(dev(X)), P(dev(gamma)), P(dev(beta)), P(mm), P(mv), P(Y), P(xhat), P(mean_o),
             P(istd_o), P(part), R, C, training, P(st), site, ctypes.c_float(p), ctypes.c_int64(off), S())
    assert_close(host(Y), Yw, rtol=3e-5, atol=3e-6, what="bn Y")
    assert_close(host(istd_o), istd, rtol=2e-5, atol=0, what="istd")
    if training:
        assert_close(host(mm), mm0 * 0.99 + mu * 0.01, rtol=1e-5, atol=1e-7, what="moving mean")
        assert_close(host(mv), mv0 * 0.99 + var * 0.01, rtol=1e-5, atol=1e-7, what="moving var")
    else:
        assert_close(host(mm), mm0, rtol=0, atol=1e-7, what="moving mean untouched")
    dY = rng.standard_normal((R, C))
    g = dY * msk
    dgamma, dbeta = (g * xh).sum(0), g.sum(0)
    dxh = g * gamma
    if training:
        dXw = istd / R * (R * dxh - dxh.sum(0) - xh * (dxh * xh).sum(0))
    else:
        dXw = dxh * istd
    dX = torch.empty(R, C, device="cuda")
    dg = torch.ones(C, device="cuda")
    dbt = torch.ones(C, device="cuda")
    hip.call("ebn_batchnorm_bwd_f32", P(dev(dY)), P(xhat), P(dev(gamma)), P(istd_o), P(dX), P(dg), P(dbt), P(part), R, C,
             training, 1, P(st), site, ctypes.c_float(p), ctypes.c_int64(off), S())
    assert_close(host(dg), dgamma + 1, rtol=3e-5, atol=3e-5, what="dgamma (accumulated onto 1)")
    assert_close(host(dbt), dbeta + 1, rtol=3e-5, atol=3e-5, what="dbeta")
    assert_close(host(dX), dXw, rtol=5e-5, atol=5e-6, what="bn dX")


def test_axpy_and_sumsq(hip):
    rng = np.random.default_rng(71)
    n = 262145
    x = rng.standard_normal(n)
    y = rng.standard_normal(n)
    yd = dev(y)
    hip.call("ebn_axpy_f32", ctypes.c_float(0.25), P(dev(x)), P(yd), n, S())
    assert_close(host(yd), 0.25 * x + y, rtol=1e-6, atol=1e-6, what="axpy")
    out = torch.full((1,), 2.0, device="cuda")
    hip.call("ebn_sumsq_f32", P(dev(x)), n, ctypes.c_float(0.5), P(out), 1, S())
    assert abs(float(out.item()) - (2.0 + 0.5 * (x.astype(np.float32).astype(np.float64) ** 2).sum())) < 2e-4 * n ** 0.5


# ---------------------------------------------------------------- a13 device-side batch assembly
def test_expand_titles_is_the_loader_gather_bit_exact(hip):
    rng = np.random.default_rng(81)
    n_rows, T, n_titles = 300, 30, 800
    matrix = rng.integers(0, 250002, (n_rows, T)).astype(np.int32)
    matrix[0] = 0  # the unknown / padded article
    idx = rng.integers(0, n_rows, n_titles).astype(np.int32)
    idx[:50] = 0
    out = torch.full((n_titles * T,), -7, dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("ebn_expand_titles_i32", P(dev(idx, torch.int32)), P(dev(matrix, torch.int32)), P(out), n_titles, T, n_rows, P(flag), S())
    assert np.array_equal(out.cpu().numpy().reshape(n_titles, T), matrix[idx]) and int(flag.item()) == 0
    idx[3] = n_rows
    hip.call("ebn_expand_titles_i32", P(dev(idx, torch.int32)), P(dev(matrix, torch.int32)), P(out), n_titles, T, n_rows, P(flag), S())
    assert int(flag.item()) == 1 and (out.cpu().numpy().reshape(n_titles, T)[3] == 0).all()


def test_fixed_point_scatter_is_order_independent_and_accurate(hip):
    rng = np.random.default_rng(91)
    V, D, n_tok = 64, 20, 6000
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    ids[:3000] = 0  # a very hot row
    dX = (rng.standard_normal((n_tok, D)) * 10 ** rng.uniform(-6, 0, (n_tok, 1))).astype(np.float32)
    want = on.embedding_bwd(ids, dX.astype(np.float64), V)
    outs = []
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for perm in (np.arange(n_tok), rng.permutation(n_tok)):  # same multiset of (id, grad row), different arrival order
        acc = torch.zeros(V, D, dtype=torch.int64, device="cuda")
        g = torch.full((V, D), 3.0, device="cuda")
        hip.call("ebn_embedding_grad_scatter_fixed", P(dev(ids[perm], torch.int32)), P(dev(dX[perm])), P(acc), n_tok, D, V, None, -1,
                 ctypes.c_float(0.0), P(flag), S())
        hip.call("ebn_fixed_to_f32", P(acc), P(g), V * D, P(flag), S())
        assert int(acc.abs().max().item()) == 0  # accumulator is left zeroed for the next step
        outs.append(g.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])  # bitwise, whatever the order
    assert_close(outs[0], want, rtol=2e-7, atol=1e-9, what="fixed-point dTable")
    assert int(flag.item()) == 0


def test_fixed_point_range_is_guarded_and_fused_adam_equals_the_two_kernels(hip):
    """|sum| < 2^23 is the accumulator's range: a term >= 2^21 or an accumulated |sum| >= 2^22 raises the range flag
    (never a silent wrap); ebn_adam_keras_step_fixed_f32 == ebn_fixed_to_f32 followed by ebn_adam_keras_step_f32, bit
    for bit, and leaves the accumulator zeroed."""
    rng = np.random.default_rng(17)
    V, D, n_tok = 33, 7, 500  # odd element count: the 2-wide fused kernel has a tail
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    dX = rng.standard_normal((n_tok, D)).astype(np.float32)
    st = make_state(seed=3, step=4, lr=1e-3, alpha=7e-4)
    th0, m0, v0 = rng.standard_normal((V, D)).astype(np.float32), rng.standard_normal((V, D)).astype(np.float32) * 0.1, rng.random((V, D)).astype(np.float32) * 0.01
    res = []
    for fused in (False, True):
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        acc = torch.zeros(V, D, dtype=torch.int64, device="cuda")
        th, m, v = dev(th0), dev(m0), dev(v0)
        hip.call("ebn_embedding_grad_scatter_fixed", P(dev(ids, torch.int32)), P(dev(dX)), P(acc), n_tok, D, V, None, -1, ctypes.c_float(0.0), P(flag), S())
        if fused:
            hip.call("ebn_adam_keras_step_fixed_f32", P(th), P(acc), P(m), P(v), V * D, P(st), 0.9, 0.999, 1e-7, ctypes.c_float(0.5), P(flag), S())
        else:
            g = torch.empty(V, D, device="cuda")
            hip.call("ebn_fixed_to_f32", P(acc), P(g), V * D, P(flag), S())
            hip.call("ebn_adam_keras_step_f32", P(th), P(g), P(m), P(v), V * D, P(st), 0.9, 0.999, 1e-7, ctypes.c_float(0.5), S())
        assert int(acc.abs().max().item()) == 0 and int(flag.item()) == 0
        res.append([t.cpu().numpy() for t in (th, m, v)])
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    # range guard: a huge term, then a sum of large-but-legal terms that leaves the safe range
    for vals, expect in (([3.0e6], 1), ([1.5e6] * 4, 1), ([1.0e6] * 2, 0), ([float("nan")], 1)):
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        acc = torch.zeros(2, 4, dtype=torch.int64, device="cuda")
        n = len(vals)
        dXv = np.zeros((n, 4), np.float32)
        dXv[:, 1] = vals
        hip.call("ebn_embedding_grad_scatter_fixed", P(dev(np.ones(n), torch.int32)), P(dev(dXv)), P(acc), n, 4, 2, None, -1, ctypes.c_float(0.0), P(flag), S())
        g = torch.empty(2, 4, device="cuda")
        hip.call("ebn_fixed_to_f32", P(acc), P(g), 8, P(flag), S())
        assert int(flag.item()) == expect, (vals, int(flag.item()))


@pytest.mark.parametrize("R0,R1,C,p", [(640, 160, 512, 0.2), (7, 3, 20, 0.0), (1023, 1, 40, 0.5), (5, 0, 16, 0.2), (0, 9, 33, 0.2)])
def test_two_site_batchnorm_launches_equal_the_per_site_calls(hip, R0, R1, C, p):
    """ebn_batchnorm2_fwd_f32 == two ebn_batchnorm_fwd_f32 (history site, then candidate site), and
    ebn_batchnorm2_relu_bwd_f32 == two ebn_batchnorm_bwd_f32 + ebn_bias_relu_bwd_f32, to the last bits."""
    rng = np.random.default_rng(R0 + C)
    N = R0 + R1
    X = np.maximum(rng.standard_normal((N, C)) + 0.3, 0).astype(np.float32)
    gamma, beta = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    mm0, mv0 = (0.05 * rng.standard_normal(C)).astype(np.float32), (1 + 0.1 * rng.random(C)).astype(np.float32)
    dY = rng.standard_normal((N, C)).astype(np.float32)
    st = make_state(seed=3, step=4)
    site = 9
    f = lambda *shape: torch.empty(*shape, device="cuda")
    part = f(int(hip.lib().ebn_colsum_partials_len(N, C)))
    # reference: per-site entry points
    mm, mv, Y, xh = dev(mm0), dev(mv0), f(N, C), f(N, C)
    stats = [[f(C), f(C)], [f(C), f(C)]]
    dX, dg, db, dbias = f(N, C), f(C), f(C), f(C)
    Xd, dYd = dev(X), dev(dY)
    first = True
    for k, (r0, nr) in enumerate(((0, R0), (R0, R1))):
        if nr == 0:
            continue
        hip.call("ebn_batchnorm_fwd_f32", P(Xd[r0:]), P(dev(gamma)), P(dev(beta)), P(mm), P(mv), P(Y[r0:]), P(xh[r0:]), P(stats[k][0]),
                 P(stats[k][1]), P(part), nr, C, 1, P(st), site, ctypes.c_float(p), ctypes.c_int64(r0 * C), S())
    for k, (r0, nr) in enumerate(((0, R0), (R0, R1))):
        if nr == 0:
            continue
        hip.call("ebn_batchnorm_bwd_f32", P(dYd[r0:]), P(xh[r0:]), P(dev(gamma)), P(stats[k][1]), P(dX[r0:]), P(dg), P(db), P(part), nr, C, 1,
                 0 if first else 1, P(st), site, ctypes.c_float(p), ctypes.c_int64(r0 * C), S())
        first = False
    hip.call("ebn_bias_relu_bwd_f32", P(Xd), P(dX), P(dX), P(dbias), P(part), N, C, 0, S())
    # two-site entry points
    mm2, mv2, Y2, xh2 = dev(mm0), dev(mv0), f(N, C), f(N, C)
    stats2 = [[torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")] for _ in range(2)]
    hip.call("ebn_batchnorm2_fwd_f32", P(Xd), P(dev(gamma)), P(dev(beta)), P(mm2), P(mv2), P(Y2), P(xh2), P(stats2[0][0]), P(stats2[0][1]),
             P(stats2[1][0]), P(stats2[1][1]), R0, R1, C, P(st), site, ctypes.c_float(p), S())
    dX2, dg2, db2, dbias2 = f(N, C), f(C), f(C), f(C)
    hip.call("ebn_batchnorm2_relu_bwd_f32", P(dYd), P(xh2), P(Xd), P(dev(gamma)), P(stats2[0][1]), P(stats2[1][1]), P(dX2), P(dg2), P(db2),
             P(dbias2), R0, R1, C, P(st), site, ctypes.c_float(p), S())
    # different kernels, same formulae: column sums in another order, multiply-adds contracted differently -> last bits
    for a, b, what in ((Y2, Y, "Y"), (xh2, xh, "xhat"), (mm2, mm, "moving mean"), (mv2, mv, "moving var"), (dX2, dX, "dX"), (dg2, dg, "dgamma"),
                       (db2, db, "dbeta")):
        assert_close(host(a), host(b), rtol=2e-5, atol=2e-5, what=what)
    for k, nr in enumerate((R0, R1)):
        if nr:
            assert_close(host(stats2[k][0]), host(stats[k][0]), rtol=1e-5, atol=1e-6, what="site mean")
            assert_close(host(stats2[k][1]), host(stats[k][1]), rtol=1e-5, atol=1e-6, what="site istd")
    assert_close(host(dbias2), host(dbias), rtol=1e-5, atol=1e-5, what="dbias")
    assert hip.lib().ebn_batchnorm2_fwd_f32(P(Xd), P(dev(gamma)), P(dev(beta)), P(mm2), P(mv2), P(Y2), P(xh2), P(stats2[0][0]), P(stats2[0][1]),
                                            P(stats2[1][0]), P(stats2[1][1]), 1024, 1, C, P(st), site, ctypes.c_float(p), S()) == -2


@pytest.mark.parametrize("M,N,K", [(800, 512, 768), (160, 256, 512), (37, 20, 12), (1300, 64, 40), (5, 7, 3)])
def test_dense_relu_forward_is_gemm_bias_relu(hip, M, N, K):
    """Dense(units, activation="relu") forward in one pass: epilogue of the GEMM or of its split-K reduce."""
    rng = np.random.default_rng(M + N + K)
    A, B, b = rng.standard_normal((M, K)), rng.standard_normal((K, N)) / np.sqrt(K), rng.standard_normal(N) * 0.3
    want = np.maximum(A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64) + b.astype(np.float32), 0)
    for use_ws in (False, True):
        ws = torch.empty(max(int(hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda") if use_ws else None
        C = torch.full((M, N), -3.0, device="cuda")
        hip.call("ebn_dense_relu_fwd_f32", M, N, K, P(dev(A)), K, P(dev(B)), N, P(dev(b)), P(C), N, P(ws),
                 0 if ws is None else ws.numel(), S())
        assert_close(host(C), want, rtol=2e-6, atol=1e-5 + 3e-7 * K, what=f"dense relu {M}x{N}x{K} ws={use_ws}")
        assert (host(C) >= 0).all()


def test_l2_regulariser_of_a_stack_in_two_launches(hip):
    rng = np.random.default_rng(5)
    Ws = [rng.standard_normal(n).astype(np.float32) for n in (768 * 512, 1000, 7)]
    gs = [rng.standard_normal(w.size).astype(np.float32) for w in Ws]
    lam = 1e-3
    loss = torch.full((1,), 2.5, device="cuda")
    part = torch.empty(1024, device="cuda")
    dW, dg = [dev(w) for w in Ws], [dev(g) for g in gs]
    hip.call("ebn_l2_reg4_f32", P(dW[0]), P(dg[0]), Ws[0].size, P(dW[1]), P(dg[1]), Ws[1].size, None, None, 0, P(dW[2]), P(dg[2]),
             Ws[2].size, ctypes.c_float(lam), P(part), P(loss), S())
    for w, g, d in zip(Ws, gs, dg):
        assert_close(host(d), g.astype(np.float64) + 2 * lam * w.astype(np.float64), rtol=1e-6, atol=1e-7, what="gW += 2 lambda W")
    want = 2.5 + lam * sum(float((w.astype(np.float64) ** 2).sum()) for w in Ws)
    assert abs(float(loss.item()) - want) <= 1e-5 * want


def test_streaming_auc_histogram_kernel_equals_the_numpy_accumulation(hip):
    """ebn_auc_hist_f32 (one launch per batch, device labels) == StreamingAUC.update_numpy bucket for bucket, including
    predictions exactly on a threshold, 0, 1 and ties; and through StreamingAUC.update_device end to end."""
    from ebrec.models.newsrec.callbacks import StreamingAUC

    rng = np.random.default_rng(71)
    n = 5003
    p = rng.random(n).astype(np.float32)
    p[:300] = (np.arange(300) / 199.0).astype(np.float32)[:300] % 1.0  # values on / next to the 200 thresholds
    p[300:310] = 0.0
    p[310:320] = 1.0
    p[320:340] = 0.5
    y = (rng.random(n) < 0.2).astype(np.float32)
    ref, got = StreamingAUC(), StreamingAUC()
    ref.update_numpy(y, p)
    for s in range(0, n, 1000):  # several "batches"
        got.update_device(dev(y[s:s + 1000]), dev(p[s:s + 1000]))
    assert np.array_equal(got._dev[1].cpu().numpy().astype(np.float64), ref.pos_hist)
    assert np.array_equal(got._dev[2].cpu().numpy().astype(np.float64), ref.neg_hist)
    assert got.result() == ref.result()


@pytest.mark.parametrize("R,D,V,p", [(24000, 1024, 5000, 0.2), (700, 300, 97, 0.0), (130, 36, 50, 0.5), (130, 37, 50, 0.5), (333, 2052, 40, 0.2)])
def test_gather_split_planes_is_the_gather_then_the_split(hip, R, D, V, p):
    """ebn_gather_split_planes_f32 (Embedding + Dropout of a training step in split precision): the planes it writes are, bit for
    bit, the planes ebn_split_planes_f32 makes of ebn_gather_rows_f32's output (same rows, same dropout mask) in both
    orientations -- and the two projection products computed from them equal the generic split GEMM on the fp32 rows."""
    rng = np.random.default_rng(R + D)
    table = rng.standard_normal((V, D)).astype(np.float32)
    ids = rng.integers(0, V, R).astype(np.int32)
    ids[:5] = 0
    st = make_state(seed=3, step=2)
    L = hip.lib()
    d_ids, d_tab = torch.from_numpy(ids).cuda(), dev(table)
    X = torch.empty(R, D, device="cuda")
    hip.call("ebn_gather_rows_f32", P(d_ids), P(d_tab), P(X), R, D, V, P(st), 0, ctypes.c_float(p), None, S())
    X += 0.0  # (the scalar gather of an odd D writes a dropped negative element as -0.0, x * 0; the planes carry +0.0)
    u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device="cuda")
    refN, refT = u8(L.ebn_planes_bytes(R, D)), u8(L.ebn_planes_bytes(D, R))
    hip.call("ebn_split_planes_f32", P(X), D, R, D, 0, P(refN), S())
    hip.call("ebn_split_planes_f32", P(X), D, D, R, 1, P(refT), S())
    gotN, gotT = u8(L.ebn_planes_bytes(R, D)) + 7, u8(L.ebn_planes_bytes(D, R)) + 7  # poisoned: the kernel must write the zero padding too
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("ebn_gather_split_planes_f32", P(d_ids), P(d_tab), R, D, V, P(st), 0, ctypes.c_float(p), P(flag), P(gotN), P(gotT), S())
    assert torch.equal(gotN, refN) and torch.equal(gotT, refT) and int(flag.item()) == 0
    # the planes are an exact three-way split: plane0 + plane1 + plane2 == the fp32 value (checked through a product with the identity-like W)
    E3 = 48
    W = rng.standard_normal((D, E3)).astype(np.float32)
    dQ = rng.standard_normal((R, E3)).astype(np.float32)
    Wp, dQp = u8(L.ebn_planes_bytes(E3, D)), u8(L.ebn_planes_bytes(E3, R))
    hip.call("ebn_split_planes_f32", P(dev(W)), E3, E3, D, 1, P(Wp), S())
    hip.call("ebn_split_planes_f32", P(dev(dQ)), E3, E3, R, 1, P(dQp), S())
    part = torch.empty(max(int(L.ebn_gemm_planes_workspace_floats(D, E3, R)), int(L.ebn_gemm_planes_workspace_floats(R, E3, D)), 1), device="cuda")
    QKV, dW = torch.empty(R, E3, device="cuda"), torch.empty(D, E3, device="cuda")
    hip.call("ebn_gemm_planes_f32", P(gotN), R, P(Wp), E3, D, ctypes.c_float(1.0), ctypes.c_float(0.0), P(QKV), E3, P(part), part.numel(), S())
    hip.call("ebn_gemm_planes_f32", P(gotT), D, P(dQp), E3, R, ctypes.c_float(1.0), ctypes.c_float(0.0), P(dW), E3, P(part), part.numel(), S())
    X64 = host(X)
    assert_close(host(QKV), X64 @ W.astype(np.float64), rtol=2e-6, atol=1e-5 + 3e-7 * D, what="X.W from the gather's planes")
    assert_close(host(dW), X64.T @ dQ.astype(np.float64), rtol=2e-6, atol=1e-5 + 3e-7 * R, what="X^T.dQ from the gather's transposed planes")
    ids[7] = V  # an id outside the table raises the flag, like the fp32 gather
    hip.call("ebn_gather_split_planes_f32", P(torch.from_numpy(ids).cuda()), P(d_tab), R, D, V, P(st), 0, ctypes.c_float(p), P(flag), P(gotN), P(gotT), S())
    assert int(flag.item()) == 1


@pytest.mark.parametrize("ids_kind", ["hot", "zipf", "one_id", "uniform"])
@pytest.mark.parametrize("n_tok,D,V,p", [(24000, 300, 32000, 0.2), (52800, 300, 32000, 0.0), (700, 64, 50, 0.5), (5, 7, 1000, 0.0), (4099, 33, 3, 0.2),
                                         (130, 1100, 70, 0.2), (64, 4, 9, 0.0), (65, 1024, 250002, 0.2)])
def test_duplicate_combining_gradient_accumulation_is_the_atomic_one_bit_for_bit(hip, n_tok, D, V, p, ids_kind):
    """ebn_embedding_grad_scatter_fixed (the duplicates of every 64 consecutive tokens combined in registers, one atomic per
    distinct id and column) against ebn_embedding_grad_scatter_fixed_atomic (one 64-bit atomic per element): the sums are
    2^40-scaled integers either way -> identical accumulators.  Hot rows (padded titles: runs of id 0, SURVEY 8(d) Z: Zipf ids),
    every token on one row, ids outside the table (skipped by both), chunks that end in a partial run (n_tok % 64 != 0), D beyond
    one pass of a 1024-thread workgroup, two calls in a row accumulating into the same buffer (the sparse data-parallel
    exchange does that)."""
    from bench import zipf_ids

    rng = np.random.default_rng(n_tok + D)
    if ids_kind == "zipf":
        ids = zipf_ids(rng, n_tok, V).astype(np.int32)
    elif ids_kind == "one_id":
        ids = np.full(n_tok, V // 2, np.int32)
    else:
        ids = rng.integers(0, V, n_tok).astype(np.int32)
    if n_tok > 100 and ids_kind in ("hot", "zipf"):
        pad = (rng.random(n_tok // 30 + 1) < 0.15).repeat(30)[:n_tok]   # padded titles: 30 consecutive tokens of id 0
        ids[pad] = 0
        ids[10:14] = [V, -1, V + 5, -7]        # out of range: skipped
        ids[20:60] = V - 1
    dX = rng.standard_normal((n_tok, D)).astype(np.float32)
    st = make_state(seed=4, step=3)
    d_ids, d_dx = torch.from_numpy(ids).cuda(), dev(dX)
    acc_a = torch.zeros(V, D, dtype=torch.int64, device="cuda")
    acc_s = torch.zeros(V, D, dtype=torch.int64, device="cuda")
    fa, fs = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(2):
        hip.call("ebn_embedding_grad_scatter_fixed_atomic", P(d_ids), P(d_dx), P(acc_a), n_tok, D, V, P(st), 0, ctypes.c_float(p), P(fa), S())
        hip.call("ebn_embedding_grad_scatter_fixed", P(d_ids), P(d_dx), P(acc_s), n_tok, D, V, P(st), 0, ctypes.c_float(p), P(fs), S())
        assert torch.equal(acc_a, acc_s), rep
    assert int(fa.item()) == 0 and int(fs.item()) == 0
    # against the oracle's dense gradient (float64), through the fixed-point scale
    mask = on.dropout_keep_mask(on.dropout_key(4, 3, 0), n_tok * D, p).reshape(n_tok, D) / (1.0 - p) if p > 0 else 1.0
    ok = (ids >= 0) & (ids < V)
    want = on.embedding_bwd(ids[ok], (dX * mask)[ok].astype(np.float64), V) * 2
    got = acc_s.cpu().numpy().astype(np.float64) / 2.0 ** 40
    assert_close(got, want, rtol=1e-6, atol=1e-6 + 1e-6 * np.abs(want).max(), what="duplicate-combining accumulator vs dense gradient")
    big = dX.copy()
    big[3, 1] = 3e6  # a term beyond 2^21: the range flag, like the atomic form
    hip.call("ebn_embedding_grad_scatter_fixed", P(d_ids), P(dev(big)), P(acc_s), n_tok, D, V, None, -1, ctypes.c_float(0.0), P(fs), S())
    assert int(fs.item()) == (1 if (0 <= ids[3] < V) else 0)


# ---------------------------------------------------------------- the finishing passes of a step's backward as one launch
def _finish_job(hip, kind, n_parts, rows, cols, partials, out0, out1=None, ld=0, beta=0.0, scale=1.0, loss_rows=None, loss_out=None):
    dp = lambda t: None if t is None else t.data_ptr()
    return hip.FinishJob(kind, n_parts, rows, cols, dp(partials), dp(out0), dp(out1), ld, beta, scale, dp(loss_rows), dp(loss_out))


@pytest.mark.parametrize("tA,tB,M,N,K", [(1, 0, 400, 200, 24000), (1, 0, 1024, 1200, 6000), (0, 0, 64, 64, 16), (1, 0, 300, 1200, 5000), (0, 1, 640, 400, 200),
                                         (0, 0, 4096, 200, 400)])
def test_gemm_partials_plus_finishing_pass_is_the_gemm_bit_for_bit(hip, tA, tB, M, N, K):
    """ebn_gemm_f32_partials leaves alpha * op(A).op(B) as n_parts dense slices (the split-K partials without their combining launch, or
    the product itself); ebn_grad_finish_f32 (EBN_FINISH_SPLITK) sums them in the order of the stand-alone combine: the same bits as
    ebn_gemm_f32_ws on the same workspace size -- split (weight gradients), unsplit, small-output and LDS-free kernels."""
    rng = np.random.default_rng(M + N + K)
    A = dev(rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32))
    B = dev(rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32))
    n_ws = int(hip.lib().ebn_gemm_partials_workspace_floats(M, N, K))
    assert n_ws >= M * N and n_ws % (M * N) == 0
    ws, ws2 = torch.empty(n_ws, device="cuda"), torch.empty(n_ws, device="cuda")
    C1, C2 = torch.full((M, N), 3.0, device="cuda"), torch.full((M, N + 8), 5.0, device="cuda")
    gemm(tA, tB, M, N, K, 0.5, A, A.shape[1], B, B.shape[1], 0.0, C1, N, ws)
    n = ctypes.c_int32(-1)
    hip.call("ebn_gemm_f32_partials", tA, tB, M, N, K, ctypes.c_float(0.5), P(A), A.shape[1], P(B), B.shape[1], P(ws2), n_ws, ctypes.byref(n), S())
    assert 1 <= n.value <= n_ws // (M * N)
    jobs = (hip.FinishJob * 1)(_finish_job(hip, hip.FINISH_SPLITK, n.value, M, N, ws2, C2, ld=N + 8))
    hip.call("ebn_grad_finish_f32", jobs, 1, S())
    assert torch.equal(C2[:, :N], C1) and bool((C2[:, N:] == 5.0).all())  # same bits; the padding of a strided C untouched
    ref = 0.5 * ((A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double()))
    assert_close(host(C2[:, :N]), ref.cpu().numpy(), rtol=2e-6, atol=1e-5 + 3e-7 * K, what="partials + finish vs float64")
    # beta != 0: C = sum + beta * C
    C3 = C1.clone()
    jobs = (hip.FinishJob * 1)(_finish_job(hip, hip.FINISH_SPLITK, n.value, M, N, ws2, C3, ld=N, beta=2.0))
    hip.call("ebn_grad_finish_f32", jobs, 1, S())
    assert_close(host(C3), 3.0 * host(C1), rtol=1e-6, atol=1e-6, what="finish with beta")


def test_one_finishing_launch_equals_the_four_stand_alone_passes(hip):
    """ebn_grad_finish_f32 with the jobs of a c2-shaped step -- two split-K sums, the AttLayer2 d(q) / d(b) column sums of 24000 rows, the
    per-impression head's d(q) / d(b) / loss sums of 32 impressions -- against the producers run WITH their own finishing launches:
    every output bit for bit (the device bodies and summation orders are shared)."""
    rng = np.random.default_rng(77)
    R, A, E, B, L, C = 24000, 200, 400, 32, 20, 5
    f = lambda *shape: torch.full(shape, 7.0, device="cuda")
    # AttLayer2 backward step 2 (news level): stand-alone vs deferred
    U0 = np.tanh(rng.standard_normal((R, A))).astype(np.float32)
    q, de = (rng.standard_normal(A) * 0.3).astype(np.float32), rng.standard_normal(R).astype(np.float32)
    n_part = int(hip.lib().ebn_attpool_partials_len(R, A))
    Ua, Ub, pa, pb = dev(U0), dev(U0), f(n_part), f(n_part)
    dq_a, db_a, dq_b, db_b = f(A), f(A), f(A), f(A)
    hip.call("ebn_attpool_bwd_dpre_f32", P(Ua), P(dev(q)), P(dev(de)), P(dq_a), P(db_a), P(pa), R, A, 0, S())
    hip.call("ebn_attpool_bwd_dpre_f32", P(Ub), P(dev(q)), P(dev(de)), None, None, P(pb), R, A, 0, S())
    assert torch.equal(Ua, Ub) and torch.equal(pa, pb) and bool((dq_b == 7.0).all())  # the deferred call wrote the partials, not d(q)
    # the per-impression head: stand-alone vs deferred
    Upre = (rng.standard_normal((B * L, A)) * 0.7).astype(np.float32)
    bb, X = (rng.standard_normal(A) * 0.2).astype(np.float32), (rng.standard_normal((B * L, E)) * 0.5).astype(np.float32)
    cand = (rng.standard_normal((B * C, E)) * 0.4).astype(np.float32)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]

    def head(defer):
        o = dict(U=dev(Upre), w=f(B * L), user=f(B, E), scores=f(B * C), probs=f(B * C), rows=f(B), loss=f(1), dcand=f(B * C, E), duser=f(B, E), de=f(B * L),
                 dq=f(A), db=f(A), part=f(int(hip.lib().ebn_user_head_partials_len(B, A))))
        hip.call("ebn_user_head_train_f32", P(o["U"]), P(dev(bb)), P(dev(q)), P(dev(X)), P(dev(cand)), P(dev(y)), P(o["w"]), P(o["user"]), P(o["scores"]),
                 P(o["probs"]), P(o["rows"]), P(o["loss"]), P(o["dcand"]), P(o["duser"]), P(o["de"]), None if defer else P(o["dq"]), None if defer else P(o["db"]),
                 P(o["part"]), B, L, C, E, A, 0, ctypes.c_float(1.0 / B), S())
        return o

    ha, hb = head(False), head(True)
    assert torch.equal(ha["part"], hb["part"]) and torch.equal(ha["rows"], hb["rows"]) and float(hb["loss"].item()) == 7.0
    # two weight-gradient GEMMs: stand-alone vs partials
    Y, dpre = dev(rng.standard_normal((R, E)).astype(np.float32)), dev(rng.standard_normal((R, A)).astype(np.float32))
    Xs, dQ = dev(rng.standard_normal((6000, 256)).astype(np.float32)), dev(rng.standard_normal((6000, 1200)).astype(np.float32))
    dW_a, dW_b, dWq_a, dWq_b = f(E, A), f(E, A), f(256, 1200), f(256, 1200)
    n1, n2 = int(hip.lib().ebn_gemm_partials_workspace_floats(E, A, R)), int(hip.lib().ebn_gemm_partials_workspace_floats(256, 1200, 6000))
    w1, w2, wa = torch.empty(n1, device="cuda"), torch.empty(n2, device="cuda"), torch.empty(max(n1, n2), device="cuda")
    gemm(1, 0, E, A, R, 1.0, Y, E, dpre, A, 0.0, dW_a, A, wa[:n1])
    gemm(1, 0, 256, 1200, 6000, 1.0, Xs, 256, dQ, 1200, 0.0, dWq_a, 1200, wa[:n2])
    p1, p2 = ctypes.c_int32(), ctypes.c_int32()
    hip.call("ebn_gemm_f32_partials", 1, 0, E, A, R, ctypes.c_float(1.0), P(Y), E, P(dpre), A, P(w1), n1, ctypes.byref(p1), S())
    hip.call("ebn_gemm_f32_partials", 1, 0, 256, 1200, 6000, ctypes.c_float(1.0), P(Xs), 256, P(dQ), 1200, P(w2), n2, ctypes.byref(p2), S())
    assert p1.value > 1 and p2.value > 1
    jobs = (hip.FinishJob * 4)(
        _finish_job(hip, hip.FINISH_SPLITK, p2.value, 256, 1200, w2, dWq_b, ld=1200),
        _finish_job(hip, hip.FINISH_SPLITK, p1.value, E, A, w1, dW_b, ld=A),
        _finish_job(hip, hip.FINISH_COLRED, n_part // (2 * A), 1, A, pb, dq_b, db_b, ld=A),
        _finish_job(hip, hip.FINISH_HEAD, 1, B, A, hb["part"], hb["dq"], hb["db"], ld=A, loss_rows=hb["rows"], loss_out=hb["loss"]))
    hip.call("ebn_grad_finish_f32", jobs, 4, S())
    for got, ref, what in ((dWq_b, dWq_a, "dWqkv"), (dW_b, dW_a, "dW"), (dq_b, dq_a, "news d(q)"), (db_b, db_a, "news d(b)"), (hb["dq"], ha["dq"], "user d(q)"),
                           (hb["db"], ha["db"], "user d(b)"), (hb["loss"], ha["loss"], "batch loss")):
        assert torch.equal(got, ref), what
    # an empty list and empty jobs are no-ops; a malformed job is refused
    hip.call("ebn_grad_finish_f32", None, 0, S())
    jobs = (hip.FinishJob * 1)(_finish_job(hip, hip.FINISH_SPLITK, 3, 0, 200, w1, dW_b, ld=A))
    hip.call("ebn_grad_finish_f32", jobs, 1, S())
    bad = (hip.FinishJob * 1)(_finish_job(hip, 7, 1, 4, 4, w1, dW_b, ld=4))
    with pytest.raises(hip.HipError):
        hip.call("ebn_grad_finish_f32", bad, 1, S())


# ---------------------------------------------------------------- round 5: grouped TN products, row-mapped A, the DocVec prologue
@pytest.mark.parametrize("shapes", [[(48, 40, 96), (20, 36, 96)],                                  # 32 x 32 tiles, partial tiles, one slab
                                    [(768, 512, 800), (512, 512, 800), (512, 512, 800), (512, 256, 800)],  # the c3 group: 64 x 64 tiles
                                    [(132, 68, 1000), (8, 4, 5), (64, 64, 128)]])
def test_tn_group_products_column_sums_and_l2_term(hip, shapes):
    """ebn_gemm_tn_group_f32: C_i = A_i^T . B_i for every problem of the group in one launch, colsum_i = column sums of B_i (the bias
    gradient of a Dense layer), C_i += two_lambda * W_i (kernel_regularizer=l2) -- against float64."""
    from ebrec import _hip

    rng = np.random.default_rng(len(shapes) * 7 + shapes[0][0])
    probs = (_hip.TnProblem * len(shapes))()
    keep, want = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B, W = (rng.standard_normal(s).astype(np.float32) for s in ((K, M), (K, N), (M, N)))
        dA, dB, dW = dev(A), dev(B), dev(W)
        C, cs = torch.full((M, N), float("nan"), device="cuda"), torch.full((N,), float("nan"), device="cuda")
        q = probs[i]
        q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc = M, N, K, dA.data_ptr(), M, dB.data_ptr(), N, C.data_ptr(), N
        lam = 0.25 if i % 2 == 0 else 0.0
        if i != 1:
            q.colsum = cs.data_ptr()
        if lam:
            q.l2_W, q.two_lambda = dW.data_ptr(), lam
        keep += [dA, dB, dW, C, cs]
        want.append((A.astype(np.float64).T @ B.astype(np.float64) + lam * W, B.astype(np.float64).sum(0), i != 1))
    hip.call("ebn_gemm_tn_group_f32", probs, len(shapes), S())
    for i, (c_ref, cs_ref, has_cs) in enumerate(want):
        C, cs = keep[5 * i + 3], keep[5 * i + 4]
        assert_close(host(C), c_ref, rtol=2e-5, atol=2e-5 * np.abs(c_ref).max(), what=f"C of problem {i}")
        if has_cs:
            assert_close(host(cs), cs_ref, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(cs_ref).max()), what=f"column sums of problem {i}")
    bad = (_hip.TnProblem * 1)()
    bad[0].M, bad[0].N, bad[0].K = 6, 8, 16  # M % 4 != 0: the group takes aligned problems only
    bad[0].A = bad[0].B = bad[0].C = keep[0].data_ptr()
    bad[0].lda, bad[0].ldb, bad[0].ldc = 6, 8, 8
    assert hip.lib().ebn_gemm_tn_group_f32(bad, 1, S()) == -2


@pytest.mark.parametrize("M,N,K,V", [(270, 1200, 300, 500), (1000, 64, 1024, 77), (256, 132, 40, 1000)])
def test_row_mapped_projection_equals_gather_then_gemm(hip, M, N, K, V):
    """ebn_gemm_f32_rowmap: C = table[ids] . B with the rows fetched table -> LDS inside the GEMM -- against float64, with a partial
    last 16-deep slab (K = 300, 40), ragged M / N, repeated and boundary ids; an id outside the table raises the flag; fewer than
    256 rows are EBN_ERR_UNSUPPORTED (the caller gathers first)."""
    rng = np.random.default_rng(M + N)
    table, B = rng.standard_normal((V, K)).astype(np.float32), rng.standard_normal((K, N)).astype(np.float32)
    ids = rng.integers(0, V, M).astype(np.int32)
    ids[0], ids[1], ids[-1] = 0, V - 1, ids[2]
    C = torch.full((M, N), float("nan"), device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    dt, dB, di = dev(table), dev(B), dev(ids, torch.int32)
    hip.call("ebn_gemm_f32_rowmap", P(di), V, M, N, K, P(dt), K, P(dB), N, P(C), N, P(flag), S())
    ref = table[ids].astype(np.float64) @ B.astype(np.float64)
    assert_close(host(C), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max(), what="row-mapped product")
    assert int(flag.item()) == 0
    ids[5] = V + 3
    hip.call("ebn_gemm_f32_rowmap", P(dev(ids, torch.int32)), V, M, N, K, P(dt), K, P(dB), N, P(C), N, P(flag), S())
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    assert hip.lib().ebn_gemm_f32_rowmap(P(di), V, 255, N, K, P(dt), K, P(dB), N, P(C), N, P(flag), S()) == -2


def test_docvec_step_prologue_in_one_launch(hip):
    """ebn_docvec_stage_gather_f32 = ebn_step_advance + label copy + ebn_gather_rows_f32 over two index segments, bit for bit."""
    rng = np.random.default_rng(3)
    n_rows, din, n0, n1 = 300, 768, 37, 11
    matrix = rng.standard_normal((n_rows, din)).astype(np.float32)
    i0, i1 = rng.integers(0, n_rows, n0).astype(np.int32), rng.integers(0, n_rows, n1).astype(np.int32)
    i0[0], i1[-1] = 0, n_rows - 1
    lab = rng.random(n1).astype(np.float32)
    st = make_state(seed=9, step=4, lr=1e-3)
    X0, lab_d = torch.full((n0 + n1, din), float("nan"), device="cuda"), torch.zeros(n1, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    dm = dev(matrix)
    hip.call("ebn_docvec_stage_gather_f32", P(dev(i0, torch.int32)), n0, P(dev(i1, torch.int32)), n1, P(dev(lab)), P(lab_d), n1, P(dm), n_rows, din,
             P(X0), P(flag), P(st), 0.9, 0.999, S())
    assert np.array_equal(host(X0), matrix[np.concatenate([i0, i1])]) and np.array_equal(host(lab_d), lab) and int(flag.item()) == 0
    got = read_state(st)
    assert got.step == 5 and got.drop_key[8] == on.dropout_key(9, 5, 8)
    i1[3] = n_rows
    hip.call("ebn_docvec_stage_gather_f32", P(dev(i0, torch.int32)), n0, P(dev(i1, torch.int32)), n1, None, None, 0, P(dm), n_rows, din, P(X0), P(flag),
             None, 0.9, 0.999, S())
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and not host(X0)[n0 + 3].any()


# ---------------------------------------------------------------- round 6: the optimizer inside a one-rank step's closing launch
def _flat_buffers(rng, segs):
    """flat theta / grad / m / v of the named segments (name -> element count), 64-element aligned like the engines' FlatParams"""
    off, offs = 0, {}
    for k, n in segs.items():
        offs[k] = off
        off += -(-n // 64) * 64
    th, m, v = rng.standard_normal(off), rng.standard_normal(off) * 0.1, rng.random(off) * 0.01
    return offs, off, th, m, v


def test_finishing_launch_with_adam_inside_against_float64(hip):
    """ebn_grad_finish_adam_f32: a split-K sum, a column-partials sum and a head sum whose outputs live in a flat gradient buffer, plus two `rest`
    ranges whose gradients are already there -- every gradient against float64 sums, and theta / m / v of EVERY parameter against the
    oracle's Keras-form Adam (two steps: the second one starts from non-trivial moments); elements no job and no range owns stay untouched."""
    rng = np.random.default_rng(5)
    M, N, parts, A, nblk, B = 96, 72, 5, 40, 37, 9
    offs, numel, th, m, v = _flat_buffers(rng, {"W": M * N, "q": A, "b": A, "uq": A, "ub": A, "r0": 1000, "gap": 100, "r1": 333})
    th_d, m_d, v_d, g_d = dev(th), dev(m), dev(v), torch.zeros(numel, device="cuda")
    st = make_state(seed=0, step=0, lr=1e-3)
    for t in (1, 2):
        Wp = rng.standard_normal((parts, M, N)).astype(np.float32)
        cp = rng.standard_normal((nblk, 2, A)).astype(np.float32)
        hp_, rows = rng.standard_normal((B, 2, A)).astype(np.float32), rng.standard_normal(B).astype(np.float32)
        g_rest = {k: rng.standard_normal(n).astype(np.float32) for k, n in (("r0", 1000), ("r1", 333))}
        for k, a_ in g_rest.items():
            g_d[offs[k]: offs[k] + a_.size] = dev(a_)
        view = lambda k, n: g_d[offs[k]: offs[k] + n]
        loss = torch.zeros(1, device="cuda")
        jobs = (hip.FinishJob * 3)(
            _finish_job(hip, hip.FINISH_COLRED, nblk, 1, A, dev(cp), view("q", A), view("b", A), ld=A),
            _finish_job(hip, hip.FINISH_HEAD, 1, B, A, dev(hp_), view("uq", A), view("ub", A), ld=A, loss_rows=dev(rows), loss_out=loss),
            _finish_job(hip, hip.FINISH_SPLITK, parts, M, N, dev(Wp), view("W", M * N), ld=N))
        ad = hip.AdamFlat()
        ad.theta, ad.grad, ad.m, ad.v, ad.numel = th_d.data_ptr(), g_d.data_ptr(), m_d.data_ptr(), v_d.data_ptr(), numel
        ad.beta1, ad.beta2, ad.eps, ad.grad_scale, ad.n_rest = 0.9, 0.999, 1e-7, 1.0, 2
        ad.rest_off[0], ad.rest_len[0], ad.rest_off[1], ad.rest_len[1] = offs["r0"], 1000, offs["r1"], 333
        hip.call("ebn_step_advance", P(st), 0.9, 0.999, S())
        hip.call("ebn_grad_finish_adam_f32", jobs, 3, ctypes.byref(ad), P(st), S())
        g = np.zeros(numel)
        g[offs["W"]: offs["W"] + M * N] = Wp.astype(np.float64).sum(0).reshape(-1)
        g[offs["q"]: offs["q"] + A], g[offs["b"]: offs["b"] + A] = cp.astype(np.float64).sum(0)
        g[offs["uq"]: offs["uq"] + A], g[offs["ub"]: offs["ub"] + A] = hp_.astype(np.float64).sum(0)
        for k, a_ in g_rest.items():
            g[offs[k]: offs[k] + a_.size] = a_
        owned = g != 0
        assert_close(host(g_d)[owned], g[owned], rtol=1e-5, atol=1e-5, what=f"gradients, step {t}")
        assert abs(float(loss.item()) - rows.astype(np.float64).sum()) < 1e-5
        th2, m2, v2 = th.copy(), m.copy(), v.copy()
        on.adam_keras_step(th2, g, m2, v2, t, lr=1e-3)
        th[owned], m[owned], v[owned] = th2[owned], m2[owned], v2[owned]  # parameters nobody owns (the gap, the alignment padding) do not move
        assert_close(host(th_d), th, rtol=1e-5, atol=2e-6, what=f"theta, step {t}")
        assert_close(host(m_d), m, rtol=1e-5, atol=1e-6, what=f"m, step {t}")
        assert_close(host(v_d), v, rtol=1e-5, atol=1e-7, what=f"v, step {t}")
    bad = hip.AdamFlat()
    assert hip.lib().ebn_grad_finish_adam_f32(jobs, 3, ctypes.byref(bad), P(st), S()) == -1  # no buffers


@pytest.mark.parametrize("shapes", [[(48, 40, 96), (20, 36, 96)], [(768, 512, 800), (512, 512, 800), (512, 512, 800), (512, 256, 800)]])
def test_docvec_finale_against_float64(hip, shapes):
    """ebn_dvn_finale_f32 at the kernel level: the grouped weight gradients (+ column sums + L2 term) written into a flat gradient buffer,
    Adam on them in the tiles' epilogues, Adam over a `rest` range, the head's d(q) / d(b) sums + Adam, and the batch loss = sum(loss rows)
    + l2 * sum of the L2 column-tile sums the forward launches leave in `stat` -- all against float64 (32 x 32 and 64 x 64 tiles)."""
    from ebrec import _hip

    rng = np.random.default_rng(11 + len(shapes))
    A_, B_ = 24, 7
    segs = {}
    for i, (M, N, K) in enumerate(shapes):
        segs[f"W{i}"], segs[f"b{i}"] = M * N, N
    segs.update({"uq": A_, "ub": A_, "rest": 777})
    offs, numel, th, m, v = _flat_buffers(rng, segs)
    th_d, m_d, v_d, g_d = dev(th), dev(m), dev(v), torch.zeros(numel, device="cuda")
    L = len(shapes) - 1  # hidden layers of the pretend encoder: units = the N of every problem but the last
    a = _hip.DvnArgs()
    a.n_layers, a.din, a.e_out, a.n0, a.n1, a.drop_p, a.l2 = L, shapes[0][0], shapes[-1][1], shapes[0][2] - 16, 16, 0.0, 0.01
    for l in range(L):
        a.units[l] = shapes[l][1]
    stat = torch.zeros(int(_hip.lib().ebn_dvn_stat_floats(ctypes.byref(a))), device="cuda")
    a.stat = stat.data_ptr()
    l2_tiles = [(-(-shapes[l][1] // 64)) for l in range(L)]
    l2_vals = [rng.random(nt).astype(np.float32) for nt in l2_tiles]
    l2_base = 20 * sum(shapes[l][1] for l in range(L))
    for l, vals in enumerate(l2_vals):  # the L2 column-tile sums live behind the accumulators: [EBN_DVN_MAX_LAYERS][1024 / 64]
        stat[l2_base + l * 16: l2_base + l * 16 + len(vals)] = dev(vals)
    probs = (_hip.TnProblem * len(shapes))()
    keep, g = [], np.zeros(numel)
    for i, (M, N, K) in enumerate(shapes):
        A, B, W = (rng.standard_normal(s).astype(np.float32) for s in ((K, M), (K, N), (M, N)))
        dA, dB, dW = dev(A), dev(B), dev(W)
        q = probs[i]
        q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.ldc = M, N, K, dA.data_ptr(), M, dB.data_ptr(), N, N
        q.C, q.colsum = g_d.data_ptr() + 4 * offs[f"W{i}"], g_d.data_ptr() + 4 * offs[f"b{i}"]
        lam = 0.02 if i < L else 0.0
        if lam:
            q.l2_W, q.two_lambda = dW.data_ptr(), lam
        keep += [dA, dB, dW]
        g[offs[f"W{i}"]: offs[f"W{i}"] + M * N] = (A.astype(np.float64).T @ B.astype(np.float64) + lam * W).reshape(-1)
        g[offs[f"b{i}"]: offs[f"b{i}"] + N] = B.astype(np.float64).sum(0)
    hp_, rows = rng.standard_normal((B_, 2, A_)).astype(np.float32), rng.standard_normal(B_).astype(np.float32)
    g_rest = rng.standard_normal(777).astype(np.float32)
    g_d[offs["rest"]: offs["rest"] + 777] = dev(g_rest)
    g[offs["rest"]: offs["rest"] + 777] = g_rest
    g[offs["uq"]: offs["uq"] + A_], g[offs["ub"]: offs["ub"] + A_] = hp_.astype(np.float64).sum(0)
    loss = torch.zeros(1, device="cuda")
    f = _hip.DvnFinale()
    f.theta, f.grad, f.m, f.v, f.numel = th_d.data_ptr(), g_d.data_ptr(), m_d.data_ptr(), v_d.data_ptr(), numel
    f.beta1, f.beta2, f.eps, f.grad_scale, f.n_rest = 0.9, 0.999, 1e-7, 1.0, 1
    f.rest_off[0], f.rest_len[0] = offs["rest"], 777
    hp_d, rows_d = dev(hp_), dev(rows)
    f.head_partials, f.B, f.A, f.dq, f.db = hp_d.data_ptr(), B_, A_, g_d.data_ptr() + 4 * offs["uq"], g_d.data_ptr() + 4 * offs["ub"]
    f.loss_rows, f.loss_out = rows_d.data_ptr(), loss.data_ptr()
    st = make_state(seed=0, step=0, lr=1e-3)
    hip.call("ebn_step_advance", P(st), 0.9, 0.999, S())
    hip.call("ebn_dvn_finale_f32", ctypes.byref(a), probs, len(shapes), ctypes.byref(f), P(st), S())
    owned = g != 0
    assert_close(host(g_d)[owned], g[owned], rtol=2e-5, atol=2e-5 * np.abs(g).max(), what="gradients")
    want_loss = rows.astype(np.float64).sum() + 0.01 * sum(float(x.astype(np.float64).sum()) for x in l2_vals)
    assert abs(float(loss.item()) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    g_eng = host(g_d)  # Adam is checked on the gradients the launch itself formed (fp32), so that the comparison isolates the optimizer
    th2, m2, v2 = th.copy(), m.copy(), v.copy()
    on.adam_keras_step(th2, g_eng, m2, v2, 1, lr=1e-3)
    th[owned], m[owned], v[owned] = th2[owned], m2[owned], v2[owned]
    assert_close(host(th_d), th, rtol=1e-5, atol=2e-6, what="theta")
    assert_close(host(m_d), m, rtol=1e-5, atol=1e-6, what="m")
    assert_close(host(v_d), v, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(v).max())), what="v")
    f.numel = 10  # a gradient pointer outside the flat buffer is refused
    assert hip.lib().ebn_dvn_finale_f32(ctypes.byref(a), probs, len(shapes), ctypes.byref(f), P(st), S()) == -1
