"""Feasibility probe: does running the news encoder forward as two title chunks on two streams, staggered so that the
second chunk's Q|K|V GEMM (MFMA-bound) runs under the first chunk's attention / AttLayer2 kernels (VALU / HBM-bound), beat
the one-stream sequence?  c2 sizes, C ABI calls, HIP events.   usage: overlap_probe.py [titles_in_first_chunk]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

N, T, V, D, h, d, A = 800, 30, 250002, 1024, 20, 20, 200
E = h * d
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 341
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.randn(V, D, device="cuda", generator=g) * 0.02
ids = torch.randint(0, V, (N * T,), device="cuda", generator=g, dtype=torch.int32)
Wqkv = torch.randn(D, 3 * E, device="cuda", generator=g) * 0.03
W = torch.randn(E, A, device="cuda", generator=g) * 0.05
b, q = torch.zeros(A, device="cuda"), torch.randn(A, device="cuda", generator=g) * 0.1
X, QKV, Y, U = (torch.empty(N * T, c, device="cuda") for c in (D, 3 * E, E, A))
w, out = torch.empty(N * T, device="cuda"), torch.empty(N, E, device="cuda")
ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(N * T, 3 * E, D)), 1), device="cuda")
P, f32 = _hip.ptr, ctypes.c_float


def chain(t0, t1, part):
    """news encoder forward of titles [t0, t1); part: 'gemm' = gather + projection, 'tail' = attention + AttLayer2"""
    r0, R, S = t0 * T, (t1 - t0) * T, _hip.stream_handle()
    if part == "gemm":
        _hip.call("ebn_gather_rows_f32", P(ids[r0:]), P(table), P(X[r0:]), R, D, V, None, -1, f32(0.0), None, S)
        _hip.call("ebn_gemm_f32_site", 0, 0, R, 3 * E, D, f32(1.0), P(X[r0:]), D, P(Wqkv), 3 * E, f32(0.0), P(QKV[r0:]), 3 * E, None, 0, 1, S)
    else:
        _hip.call("ebn_attn_fwd_f32", P(QKV[r0:]), 3 * E, P(Y[r0:]), E, t1 - t0, T, h, d, None, -1, f32(0.0), S)
        _hip.call("ebn_gemm_f32_ws", 0, 0, R, A, E, f32(1.0), P(Y[r0:]), E, P(W), A, f32(0.0), P(U[r0:]), A, None, 0, S)
        _hip.call("ebn_attpool_fwd_f32", P(U[r0:]), P(b), P(q), P(Y[r0:]), P(out[t0:]), P(w[r0:]), t1 - t0, T, E, A, S)


def one_stream():
    chain(0, N, "gemm")
    chain(0, N, "tail")


s2 = torch.cuda.Stream()


def two_streams():
    s1 = torch.cuda.current_stream()
    chain(0, n0, "gemm")
    e0 = torch.cuda.Event()
    e0.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(e0)
        chain(n0, N, "gemm")   # runs under chunk 0's tail
        e1 = torch.cuda.Event()
        e1.record(s2)
    chain(0, n0, "tail")
    s1.wait_event(e1)
    chain(n0, N, "tail")


def two_chunks_one_stream():
    chain(0, n0, "gemm"); chain(0, n0, "tail"); chain(n0, N, "gemm"); chain(n0, N, "tail")


def timeit(fn, reps=30):
    for _ in range(40):   # steady state (clocks ramp for ~50 ms after an idle gap)
        fn()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b_.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b_) / reps * 1e3


ref = None
for name, fn in (("one stream, whole batch", one_stream), (f"one stream, chunks {n0}+{N - n0}", two_chunks_one_stream),
                 (f"two streams, chunks {n0}+{N - n0}, staggered", two_streams)):
    us = timeit(fn)
    torch.cuda.synchronize()
    chk = float(out.double().sum())
    ref = chk if ref is None else ref
    print(f"{name:45s} {us:8.1f} us   (checksum {'ok' if abs(chk - ref) <= 1e-6 * abs(ref) else 'DIFFERS'})")
