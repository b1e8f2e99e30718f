"""Times the two projection GEMMs of a c2 step in both precisions (exact fp32 / bf16x6 split) and the split passes alone.
usage: split_gemm_probe.py [n_tok] [D] [E3]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
P, S = _hip.ptr, _hip.stream_handle
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, launches=5, replays=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(launches):
            fn()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (launches * replays) * 1e3


X = torch.randn(R, D, device="cuda", generator=g)
W = torch.randn(D, N, device="cuda", generator=g) * 0.03
dQ = torch.randn(R, N, device="cuda", generator=g)
for name, tA, tB, M, Nn, K, A, B in [("fwd  QKV = X.W", 0, 0, R, N, D, X, W), ("bwd  dW = X^T.dQKV", 1, 0, D, N, R, X, dQ)]:
    C0, C1 = torch.empty(M, Nn, device="cuda"), torch.empty(M, Nn, device="cuda")
    for prec, C in ((0, C0), (1, C1)):
        nb = int(_hip.lib().ebn_gemm_prec_workspace_bytes(M, Nn, K, prec))
        ws = torch.empty(nb // 4 + 64, device="cuda")
        t = timed(lambda: _hip.call("ebn_gemm_f32_prec", tA, tB, M, Nn, K, ctypes.c_float(1.0), P(A), A.shape[1], P(B), B.shape[1], ctypes.c_float(0.0),
                                    P(C), Nn, P(ws), nb, prec, S()))
        print(f"{name:20s} {M}x{Nn}x{K} precision {prec}: {t:8.1f} us  {2.0 * M * Nn * K / t / 1e6:7.1f} TF-equivalent  (workspace {nb / 1e6:.0f} MB)")
    ref = (A.t() if tA else A).double() @ B.double()
    for nm, C in (("exact", C0), ("split", C1)):
        print(f"      {nm}: max abs err vs fp64 {float((C.double() - ref).abs().max()):.3e}  (|ref| max {float(ref.abs().max()):.1f})")
