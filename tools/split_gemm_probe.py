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

# the split passes on their own: X (both orientations in one pass, with the gather; or one orientation from fp32), dQKV^T, W^T
L = _hip.lib()
u8 = lambda n: torch.empty(int(n) + 64, dtype=torch.uint8, device="cuda")
V = 250002
table = torch.randn(V, D, device="cuda", generator=g)
ids = [torch.randint(0, V, (R,), device="cuda", generator=g, dtype=torch.int32) for _ in range(5)]
XN, XT, dQp, Wp = u8(L.ebn_planes_bytes(R, D)), u8(L.ebn_planes_bytes(D, R)), u8(L.ebn_planes_bytes(N, R)), u8(L.ebn_planes_bytes(N, D))
st = _hip.StepState()
st.step, st.seed = 3, 7
for s_ in range(_hip.binding.EBN_N_SITES):
    st.drop_key[s_] = 0x9E3779B9 * (s_ + 1) & 0xFFFFFFFF
st_dev = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
k = [0]


def gather_split():
    k[0] = (k[0] + 1) % 5
    _hip.call("ebn_gather_split_planes_f32", P(ids[k[0]]), P(table), R, D, V, P(st_dev), 0, ctypes.c_float(0.2), None, P(XN), P(XT), S())


def gather_f32():
    k[0] = (k[0] + 1) % 5
    _hip.call("ebn_gather_rows_f32", P(ids[k[0]]), P(table), P(X), R, D, V, P(st_dev), 0, ctypes.c_float(0.2), None, S())


t = timed(gather_f32)
print(f"gather fp32 (+dropout)            {t:8.1f} us  {R * (4 + 2 * D * 4) / t / 1e3:7.1f} GB/s")
t = timed(gather_split)
print(f"gather + split, both orientations {t:8.1f} us  {(R * D * 4 + L.ebn_planes_bytes(R, D) + L.ebn_planes_bytes(D, R)) / t / 1e3:7.1f} GB/s")
t = timed(lambda: _hip.call("ebn_split_planes_f32", P(X), D, R, D, 0, P(XN), S()))
print(f"split X  [R][D] -> planes         {t:8.1f} us  {(R * D * 4 + L.ebn_planes_bytes(R, D)) / t / 1e3:7.1f} GB/s")
t = timed(lambda: _hip.call("ebn_split_planes_f32", P(dQ), N, N, R, 1, P(dQp), S()))
print(f"split dQKV^T [R][3E] -> planes    {t:8.1f} us  {(R * N * 4 + L.ebn_planes_bytes(N, R)) / t / 1e3:7.1f} GB/s")
t = timed(lambda: _hip.call("ebn_split_planes_f32", P(W), N, N, D, 1, P(Wp), S()))
print(f"split W^T [D][3E] -> planes       {t:8.1f} us")
