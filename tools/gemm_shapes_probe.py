"""Times every GEMM shape of a training step in isolation (HIP events, graph-captured batch of launches to exclude launch
gaps) under the planner's choice, and torch.matmul (hipBLASLt) on the same data for scale.
usage: gemm_shapes_probe.py [c1|c2]"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
D = {"c1": 300, "c2": 1024, "c3": 768, "c4": 300}[cfg]
R, E, A, Ru = (52800, 400, 200, 1600) if cfg == "c4" else (24000, 400, 200, 640)
SHAPES = [  # (name, tA, tB, M, N, K)
    ("n QKV fwd", 0, 0, R, 3 * E, D), ("n dWqkv", 1, 0, D, 3 * E, R), ("n U=Y.W", 0, 0, R, A, E), ("n dW", 1, 0, E, A, R),
    ("n dY=dpre.W^T", 0, 1, R, E, A), ("u QKV fwd", 0, 0, Ru, 3 * E, E), ("u dWqkv", 1, 0, E, 3 * E, Ru), ("u U", 0, 0, Ru, A, E),
    ("u dW", 1, 0, E, A, Ru), ("u dY", 0, 1, Ru, E, A), ("u dX", 0, 1, Ru, E, 3 * E)]
if cfg in ("c1", "c4"):
    SHAPES.insert(2, ("n dX", 0, 1, R, D, 3 * E))
if cfg == "c3":  # NRMSDocVec: 800 document vectors per step through Dense 768-512-512-512-256, user encoder 16 heads x 16
    Rn, E3 = 800, 256
    SHAPES = [("d0 fwd", 0, 0, Rn, 512, 768), ("d1 fwd", 0, 0, Rn, 512, 512), ("out fwd", 0, 0, Rn, 256, 512),
              ("d0 dW", 1, 0, 768, 512, Rn), ("d1 dW", 1, 0, 512, 512, Rn), ("out dW", 1, 0, 512, 256, Rn),
              ("d1 dX", 0, 1, Rn, 512, 512), ("out dX", 0, 1, Rn, 512, 256),
              ("u QKV fwd", 0, 0, 640, 3 * E3, E3), ("u dWqkv", 1, 0, E3, 3 * E3, 640), ("u U", 0, 0, 640, A, E3),
              ("u dW", 1, 0, E3, A, 640), ("u dY", 0, 1, 640, E3, A), ("u dX", 0, 1, 640, E3, 3 * E3)]
g = torch.Generator(device="cuda").manual_seed(0)
print(f"config {cfg}")
tot = 0.0
for name, tA, tB, M, N, K in SHAPES:
    Am = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    Bm = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda")
    bm, bn, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _hip.call("ebn_gemm_plan", M, N, K, ws.numel(), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))

    def ours():
        _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(Am), Am.shape[1], _hip.ptr(Bm), Bm.shape[1],
                  ctypes.c_float(0.0), _hip.ptr(C), N, _hip.ptr(ws), ws.numel(), _hip.stream_handle())

    def blas():
        torch.matmul(Am.t() if tA else Am, Bm.t() if tB else Bm, out=C)

    res = []
    for fn in (ours, blas):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    fl = 2.0 * M * N * K
    tot += res[0]
    print(f"{name:16s} tA={tA} tB={tB} {M:6d}x{N:5d}x{K:6d}  plan {bm.value}x{bn.value} s{sp.value:<2d}  ours {res[0]:7.1f} us {fl / res[0] / 1e6:6.1f} TF | "
          f"hipBLASLt {res[1]:7.1f} us {fl / res[1] / 1e6:6.1f} TF")
print(f"sum ours {tot:.1f} us")
