"""Time of one GEMM shape family as a function of K (fixed overhead vs per-slab cost).  usage: gemm_k_scan.py tA tB M N K1,K2,..."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

tA, tB, M, N = (int(x) for x in sys.argv[1:5])
g = torch.Generator(device="cuda").manual_seed(0)
for K in (int(k) for k in sys.argv[5].split(",")):
    A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda")
    bm, bn, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _hip.call("ebn_gemm_plan", M, N, K, ws.numel(), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))

    def fn():
        _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), A.shape[1], _hip.ptr(B), B.shape[1],
                  ctypes.c_float(0.0), _hip.ptr(C), N, _hip.ptr(ws), ws.numel(), _hip.stream_handle())

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    print(f"tA={tA} tB={tB} {M}x{N}x{K} plan {bm.value}x{bn.value} s{sp.value}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  ({t / (K / 16):.3f} us per 16-deep slab)")
