"""Times the LDS-free 16x16-block GEMM (csrc/ebn_gemm_direct.hip) on the AttLayer2 shapes of a config under the tuning
overrides of THIS process's environment (EBN_GEMM_DIRECT=0|1, EBN_GEMM_DIRECT_DEPTH=2|3, EBN_GEMM_DIRECT_R, EBN_GEMM_DIRECT_C).
usage: direct_gemm_probe.py [rows]      (tools/direct_gemm_sweep.sh runs it over the variants)"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
g = torch.Generator(device="cuda").manual_seed(0)
tag = " ".join(f"{k[9:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("EBN_GEMM_DIRECT")) or "default"
out = []
for name, tB, M, N, K in (("U=Y.W", 0, R, 200, 400), ("dY=dpre.W^T", 1, R, 400, 200)):
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(1, device="cuda")

    def run():
        _hip.call("ebn_gemm_f32_ws", 0, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), K, _hip.ptr(B), B.shape[1], ctypes.c_float(0.0), _hip.ptr(C), N,
                  _hip.ptr(ws), 0, _hip.stream_handle())

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            run()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    ref = A[:256].double() @ (B.double().t() if tB else B.double())
    err = float((C[:256].double() - ref).abs().max() / ref.abs().max())
    out.append(f"{name} {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.1f} TF (err {err:.1e})")
print(f"{tag:40s} rows {R}: " + " | ".join(out), flush=True)
