"""Times the K-chunked 16x16-block weight-gradient kernel (csrc/ebn_gemm_direct.hip, TN form) + its combining pass on the AttLayer2
dW shape under this process's tuning environment (EBN_GEMM_DIRECT_TN_WGS / _R / _C, EBN_GEMM_DIRECT=0 for the 64x64 split-K tiles).
usage: tn_gemm_probe.py [rows]"""
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
tag = " ".join(f"{k[16:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("EBN_GEMM_DIRECT")) or "default"
M, N, K = 400, 200, R
A, B = torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
C = torch.empty(M, N, device="cuda")
ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda")


def run():
    _hip.call("ebn_gemm_f32_ws", 1, 0, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), M, _hip.ptr(B), N, ctypes.c_float(0.0), _hip.ptr(C), N, _hip.ptr(ws), ws.numel(),
              _hip.stream_handle())


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
ref = A.double().t() @ B.double()
err = float((C.double() - ref).abs().max() / ref.abs().max())
print(f"{tag:44s} dW {M}x{N}x{K}: {us:6.1f} us (GEMM + combine) {2.0 * M * N * K / us / 1e6:5.1f} TF  workspace {ws.numel() // (M * N)} slices  err {err:.1e}", flush=True)
