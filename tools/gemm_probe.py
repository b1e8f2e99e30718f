"""Runs one GEMM shape repeatedly (for rocprofv3 --pmc passes). usage: gemm_probe.py tA tB M N K [iters] [nows]
(nows: no split-K workspace -> the planner cannot split)"""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip
tA, tB, M, N, K = (int(x) for x in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
C = torch.empty(M, N, device="cuda")
n = int(_hip.lib().ebn_gemm_workspace_floats(M, N, K))
ws = torch.empty(max(n, 1), device="cuda")
nows = len(sys.argv) > 7 and sys.argv[7] == "nows"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters + 2):
    if i == 2:
        e0.record()
    _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), A.shape[1], _hip.ptr(B), B.shape[1],
              ctypes.c_float(0.0), _hip.ptr(C), N, None if nows else _hip.ptr(ws), 0 if nows else ws.numel(), _hip.stream_handle())
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
print(f"gemm tA={tA} tB={tB} {M}x{N}x{K}{' nows' if nows else ''}: {t*1e6:.1f} us, {2.0*M*N*K/t/1e12:.1f} TFLOP/s")
