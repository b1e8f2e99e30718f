import ctypes, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip
M = int(sys.argv[1]); N, K = 300, 1200
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g); C = torch.empty(M, N, device="cuda")
ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda")
def run():
    _hip.call("ebn_gemm_f32_ws", 0, 1, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), K, _hip.ptr(B), K, ctypes.c_float(0.0), _hip.ptr(C), N, _hip.ptr(ws), ws.numel(), _hip.stream_handle())
for _ in range(3): run()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(10): run()
for _ in range(10): gr.replay()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [gr.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50 * 1e3)
t = sorted(ts)[1]
ref = A @ B.t()
tag = " ".join(f"{k[16:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("EBN_GEMM_DIRECT_")) or "lds-staged"
print(f"M={M} {tag:28s} {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF  err {float((C-ref).abs().max()/ref.abs().max()):.1e}")
