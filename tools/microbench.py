"""Kernel-level timings on one MI355X (HIP events on the launch stream). Not the bench
contract (that is bench.py) -- a development probe: gather GB/s and GEMM TFLOP/s at the hot
shapes of configs c1/c2."""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

P, S = _hip.ptr, _hip.stream_handle


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, V, D, ntok in (("c1", 32000, 300, 24000), ("c2", 250002, 1024, 24000), ("c2_x8", 250002, 1024, 192000)):
        table = torch.randn(V, D, device="cuda", generator=g)
        ids = torch.randint(0, V, (ntok,), device="cuda", dtype=torch.int32, generator=g)
        o = torch.empty(ntok, D, device="cuda")
        t = timeit(lambda: _hip.call("ebn_gather_rows_f32", P(ids), P(table), P(o), ntok, D, V, None, -1,
                                     ctypes.c_float(0.0), None, S()))
        by = ntok * (4 + 2 * D * 4)
        out[f"gather_{name}"] = {"s": t, "GBps_materialised": by / t / 1e9}
        del table, o
    for name, tA, tB, M, N, K in (("qkv_fwd_c2", 0, 0, 24000, 1200, 1024), ("qkv_fwd_c1", 0, 0, 24000, 1200, 300),
                                  ("dWqkv_c2", 1, 0, 1024, 1200, 24000), ("dX_c1", 0, 1, 24000, 300, 1200),
                                  ("att_fwd", 0, 0, 24000, 200, 400), ("user_qkv", 0, 0, 640, 1200, 400),
                                  ("sq4096", 0, 0, 4096, 4096, 4096)):
        A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
        B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
        C = torch.empty(M, N, device="cuda")
        n = int(_hip.lib().ebn_gemm_workspace_floats(M, N, K))
        ws = torch.empty(max(n, 1), device="cuda")
        t = timeit(lambda: _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), P(A), A.shape[1], P(B),
                                     B.shape[1], ctypes.c_float(0.0), P(C), N, P(ws), ws.numel(), S()))
        out[f"gemm_{name}"] = {"s": t, "TFLOPs": 2.0 * M * N * K / t / 1e12, "ws_floats": n}
        ref = (A.t() if tA else A) @ (B.t() if tB else B)
        out[f"gemm_{name}"]["max_abs_err_vs_torch"] = float((C - ref).abs().max())
        t2 = timeit(lambda: torch.matmul(A.t() if tA else A, B.t() if tB else B))
        out[f"gemm_{name}"]["torch_TFLOPs"] = 2.0 * M * N * K / t2 / 1e12
    # attention core at c1/c2 news-encoder size
    n_seq, L, h, d = 800, 30, 20, 20
    E = h * d
    qkv = torch.randn(n_seq * L, 3 * E, device="cuda", generator=g)
    y = torch.empty(n_seq * L, E, device="cuda")
    dq = torch.empty_like(qkv)
    out["attn_fwd_800x30"] = {"s": timeit(lambda: _hip.call("ebn_attn_fwd_f32", P(qkv), 3 * E, P(y), E, n_seq, L, h, d,
                                                            None, -1, ctypes.c_float(0.0), S()))}
    out["attn_bwd_800x30"] = {"s": timeit(lambda: _hip.call("ebn_attn_bwd_f32", P(qkv), 3 * E, P(y), E, P(dq), 3 * E,
                                                            n_seq, L, h, d, None, -1, ctypes.c_float(0.0), S()))}
    print(json.dumps(out, indent=1))
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "microbench.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
