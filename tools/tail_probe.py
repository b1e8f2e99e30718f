"""Times the non-projection kernels of the news encoder in isolation at one config's sizes: attention forward / backward
(with the pooling term) and the three AttLayer2 GEMMs.  Ten launches captured into a hipGraph, warm replays first, HIP
events over five replays.  EBNERD_HIP_LIB selects the library (variant builds under csrc/variants/ travel to the GPU box).
usage: tail_probe.py [n_titles] [L] [which]      which: any of a (attention) g (AttLayer2 GEMMs), default both"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec import _hip  # noqa: E402

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 800
L = int(sys.argv[2]) if len(sys.argv) > 2 else 30
which = sys.argv[3] if len(sys.argv) > 3 else "ag"
h, d, A = 20, 20, 200
E, R = h * d, n_seq * L
g = torch.Generator(device="cuda").manual_seed(0)
P, S = _hip.ptr, _hip.stream_handle


def timed(fn, launches=10, replays=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(launches):
            fn()
    for _ in range(20):  # clocks up
        gr.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / (launches * replays) * 1e3)
    return sorted(best)[1]


st = _hip.StepState()
st.step, st.seed, st.lr = 3, 7, 1e-4
for s in range(_hip.binding.EBN_N_SITES):
    st.drop_key[s] = 0x9E3779B9 * (s + 1) & 0xFFFFFFFF
st_dev = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
print(f"lib {os.environ.get('EBNERD_HIP_LIB', 'default')}  n_seq {n_seq} L {L} h {h} d {d}")
if "a" in which:
    qkv = torch.randn(R, 3 * E, device="cuda", generator=g)
    Y = torch.empty(R, E, device="cuda")
    dY = torch.randn(R, E, device="cuda", generator=g)
    w = torch.rand(R, device="cuda", generator=g)
    dpool = torch.randn(n_seq, E, device="cuda", generator=g)
    dqkv = torch.empty(R, 3 * E, device="cuda")
    t = timed(lambda: _hip.call("ebn_attn_fwd_f32", P(qkv), 3 * E, P(Y), E, n_seq, L, h, d, P(st_dev), 1, ctypes.c_float(0.2), S()))
    print(f"attn fwd          {t:7.1f} us  {(R * 4 * E * 4) / t / 1e3:7.1f} GB/s")
    t = timed(lambda: _hip.call("ebn_attn_bwd_pooled_f32", P(qkv), 3 * E, P(dY), E, P(w), P(dpool), E, P(dqkv), 3 * E, n_seq, L, h, d,
                                P(st_dev), 1, ctypes.c_float(0.2), S()))
    print(f"attn bwd (pooled) {t:7.1f} us  {(R * 7 * E * 4) / t / 1e3:7.1f} GB/s")
    del qkv, Y, dY, dqkv
if "g" in which:
    for name, tA, tB, M, N, K in [("U=Y.W", 0, 0, R, A, E), ("dW=Y^T.dpre", 1, 0, E, A, R), ("dY=dpre.W^T", 0, 1, R, E, A)]:
        Am = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
        Bm = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(int(_hip.lib().ebn_gemm_workspace_floats(M, N, K)), 1), device="cuda")
        bm, bn, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _hip.call("ebn_gemm_plan", M, N, K, ws.numel(), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))
        t = timed(lambda: _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), P(Am), Am.shape[1], P(Bm), Bm.shape[1],
                                    ctypes.c_float(0.0), P(C), N, P(ws), ws.numel(), S()))
        ref = torch.matmul(Am.t() if tA else Am, Bm.t() if tB else Bm)
        err = float((C - ref).abs().max() / ref.abs().max())
        print(f"{name:14s} {M}x{N}x{K} plan {bm.value}x{bn.value} s{sp.value}  {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  rel err {err:.1e}")
