#!/usr/bin/env python
"""Headline benchmark of the MI355X-native NRMS training path.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5|c5h50]

N > 1: either launched by ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` (one rank per
GPU over RCCL), or started directly, in which case it re-launches itself that way.  With fewer visible GPUs than ranks
(a 1-GPU box) the ranks share the GPU over gloo: a functional dry run of the multi-rank path, flagged in the JSON line.

Metric (BASELINE.json): training impressions/sec, plus the embedding-gather HBM GB/s.
A "step" is one optimizer step -- forward, loss, backward, (RCCL gradient all-reduce), Keras-form
Adam -- over one synthetic EB-NeRD-shaped batch per GPU.  Workload = BASELINE.json configs[1]
("c2" in SURVEY.md section 8): NRMS, history_size=20, npratio=4 (C=5), title_len=30, head 20x20,
attention_hidden 200, dropout 0.2, 250002 x 1024 xlm-roberta-large-shaped token table as a frozen
lookup, batch 32 per GPU.  Multi-GPU = data parallel, weak scaling (per-GPU batch fixed), no
data-path collective other than the gradient all-reduce.  Inputs are resident in HBM before the
timed region starts.

Extra objects on the JSON line:
  roofline        the time-dominant kernel (the Q|K|V projection GEMM, MFMA-bound) on the step's own buffers and
                  arguments: 10 launches captured into a hipGraph (the launch path of the timed region), HIP events
                  on the launch stream around 5 replays
  roofline_gather the title-embedding gather (HBM-bound) the metric string names, same method
  roofline_step   the WHOLE step against the exact-fp32 MFMA peak: exact matmul FLOPs of one step / the step's time
  cpu_baseline    oracle/nrms_torch.py (fp32 torch-eager port of the reference math; the reference's
                  TF path cannot run here) timed on the host cores: 20 timed steps after 5 warm-ups

Every figure on the line is measured in this run or labelled static:  the kernel NAMES and the HBM `traffic` of the two
roofline kernels come from three short rocprofv3 passes (--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- each
counter in its own pass, corrected as MI355X_MICROARCH.md section HBM prescribes) over `bench.py --kernel-probe`, a
subprocess that launches exactly those two kernels on the step's shapes.  When rocprofv3 is missing or this process is
itself being profiled, the line says `"traffic_source": "static: profiles/traffic.json ..."` instead.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: exact-fp32 MFMA = vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (16x the exact-fp32 rate)

CONFIGS = {
    # name: (V, D, H, C, T, h, d, A, train_embedding, per-GPU batch)
    "c1": dict(V=32000, D=300, H=20, C=5, T=30, h=20, d=20, A=200, train_embedding=True, B=32),
    "c2": dict(V=250002, D=1024, H=20, C=5, T=30, h=20, d=20, A=200, train_embedding=False, B=32),
    "c4": dict(V=32000, D=300, H=50, C=5, T=30, h=20, d=20, A=200, train_embedding=True, B=32),
    # configs[4]: the c2 table row-sharded over the ranks, global batch 512 on 8 GPUs = 64 per GPU (H=20, and H=50 as on ebnerd_large)
    "c5": dict(V=250002, D=1024, H=20, C=5, T=30, h=20, d=20, A=200, train_embedding=False, B=64, shard_table=True),
    "c5h50": dict(V=250002, D=1024, H=50, C=5, T=30, h=20, d=20, A=200, train_embedding=False, B=64, shard_table=True),
    # NRMSDocVec: 125542 EB-NeRD articles x 768-d document vectors resident in HBM, MLP 512-512-512 -> 256
    "c3": dict(n_articles=125542, doc=768, units=[512, 512, 512], H=20, C=5, h=16, d=16, A=200, B=32),
}


class HP:
    optimizer = "adam"
    loss = "cross_entropy_loss"
    dropout = 0.2
    learning_rate = 1e-4
    newsencoder_units_per_layer = None
    newsencoder_l2_regularization = 1e-4


def make_hparams(c):
    return type("hparams_bench", (HP,), dict(title_size=c["T"], history_size=c["H"], head_num=c["h"], head_dim=c["d"],
                                             attention_hidden_dim=c["A"]))


_ZIPF_CDF = {}


def zipf_ids(rng, shape, V, s=1.0):
    """SURVEY.md 8(d) "Z": token ids drawn Zipf(s) over the V rows, id = frequency rank (tokenizer vocabularies are roughly
    frequency-ordered; id 0 -- the padding / unknown row the reference maps missing articles to, dataloader.py:43,
    _python.py:474-484 -- is the hottest).  Inverse-CDF sampling over the FINITE support: p(k) = (k+1)^-s / H_V."""
    cdf = _ZIPF_CDF.get((V, s))
    if cdf is None:
        cdf = _ZIPF_CDF[(V, s)] = np.cumsum(1.0 / np.arange(1, V + 1, dtype=np.float64) ** s)
    return np.minimum(np.searchsorted(cdf, rng.random(shape) * cdf[-1], side="right"), V - 1).astype(np.int64)


def synthetic_ids(rng, B, H, C, T, V, ids="uniform", pad_frac=0.15):
    """One batch of title-token ids, SURVEY.md 8(d).  "uniform" (U): ids uniform in [0, V) (examples/quick_start/nrms_dummy.py:8-40).
    "zipf" (Z, the realism check): Zipf(1.0) ids + `pad_frac` of the history slots all-zero titles -- the reference left-pads a
    short history with article 0 (_behaviors.py:647-654), whose title row is all token 0: real batches hammer table row 0."""
    if ids == "uniform":
        return rng.integers(0, V, (B, H, T)), rng.integers(0, V, (B, C, T))
    if ids != "zipf":
        raise ValueError(f"ids must be uniform | zipf, got {ids}")
    his, pred = zipf_ids(rng, (B, H, T), V), zipf_ids(rng, (B, C, T), V)
    his[rng.random((B, H)) < pad_frac] = 0
    return his, pred


def synthetic_batches(c, n, seed, device, ids="uniform"):
    """ids uniform in [0,V) (U) or Zipf + padded history (Z), one positive per row at a uniform position."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        if ids == "uniform":  # (torch's generator: the id streams of every earlier round's lines)
            his = torch.randint(0, c["V"], (c["B"], c["H"], c["T"]), generator=g, dtype=torch.int32)
            pred = torch.randint(0, c["V"], (c["B"], c["C"], c["T"]), generator=g, dtype=torch.int32)
        else:
            his, pred = (torch.from_numpy(a.astype(np.int32)) for a in synthetic_ids(rng, c["B"], c["H"], c["C"], c["T"], c["V"], ids))
        y = torch.zeros(c["B"], c["C"])
        y[torch.arange(c["B"]), torch.randint(0, c["C"], (c["B"],), generator=g)] = 1.0
        out.append((his.to(device), pred.to(device), y.to(device)))
    return out


def cpu_baseline(c, steps=20, warmup=5, max_seconds=150.0, ids="uniform", sweep=(4, 8, 16, 32, 64, 128)):
    """fp32 torch-eager port of the reference train step on the host cores: SURVEY.md 8(d)'s protocol -- `warmup` untimed
    steps, then `steps` timed ones (median) -- cut short only if the timed part would pass `max_seconds` (said so in `sample`).
    The intra-op thread count is SWEPT first (1 warm-up + 2 timed steps at each of `sweep` that the box has, plus torch's
    default) and the protocol runs at the best one; the dropout masks are drawn by a thread pool (torch's CPU generator is
    serial: 43 M draws per c2 step)."""
    from oracle.nrms_torch import CpuNRMSTrainer

    rng = np.random.default_rng(123)
    E = c["h"] * c["d"]
    lim = lambda a, b: np.sqrt(6.0 / (a + b))
    P = {"emb": (rng.standard_normal((c["V"], c["D"])) * 0.02).astype(np.float32)}
    for pre, din in (("n", c["D"]), ("u", E)):
        for nm in ("WQ", "WK", "WV"):
            P[f"{pre}_{nm}"] = rng.uniform(-lim(din, E), lim(din, E), (din, E)).astype(np.float32)
        P[f"{pre}_W"] = rng.uniform(-lim(E, c["A"]), lim(E, c["A"]), (E, c["A"])).astype(np.float32)
        P[f"{pre}_b"] = np.zeros(c["A"], np.float32)
        P[f"{pre}_q"] = rng.uniform(-lim(c["A"], 1), lim(c["A"], 1), (c["A"], 1)).astype(np.float32)
    n_cpu = os.cpu_count() or 1
    default_threads = int(torch.get_num_threads())
    tr = CpuNRMSTrainer(P, c["h"], c["d"], loss="cross_entropy_loss", lr=1e-4, dropout=0.2,
                        train_embedding=c["train_embedding"], seed=0, mask_threads=min(16, n_cpu))

    def batch():
        his, pred = synthetic_ids(rng, c["B"], c["H"], c["C"], c["T"], c["V"], ids)
        y = np.zeros((c["B"], c["C"]), np.float32)
        y[np.arange(c["B"]), rng.integers(0, c["C"], c["B"])] = 1
        return his, pred, y

    # thread sweep.  (Round 3 kept torch's default = the physical cores, 128 on the GPU box; round 4's first sweep measured 16 threads
    # 5.7x FASTER than that on this eager workload -- 438 vs 2505 ms per c2 step: the step is dominated by memory-bound elementwise
    # ops and small batched matmuls that do not scale over two sockets -- so the sweep starts low.)
    tr.step(*batch())  # allocator, first touch of the table
    t_sweep0 = time.perf_counter()
    sweep_ms = {}
    for nt in sorted({n for n in sweep if n <= n_cpu} | {default_threads}):
        torch.set_num_threads(nt)
        tr.step(*batch())
        ts = []
        for _ in range(2):
            t1 = time.perf_counter()
            tr.step(*batch())
            ts.append(time.perf_counter() - t1)
        sweep_ms[nt] = min(ts) * 1e3
        if time.perf_counter() - t_sweep0 > 0.4 * max_seconds or (len(sweep_ms) >= 3 and sweep_ms[nt] > 3.0 * min(sweep_ms.values())):
            break  # (far past the optimum: the remaining, larger counts only get slower)
    best = min(sweep_ms, key=sweep_ms.get)
    torch.set_num_threads(best)
    n_untimed = 1 + 3 * len(sweep_ms)
    for _ in range(max(warmup - n_untimed, 2)):  # the sweep's steps were warm-ups too (SURVEY.md 8d asks for >= 5 before the timed ones)
        tr.step(*batch())
        n_untimed += 1
    times = []
    t0 = time.perf_counter()
    while len(times) < steps:
        t1 = time.perf_counter()
        tr.step(*batch())
        times.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 >= max_seconds and len(times) >= 3:
            break
    el = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    n, med = len(times), float(np.median(times))
    cut = "" if n == steps else f" (cut from {steps} steps at the {max_seconds:.0f} s cap)"
    return {"value": c["B"] / med, "unit": "impressions/s", "cores": int(best), "kind": "port",
            "timed_steps": n, "warmup_steps": n_untimed, "thread_sweep_ms_per_step": {str(k): round(v, 1) for k, v in sorted(sweep_ms.items())},
            "sample": f"median of {n} timed train steps{cut} of batch {c['B']} ({c['H']}+{c['C']} titles x {c['T']} tokens, table {c['V']}x{c['D']}"
                      f"{' frozen' if not c['train_embedding'] else ' trainable'}, {ids} ids) after {n_untimed} untimed steps (thread sweep included), {el:.1f}s of CPU work "
                      f"(step min/max {min(times) * 1e3:.0f}/{max(times) * 1e3:.0f} ms), oracle/nrms_torch.py fp32 eager at the best of a sweep over "
                      f"{sorted(sweep_ms)} intra-op threads (= {best}; torch's default here {default_threads}, {n_cpu} logical CPUs), dropout masks "
                      f"drawn by {tr.mask_threads} threads.  An untuned eager port: a stated baseline, not a target"}


def step_flops(c):
    """Exact matmul FLOPs (2 per multiply-add) of ONE training step at per-GPU batch B -- every GEMM and attention
    contraction of forward and backward, nothing else (elementwise work, softmaxes, the optimizer are not counted)."""
    B, H, C, T, D, h, d, A = c["B"], c["H"], c["C"], c["T"], c["D"], c["h"], c["d"], c["A"]
    E, N = h * d, B * (H + C)
    n_tok, n_his = N * T, B * H
    f = 0.0
    # news encoder: Q|K|V projection fwd + weight gradient (+ input gradient when the table trains)
    f += 2.0 * n_tok * D * 3 * E * (3 if c["train_embedding"] else 2)
    f += N * h * (4.0 + 8.0) * T * T * d          # attention core: fwd QK^T, P^T V; bwd dV, dP, dQ, dK
    f += 3 * 2.0 * n_tok * E * A                  # AttLayer2: fwd, dW, d(input)
    # user encoder: the same on B sequences of H news vectors (input gradient always needed)
    f += 3 * 2.0 * n_his * E * 3 * E
    f += B * h * (4.0 + 8.0) * H * H * d
    f += 3 * 2.0 * n_his * E * A
    f += 3 * 2.0 * B * C * E                      # scorer fwd, d(cand), d(user)
    return f


def timed_repeats(step_fn, args, sync, world, device):
    """W untimed warm-up steps, then R repeats of EXACTLY K steps, each bracketed by barrier + synchronize on both sides;
    per repeat the MAX over ranks.  Returns the R wall times (seconds)."""
    k = 0
    for _ in range(args.warmup):
        step_fn(k)
        k += 1
    times = []
    for _ in range(args.repeats):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn(k)
            k += 1
        sync()
        dt = time.perf_counter() - t0
        if world > 1 or dist.is_initialized():  # (a one-rank group under --force-dist: the same collective, an identity)
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    return times


def launches_of_one_step(eng, step_fn, sync):
    """Kernel launches of ONE training step, COUNTED: the library's own launch counter (ebn_launch_count: every launch site goes
    through it) around one eager step of the same engine -- the captured graphs replay exactly these launches."""
    from ebrec import _hip

    sync()
    was, eng.use_graph = eng.use_graph, False
    try:
        n0 = int(_hip.lib().ebn_launch_count())
        step_fn()
        n1 = int(_hip.lib().ebn_launch_count())
    finally:
        eng.use_graph = was
    sync()
    return n1 - n0


def time_kernel(fns, sync, reps=10, replays=5):
    """Average duration of ONE launch of a kernel: `reps` launches (cycling through the launchers `fns` -- e.g. the same
    gather over the id sets of different batches, so that no launch re-reads rows the previous ones left in the memory-side
    cache) captured into a hipGraph (the timed region replays graphs too: same launch path, no host in the loop),
    bracketed by HIP events on the launch stream, replayed `replays` times.  Seconds."""
    fns = fns if isinstance(fns, (list, tuple)) else [fns]
    fns[0]()
    sync()
    g = torch.cuda.CUDAGraph()
    from ebrec import _hip

    with _hip.capture(g):  # (thread_local error mode: the RCCL watchdog thread may poll events meanwhile; GC held off)
        for i in range(reps):
            fns[i % len(fns)]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    sync()
    # The chip needs ~50 ms of sustained load after an idle gap before a kernel reaches its steady duration (the first
    # launches after a host sync run 15-20 % slower: per-launch trace in profiles/r02_gemm_launch_trace.txt), and the timed
    # region of the benchmark runs in that steady state.  Warm replays (0.25 s worth, so that the ramp is also a small part of
    # what a profiler averages over this process) without a host sync in between, then the timed ones.
    warm = int(min(max(0.25 / max(e0.elapsed_time(e1) * 1e-3, 1e-6), 3), 1000))
    for _ in range(warm):
        g.replay()
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    sync()
    return e0.elapsed_time(e1) / (reps * replays) * 1e-3


def hbm_calibration(sync, device, mib=1024):
    """What this box's HBM delivers to a plain streaming kernel, measured in this run ("confirm on the box with a
    DtoD / stream calibration and report both"), on `mib` MiB (4x the 256 MB memory-side cache, so neither side is
    cache-resident), counted as bytes read + bytes written, timed like the kernels (graph of 10 launches, warm replays, HIP
    events on the launch stream).  Two copies: torch's device-to-device `copy_` and a float4 row copy -- this library's gather
    kernel over the identity permutation of 4 KB rows, no dropout: the same 16-byte-per-lane streaming access as
    MI355X_MICROARCH.md's "float4 copy" (6.29 TB/s there).  `achievable_gbs` = the better of the two; the spec figure (8 TB/s)
    stays the `peak` of the roofline objects."""
    from ebrec import _hip

    n = mib * (1 << 20) // 4
    src, dst = torch.empty(n, device=device).normal_(), torch.empty(n, device=device)
    t_copy = time_kernel(lambda: dst.copy_(src), sync)
    rows = n // 1024
    ids = torch.arange(rows, dtype=torch.int32, device=device)
    flag = torch.zeros(1, dtype=torch.int32, device=device)

    def row_copy():
        _hip.call("ebn_gather_rows_f32", _hip.ptr(ids), _hip.ptr(src), _hip.ptr(dst), rows, 1024, rows, None, -1, ctypes.c_float(0.0), _hip.ptr(flag),
                  _hip.stream_handle())

    t_rows = time_kernel(row_copy, sync)
    assert torch.equal(dst[:4096], src[:4096])
    del src, dst
    torch.cuda.empty_cache()
    copy_gbs, rows_gbs = 2.0 * n * 4 / t_copy / 1e9, (2.0 * n * 4 + rows * 4) / t_rows / 1e9
    return {"achievable_gbs": max(copy_gbs, rows_gbs), "torch_copy_gbs": copy_gbs, "float4_row_copy_gbs": rows_gbs, "spec_gbs": HBM_PEAK_GBS,
            "guide_float4_copy_gbs": 6290.0,
            "method": f"device-to-device copies of {mib} MiB, (read + write bytes) / time, graph of 10 launches, HIP events around 5 warm replays: "
                      "torch copy_ and a float4 row copy (gather_rows_vec4_kernel over the identity permutation of 4 KB rows, no dropout); "
                      "achievable_gbs = the better one"}


def fit_loop_leg(model, c, n_steps=100):
    """`model.model.fit(loader)` end to end -- host loop, this repo's NRMSDataLoader, pinned staging, device step -- on a synthetic
    loader of the bench's shape: what the timed region above leaves out (it replays pre-staged batches).  One warm-up epoch
    (buffers, graph capture for the loader's launch form), then one timed epoch of n_steps full batches."""
    import pandas as pd
    from ebrec.models.newsrec.dataloader import NRMSDataLoader

    rng = np.random.default_rng(7)
    n_imp, n_art = n_steps * c["B"], 20000
    art = np.arange(1000, 1000 + n_art)
    toks = rng.integers(1, c["V"], (n_art, c["T"]))
    mapping = {int(a): toks[i].tolist() for i, a in enumerate(art)}
    df = pd.DataFrame({"user_id": rng.integers(0, 1000, n_imp),
                       "article_id_fixed": list(rng.choice(art, (n_imp, c["H"]))), "article_ids_inview": list(rng.choice(art, (n_imp, c["C"]))),
                       "labels": list(np.eye(c["C"], dtype=int)[rng.integers(0, c["C"], n_imp)])})
    loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros",
                            batch_size=c["B"])
    model.model.fit(loader, epochs=1, verbose=0, shuffle=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.model.fit(loader, epochs=1, verbose=0, shuffle=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n_imp / dt, "unit": "impressions/s", "ms_per_step": dt / len(loader) * 1e3, "steps": len(loader),
            "what": "model.fit(NRMSDataLoader) end to end (host loop + loader + pinned staging + device step), one timed epoch after a warm-up epoch; "
                    "the headline `value` replays batches already resident in HBM"}


def split_precision_leg(c, make_model, batches, args, sync, device, exact_ms):
    """The opt-in second precision of the projection GEMMs (profiles/HISTORY.md section 4a: every fp32 operand split exactly into three bf16 values, six
    bf16 MFMA cross products, fp32 accumulate), timed in the SAME run as the exact-fp32 headline so that the driver's line carries
    it: K steps x R repeats of the same batches on a second engine, and the accuracy evidence next to it -- the max abs error of
    the Q|K|V projection of one real step (this run's gathered, dropped-out X and this model's Wqkv) against float64, for the
    split GEMM and for the exact-fp32 GEMM on the same operands."""
    from ebrec import _hip

    model = make_model("split")
    eng = model._engine
    eng.enable_graphs(not args.no_graph)
    times = timed_repeats(lambda k: eng.train_step(*batches[k % len(batches)]), argparse.Namespace(warmup=args.warmup, repeats=args.repeats, steps=args.steps),
                          sync, 1, device)
    ms = float(np.median([t / args.steps * 1e3 for t in times]))
    # accuracy of the projection on a sample of real operands: 2048 token rows of an eager exact-precision gather
    ex = make_model("exact")._engine
    ex.train_step(*batches[0])
    sync()
    nb = ex._bufs[("news", True)]
    R, D, E3 = 2048, c["D"], 3 * c["h"] * c["d"]
    X, W = nb.X[:R].contiguous(), ex.params.view("n_Wqkv").contiguous()
    ref = X.double() @ W.double()
    errs = {}
    for name, prec in (("exact_fp32", 0), ("split_bf16x6", 1)):
        nbytes = int(_hip.lib().ebn_gemm_prec_workspace_bytes(R, E3, D, prec))
        ws, C = torch.empty(nbytes // 4 + 64, device=device), torch.empty(R, E3, device=device)
        _hip.call("ebn_gemm_f32_prec", 0, 0, R, E3, D, ctypes.c_float(1.0), _hip.ptr(X), D, _hip.ptr(W), E3, ctypes.c_float(0.0), _hip.ptr(C), E3,
                  _hip.ptr(ws), nbytes, prec, _hip.stream_handle())
        errs[name] = float((C.double() - ref).abs().max())
    del model, ex
    torch.cuda.empty_cache()
    return {"value": c["B"] / (ms * 1e-3), "unit": "impressions/s", "ms_per_step": ms, "ms_per_step_repeats": [t / args.steps * 1e3 for t in times],
            "speedup_vs_exact": exact_ms / ms, "dtype": "f32 (bf16x6 split, fp32 accumulate)",
            "projection_max_abs_err_vs_fp64": errs, "projection_ref_max_abs": float(ref.abs().max()),
            "what": "OPT-IN second precision, never the headline `value`: the news encoder's projection GEMMs as six bf16 MFMA products of "
                    "exactly-split fp32 operands with fp32 accumulation; same steps, same batches, same run.  `projection_max_abs_err_vs_fp64`: "
                    f"{R} token rows of this run's X . Wqkv against float64, for both GEMM kernels"}


def being_profiled() -> bool:
    return any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def probe_kernels(config, batch, precision="exact", ids="uniform", pick=None):
    """Kernel names and HBM traffic of the two roofline kernels, observed IN THIS RUN: three rocprofv3 passes over
    `bench.py --kernel-probe` (a subprocess that builds the same engine and launches the Q|K|V projection and the gather
    eagerly on the step's buffers): (1) --kernel-trace --stats -> the names as the profiler sees them, (2) --pmc FETCH_SIZE
    and (3) --pmc WRITE_SIZE -> per-launch fabric traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB (each counter in its own pass;
    FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM: gfx950 tallies the 128-B requests of 16-B/lane loads at 64 B).
    Returns None when rocprofv3 is unavailable or this process is itself under a profiler."""
    import csv
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None or being_profiled():
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="ebn_probe_", dir=os.environ.get("TMPDIR", "/tmp"))
    probe = [sys.executable, str(Path(__file__).resolve()), "--kernel-probe", "--config", config, "--precision", precision, "--ids", ids] + \
        (["--batch", str(batch)] if batch else [])
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    try:
        for tag, flags in (("stats", ["--kernel-trace", "--stats"]), ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"]),
                           ("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"])):
            # the statistics pass launches each kernel 300 times back to back (the chip reaches its steady clocks after ~50 ms of load: its
            # AVERAGE then is the steady figure of `avg_launch_us`); the counter passes serialise every dispatch and need only a few
            n_launch = ["--probe-launches", "300" if tag == "stats" else "24"]
            r = subprocess.run([exe] + flags + ["--output-format", "csv", "-d", f"{tmp}/{tag}", "-o", "p", "--"] + probe + n_launch,
                               cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)  # a pass takes ~15 s; a hung profiler (seen: 15 min after an aborted pass) falls back to the static labels
            if r.returncode != 0:
                return None
        pick = pick or (lambda name: "gather" if "gather" in name else ("qkv_gemm" if "gemm" in name else None))
        stats = next(Path(tmp, "stats").rglob("*kernel_stats.csv"))
        for row in csv.DictReader(open(stats)):
            k = pick(row["Name"])
            if k and k not in out:
                out[k] = {"name": row["Name"], "probe_calls": int(row["Calls"]), "probe_avg_us": float(row["AverageNs"]) * 1e-3}
        for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            acc = {}
            for row in csv.DictReader(open(next(Path(tmp, tag).rglob("*counter_collection.csv")))):
                k = pick(row["Kernel_Name"])
                if k and row["Counter_Name"] == counter:
                    acc.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                out.setdefault(k, {})[counter + "_KiB"] = sum(v) / len(v)
        for k, v in out.items():
            if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
                v["traffic"] = (2.0 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0
        keep = os.environ.get("EBN_PROBE_KEEP_DIR")
        if keep:  # the summaries behind `roofline*.kernel` / `.traffic`, kept for profiles/: PROBE-ONLY statistics (a handful of launches
            # of exactly the two roofline kernels on the step's buffers -- no calibration copies, no other role of the same kernel)
            Path(keep).mkdir(parents=True, exist_ok=True)
            tag = config + ("" if precision == "exact" else "_" + precision) + ("" if ids == "uniform" else "_" + ids)
            shutil.copyfile(stats, Path(keep, f"probe_kernel_stats_{tag}.csv"))
            Path(keep, f"probe_pmc_{tag}.json").write_text(json.dumps(
                {"what": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --kernel-probe --config " + config + "`: per-launch "
                         "averages in KiB; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (MI355X_MICROARCH.md, HBM section)", "kernels": out}, indent=1))
        return out
    except Exception:  # a profiler that cannot run here must not take the benchmark down: fall back to the static labels
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def static_traffic(config):
    tf = ROOT / "profiles" / "traffic.json"
    if not tf.exists():
        return {}, None
    blob = json.loads(tf.read_text())
    tags = [k for k in blob.get("_detail", {}) if k.endswith("_" + config)]
    src = f"static: profiles/traffic.json entry '{tags[-1]}'" if tags else "static: profiles/traffic.json"
    return blob.get(config, {}), src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run, file dated " + \
        time.strftime("%Y-%m-%d", time.gmtime(tf.stat().st_mtime)) + "; NOT measured in this run)"


def timing_fields(times, args, world, per_gpu_batch):
    """value / ms_per_step from the MEDIAN repeat; spread alongside."""
    ms = sorted(t / args.steps * 1e3 for t in times)
    med = float(np.median(ms))
    return {"value": world * per_gpu_batch / (med * 1e-3), "ms_per_step": med, "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1],
            "repeats": len(ms), "ms_per_step_repeats": [t / args.steps * 1e3 for t in times],
            "timing": f"median of {len(ms)} repeats of {args.steps} steps each (barrier + synchronize around every repeat, max over ranks); "
                      "`ms_per_step_repeats` lists them in run order -- the first one starts on a chip that has just idled (clock ramp, "
                      "~50 ms) and is the slow one the median sets aside, not an average over it"}


def dist_fields(world, backend, n_dev, forced=False):
    if world == 1 and not forced:
        return {}
    if world == 1:
        return {"ranks": 1, "backend": backend, "dist_note": f"--force-dist: the multi-rank branch on a ONE-rank {backend} group (every collective an identity) -- "
                                                               "an execution check of that code path, not a scaling number"}
    note = "RCCL over xGMI, one rank per GPU" if backend == "nccl" else \
        f"{backend}: {world} ranks share {n_dev} GPU(s) -- functional dry run of the multi-rank path, NOT a scaling number"
    return {"ranks": world, "backend": backend, "dist_note": note}


def bench_docvec(args, c, world, rank, device, sync, dfields, multi=False):
    """configs[2]: NRMSDocVec train step on article-row batches gathered on the device."""
    from ebrec.models.newsrec import NRMSDocVec

    hp = type("hparams_bench_docvec", (HP,), dict(title_size=c["doc"], history_size=c["H"], head_num=c["h"], head_dim=c["d"],
                                                    attention_hidden_dim=c["A"], newsencoder_units_per_layer=c["units"]))
    model = NRMSDocVec(hp, seed=42, device=device)
    eng = model._engine
    if args.no_adam_in_finish:
        eng.fuse_finale = False
    rng = np.random.default_rng(42)
    matrix = rng.standard_normal((c["n_articles"], c["doc"]), dtype=np.float32)
    matrix[0] = 0
    eng.set_article_matrix(matrix)
    if args.force_dist and world == 1:
        eng.force_collectives = True
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    batches = []
    for _ in range(8):
        his = torch.randint(0, c["n_articles"], (c["B"], c["H"]), generator=g, dtype=torch.int32).to(device)
        pred = torch.randint(0, c["n_articles"], (c["B"], c["C"]), generator=g, dtype=torch.int32).to(device)
        y = torch.zeros(c["B"], c["C"])
        y[torch.arange(c["B"]), torch.randint(0, c["C"], (c["B"],), generator=g)] = 1.0
        batches.append((his, pred, y.to(device)))
    if args.kernel_probe:  # the two roofline kernels, eagerly, on the step's own buffers (what probe_kernels() wraps rocprofv3 around)
        eng.train_step(*batches[0], indexed=True)
        rk = eng.roofline_kernels(c["B"], c["C"])
        id_sets = [torch.cat([h_.reshape(-1), p_.reshape(-1)]).contiguous() for h_, p_, _ in batches]
        for i in range(args.probe_launches):
            if "dw_group" in rk:
                rk["dw_group"]()
        for i in range(args.probe_launches):
            rk["gather"](id_sets[i % len(id_sets)])()
        sync()
        return
    eng.enable_graphs(not args.no_graph)
    times = timed_repeats(lambda k: eng.train_step(*batches[k % 8], indexed=True), args, sync, world, device)
    n_launches = launches_of_one_step(eng, lambda: eng.train_step(*batches[0], indexed=True), sync)
    # kernel-level rooflines on the step's own buffers: the document-vector gather (HBM) and the time-dominant launch of the step,
    # the grouped weight-gradient product of all Dense kernels (MFMA)
    rk = eng.roofline_kernels(c["B"], c["C"])
    id_sets = [torch.cat([h.reshape(-1), p.reshape(-1)]).contiguous() for h, p, _ in batches]  # the 8 batches' row numbers
    kt = {"gather": time_kernel([rk["gather"](ids) for ids in id_sets], sync)}
    if "dw_group" in rk:
        kt["dw_group"] = time_kernel(rk["dw_group"], sync)
    if rank == 0:
        n_rows = c["B"] * (c["H"] + c["C"])
        E, A, H, B, C = c["h"] * c["d"], c["A"], c["H"], c["B"], c["C"]
        dims = [c["doc"]] + list(c["units"]) + [E]
        # exact matmul FLOPs of one step: the MLP's Dense products forward, weight gradient and input gradient (none for the
        # document vectors themselves), the user encoder (as in step_flops) and the scorer
        fl_step = sum(2.0 * n_rows * dims[i] * dims[i + 1] * (3 if i else 2) for i in range(len(dims) - 1))
        fl_step += 3 * 2.0 * B * H * E * 3 * E + B * c["h"] * 12.0 * H * H * c["d"] + 3 * 2.0 * B * H * E * A + 3 * 2.0 * B * C * E
        probe = None if (args.no_probe or args.no_roofline or multi) else probe_kernels("c3", 0, pick=lambda n: "gather" if "gather" in n else ("dw_group" if ("tn_finale" in n if rk.get("dw_group_is_finale") else "tn_group" in n) else None))
        line = {"metric": "training impressions/sec", **timing_fields(times, args, world, c["B"]), "unit": "impressions/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup,
                "launch": "eager" if args.no_graph else "hipGraph replay", "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"NRMSDocVec train step, BASELINE.json configs[2] (c3): {c['n_articles']} x {c['doc']} document vectors in HBM, "
                                       f"MLP {c['units']} -> {c['h'] * c['d']}, history_size={c['H']} npratio={c['C'] - 1}, dropout 0.2, adam lr=1e-4",
                           "global_batch": world * c["B"], "per_gpu_batch": c["B"], "parallelism": f"dp{world}",
                           "final_loss": float(eng.loss_dev.item())}, **dfields, "oracle_pin": oracle_pin_status(), "env": nondefault_env()}
        gather_bytes = n_rows * (4 + 2 * c["doc"] * 4)
        src = "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --config c3 --kernel-probe`, (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch" \
            if probe else "not measured (no rocprofv3 passes in this run)"
        if "dw_group" in kt:
            fl = rk["dw_group_flops"]
            fin = bool(rk.get("dw_group_is_finale"))
            line["roofline"] = {"kernel": (probe or {}).get("dw_group", {}).get("name", "gemm_small_tn_finale_kernel<64>" if fin else "gemm_small_tn_group_kernel<64>") +
                                          f" (the weight gradients of the {len(dims) - 1} Dense kernels of the news encoder as ONE grouped launch, "
                                          f"K = {n_rows} rows" + (", with the step's closing work dealt over the same workgroups: Adam on every parameter, the user head's finishing sums, the batch loss"
                                                                  if fin else "") + ": the time-dominant launch of a step that is latency-bound as a whole)", "bound": "mfma",
                                "achieved": fl / kt["dw_group"] / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": fl / kt["dw_group"] / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": (probe or {}).get("dw_group", {}).get("traffic"),
                                "traffic_source": src, "avg_launch_us": kt["dw_group"] * 1e6, "algorithmic_flops_per_launch": fl,
                                "algorithmic_bytes_per_launch": rk["dw_group_bytes"]}
        line["roofline_gather"] = {"kernel": (probe or {}).get("gather", {}).get("name", "gather_rows_vec4_kernel") + " (document-vector gather)", "bound": "hbm",
                                   "achieved": gather_bytes / kt["gather"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gather_bytes / kt["gather"] / 1e9 / HBM_PEAK_GBS, "traffic": (probe or {}).get("gather", {}).get("traffic"),
                                   "traffic_source": src, "avg_launch_us": kt["gather"] * 1e6, "algorithmic_bytes_per_launch": gather_bytes}
        ms = line["ms_per_step"]
        line["roofline_step"] = {"what": "the whole training step against the exact-fp32 MFMA peak: exact matmul FLOPs of one step (every Dense product of the "
                                         "news encoder forward / weight gradient / input gradient, the user encoder, the scorer) / ms_per_step",
                                 "bound": "mfma", "achieved": fl_step / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": fl_step / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "flops_per_step": fl_step,
                                 "launches_per_step": n_launches,
                                 "launches_note": "counted (ebn_launch_count around one eager step of this engine; the hipGraphs replay the same launches): "
                                                  f"news-encoder MLP of {len(c['units'])} hidden layers in the {'fused' if 'dw_group' in rk else 'per-pass'} form, "
                                                  "the user stage, Adam"}
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_docvec(c)
        print("\n" + json.dumps(line), flush=True)  # (own line even when a library -- gloo -- has left an unterminated one on stdout)


def cpu_baseline_docvec(c, steps=20, warmup=5, sweep=(1, 2, 4, 8, 16, 32)):
    """configs[2] on the host cores: oracle/nrms_torch.py:CpuDocVecTrainer (fp32 torch eager, kind "port"), the protocol of
    cpu_baseline(): thread sweep, then `warmup` untimed and `steps` timed steps at the best count (median)."""
    from oracle import nrms_numpy as on
    from oracle.nrms_torch import CpuDocVecTrainer

    rng = np.random.default_rng(123)
    P = on.init_docvec_params(c["doc"], c["units"], c["h"], c["d"], c["A"], seed=1, dtype=np.float32)
    tr = CpuDocVecTrainer(P, c["units"], c["h"], c["d"], lr=1e-4, dropout=0.2, l2=1e-4, seed=0)
    matrix = rng.standard_normal((4096, c["doc"]), dtype=np.float32)

    def batch():
        his = matrix[rng.integers(0, len(matrix), (c["B"], c["H"]))]
        pred = matrix[rng.integers(0, len(matrix), (c["B"], c["C"]))]
        y = np.zeros((c["B"], c["C"]), np.float32)
        y[np.arange(c["B"]), rng.integers(0, c["C"], c["B"])] = 1
        return his, pred, y

    default_threads, n_cpu = int(torch.get_num_threads()), os.cpu_count() or 1
    tr.step(*batch())
    sweep_ms = {}
    for nt in sorted({n for n in sweep if n <= n_cpu} | {default_threads}):
        torch.set_num_threads(nt)
        tr.step(*batch())
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            tr.step(*batch())
            ts.append(time.perf_counter() - t1)
        sweep_ms[nt] = min(ts) * 1e3
    best = min(sweep_ms, key=sweep_ms.get)
    torch.set_num_threads(best)
    for _ in range(warmup):
        tr.step(*batch())
    times = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        tr.step(*batch())
        times.append(time.perf_counter() - t1)
    el = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    med = float(np.median(times))
    return {"value": c["B"] / med, "unit": "impressions/s", "cores": int(best), "kind": "port", "timed_steps": steps,
            "thread_sweep_ms_per_step": {str(k): round(v, 2) for k, v in sorted(sweep_ms.items())},
            "sample": f"median of {steps} timed NRMSDocVec train steps of batch {c['B']} ({c['H']}+{c['C']} document vectors x {c['doc']}, MLP {c['units']}, "
                      f"dropout 0.2, l2 1e-4) after the sweep and {warmup} untimed steps, {el:.2f}s of CPU work (step min/max {min(times) * 1e3:.1f}/{max(times) * 1e3:.1f} ms), "
                      f"oracle/nrms_torch.py:CpuDocVecTrainer fp32 eager at the best of a sweep over {sorted(sweep_ms)} intra-op threads (= {best}; "
                      f"torch's default here {default_threads}, {n_cpu} logical CPUs).  An untuned eager port: a stated baseline, not a target"}


class HangWatchdog:
    """N > 1: a phase of the benchmark that makes no progress for `timeout_s` (EBN_COLLECTIVE_TIMEOUT_S, default 180 s per phase)
    ends the process with exit code 124 and a message naming the phase and -- from the engine's SegmentTrace -- the segment of the
    step the host was launching and the first one the device has not completed, instead of sitting in RCCL until the driver's
    own limit kills the run without a word.  torch.distributed.run then tears the other ranks down: rc != 0."""

    def __init__(self, rank, describe=lambda: ""):
        import threading

        self.timeout_s = float(os.environ.get("EBN_COLLECTIVE_TIMEOUT_S", "180"))
        self.rank, self.describe, self._deadline, self._what = rank, describe, None, ""
        threading.Thread(target=self._loop, daemon=True, name="ebn-bench-watchdog").start()

    def arm(self, what):
        self._what, self._deadline = what, time.monotonic() + self.timeout_s

    def disarm(self):
        self._deadline = None

    def _loop(self):
        while True:
            time.sleep(0.25)
            d = self._deadline
            if d is not None and time.monotonic() > d:
                try:
                    where = self.describe()
                except Exception as e:
                    where = f"(no segment trace: {type(e).__name__}: {e})"
                sys.stderr.write(f"bench.py: rank {self.rank} HUNG: no progress for {self.timeout_s:.0f} s in phase [{self._what}]; {where}.  "
                                 "Exit 124 (a collective that not every rank joined, or a lost peer).\n")
                sys.stderr.flush()
                os._exit(124)


def rccl_view(world, rank, device, backend):
    """What the communication library itself saw (a COLLECTIVE): the group's world size, the library version, every rank's id
    as delivered by an all-gather on the device, and who sits on which GPU -- so that "did RCCL run with N ranks on N distinct
    GPUs" is answerable from the JSON line."""
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "local_device": device.index, "device_name": props.name, "device_uuid": str(getattr(props, "uuid", "")),
          "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid(), "host": socket.gethostname()}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    ids = torch.empty(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int32, device=device))
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            ver = f"unavailable: {type(e).__name__}"
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver,
            "rank_ids_allgathered": [int(v) for v in ids.cpu().tolist()],
            "distinct_devices": len({(r["host"], r["device_uuid"] or r["local_device"]) for r in everyone}), "ranks": everyone,
            "what": "dist.get_world_size() / torch.cuda.nccl.version() / one int32 all-gather of the rank ids on the device / all_gather_object of "
                    "each rank's GPU: the communication library's own view of the job"}


LEG_FIELDS = ("value", "unit", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "ms_per_step_repeats", "n_gpus", "ranks", "backend", "launch", "dtype",
              "scaling", "comm_exposed_us", "allreduce_bytes_per_step", "comm", "exchange", "roofline_step")


def run_legs(args, world, rank, device, watchdog):
    """N > 1: BASELINE.json configs[3] (c4: history 50, trainable table, dense all-reduce of table + bucket) and configs[4] (c5: the
    c2 table row-sharded over the ranks, all-to-all lookup) at their own per-rank shapes, measured by the SAME driver command as
    the c2 headline -- the driver passes no --config.  Each leg is a fresh set of N processes (every rank of this job starts its
    own child `bench.py --leg --config cX` with this rank's RANK / LOCAL_RANK and a new rendezvous port), under its own
    timeout: a hang or a crash in one leg becomes an `error` entry and cannot lose the headline or the other leg."""
    out = {}
    leg_timeout = float(os.environ.get("EBN_BENCH_LEG_TIMEOUT_S", "420"))
    for cfg in [c for c in args.legs.split(",") if c]:
        port = [None]
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port[0] = sk.getsockname()[1]
        watchdog.arm(f"leg {cfg}: agreeing on a rendezvous port")
        dist.broadcast_object_list(port, src=0)
        env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC_ERROR_HANDLING"))}
        env.update(MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port[0]), EBN_BENCH_LEG="1")
        cmd = [sys.executable, str(Path(__file__).resolve()), "--leg", "--config", cfg, "--gpus", str(world), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--repeats", str(args.repeats), "--ids", args.ids, "--graph-collectives", args.graph_collectives,
               "--no-roofline", "--no-probe", "--no-cpu-baseline"] + (["--no-graph"] if args.no_graph else []) + (["--force-dist"] if args.force_dist else [])
        watchdog.arm(f"leg {cfg}: child processes (own timeout {leg_timeout:.0f} s)")
        watchdog._deadline = time.monotonic() + leg_timeout + 60.0  # the child's own timeout fires first
        t0 = time.perf_counter()
        rec = {}
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=leg_timeout, cwd=str(ROOT))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0:
                rec = {"error": f"rank {rank}'s child exited {r.returncode}", "stderr_tail": r.stderr[-1500:]}
            elif rank == 0 and len(lines) != 1:
                rec = {"error": f"expected one JSON line from the leg, got {len(lines)}", "stderr_tail": r.stderr[-1500:]}
            elif rank == 0:
                d = json.loads(lines[0])
                rec = {k: d[k] for k in LEG_FIELDS if k in d}
                rec["config"] = d["config"]
        except subprocess.TimeoutExpired as e:
            rec = {"error": f"rank {rank}'s child did not finish within {leg_timeout:.0f} s (killed)",
                   "stderr_tail": (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))[-1500:]}
        rec["wall_s"] = time.perf_counter() - t0
        # every rank's verdict to rank 0 (a leg is good only if every rank's child was)
        watchdog.arm(f"leg {cfg}: collecting the ranks' verdicts")
        verdicts = [None] * world
        dist.all_gather_object(verdicts, rec.get("error"))
        bad = {r: v for r, v in enumerate(verdicts) if v}
        if bad and "error" not in rec:
            rec = {"error": "; ".join(f"rank {r}: {v}" for r, v in bad.items()), "wall_s": rec["wall_s"]}
        out[cfg] = rec
    watchdog.disarm()
    return out


SINGLE_LEG_FIELDS = LEG_FIELDS + ("repeats", "steps", "warmup", "roofline", "roofline_gather", "cpu_baseline")


def run_legs_single_gpu(args):
    """N = 1: every OTHER BASELINE.json config at its single-GPU shape, measured by the SAME driver command as the c2 headline (the
    driver passes no --config): c1 = configs[0] (the reference script's literal configuration), c3 = configs[2] (NRMSDocVec), c4 =
    configs[3] at its per-rank shape (history 50, batch 32), c5 = configs[4] on one rank (batch 64, the planned exchange with itself).
    Each leg is a child `bench.py --leg --config cX` under its own timeout: its whole timed region, its kernel rooflines (HIP events
    + the three rocprofv3 passes over its own --kernel-probe) and, for c3, its CPU baseline; a crash or a hang in one leg becomes
    an `error` entry and cannot lose the headline or the other legs."""
    out = {}
    leg_timeout = float(os.environ.get("EBN_BENCH_LEG_TIMEOUT_S", "300"))
    profiled = being_profiled()
    for cfg in [c for c in args.legs.split(",") if c]:
        if cfg not in CONFIGS or cfg == "c2":
            out[cfg] = {"error": f"unknown leg config {cfg!r}"}
            continue
        if profiled:  # a profiler around this process would also wrap the children and mix their kernels into its summary
            out[cfg] = {"skipped": "this process runs under a profiler: profile the leg directly with --config " + cfg}
            continue
        cmd = [sys.executable, str(Path(__file__).resolve()), "--leg", "--config", cfg, "--gpus", "1", "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--repeats", str(args.repeats), "--ids", args.ids, "--no-fit-loop", "--no-split-leg"] + \
              (["--no-graph"] if args.no_graph else []) + (["--no-probe"] if args.no_probe else []) + (["--no-roofline"] if args.no_roofline else []) + \
              ([] if (cfg == "c3" and not args.no_cpu_baseline) else ["--no-cpu-baseline"])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=leg_timeout, cwd=str(ROOT),
                               env=dict(os.environ, EBN_BENCH_LEG="1"))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0:
                rec = {"error": f"the leg's process exited {r.returncode}", "stderr_tail": r.stderr[-1500:]}
            elif len(lines) != 1:
                rec = {"error": f"expected one JSON line from the leg, got {len(lines)}", "stderr_tail": r.stderr[-1500:]}
            else:
                d = json.loads(lines[0])
                rec = {k: d[k] for k in SINGLE_LEG_FIELDS if k in d}
                rec["config"] = d["config"]
        except subprocess.TimeoutExpired as e:
            rec = {"error": f"the leg did not finish within {leg_timeout:.0f} s (killed)",
                   "stderr_tail": (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))[-1500:]}
        rec["wall_s"] = time.perf_counter() - t0
        out[cfg] = rec
    note = ("every other BASELINE.json config on this GPU, each measured by its own child process of this command under its own timeout: c1 = "
            "configs[0], c3 = configs[2] (NRMSDocVec, with its cpu_baseline), c4 = configs[3] at its per-rank shape, c5 = configs[4] on one rank; "
            "same steps / warmup / repeats, same timing protocol, own kernel rooflines; `value` of the line stays the c2 headline")
    return out, note


def oracle_pin_status():
    """Whether anything reference-held pins the oracle's MODEL math (tools/dump_tf_golden.py writes the file where TensorFlow
    exists; it cannot run in this image).  Travels with every number."""
    f = ROOT / "tests" / "golden" / "nrms_tf_golden.npz"
    return f"pinned by {f.relative_to(ROOT)}" if f.exists() else \
        "unpinned (no tests/golden/nrms_tf_golden.npz: TensorFlow is not installable here; evaluator metrics ARE pinned against the imported reference)"


def nondefault_env():
    """the EBN_* switches this run was started under (every one changes a default somewhere: the number of record must say so)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("EBN_") and k not in ("EBN_BENCH_LEG",)}


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun parent: become one (one rank per GPU; over-subscribed on gloo when the
    box has fewer GPUs than ranks)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="the timed loop of --steps steps is run this many times; the median is reported")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--ids", default="uniform", choices=["uniform", "zipf"],
                    help="SURVEY.md 8(d) id distribution: uniform (U, the headline) or zipf (Z: Zipf(1.0) ids + 15 %% of the history slots "
                         "all-zero titles -- hot row 0, as real left-padded EB-NeRD batches have)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of hipGraph replay")
    ap.add_argument("--graph-collectives", nargs="?", const="on", default="off", choices=["off", "on", "auto"],
                    help="N > 1 over RCCL: capture the collectives into the step's hipGraph as well (one graph per step, no eager launches "
                         "between replays).  on: unconditionally.  auto: after a start-up self-check (one step both ways from the same state, "
                         "compared bit for bit, verdict MIN-reduced over the ranks: engine.verify_graph_collectives).  The form has only ever "
                         "met a one-rank RCCL group, hence off by default")
    ap.add_argument("--no-fit-loop", action="store_true", help="skip the model.fit(loader) leg (N = 1 only)")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the split-precision leg (the opt-in bf16x6 projections timed next to the "
                                                               "exact-fp32 headline, N = 1 only)")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed steps of the CPU baseline (after 5 warm-ups; SURVEY.md 8d)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel roofline timing (counter-collection passes)")
    ap.add_argument("--no-probe", action="store_true", help="skip the rocprofv3 passes behind roofline.kernel / roofline.traffic (labelled static then)")
    ap.add_argument("--precision", default="exact", choices=["exact", "split"],
                    help="exact: every matmul on the exact-fp32 MFMA kernels (the headline). split: the news encoder's projection GEMMs as "
                         "bf16x6 split products on the bf16 matrix pipe -- fp32-accurate (three bf16 planes per operand, six cross products, "
                         "fp32 accumulate), an opt-in second precision with its own line")
    ap.add_argument("--atomic-table-grad", action="store_true",
                    help="trainable table: one 64-bit atomic per gradient element instead of combining the duplicate ids of every 64 consecutive "
                         "tokens first (same bits; the A/B behind the default (profiles/HISTORY.md), see --ids zipf)")
    ap.add_argument("--legs", default=None, help="on the default config: also measure these configs as sub-records of the line, each in its own (set of) child "
                                                  "process(es) under its own timeout ('' = none).  Default: N = 1 -> c1,c3,c4,c5 (every other BASELINE.json config at "
                                                  "its single-GPU / per-rank shape, with its own roofline objects); N > 1 -> c4,c5 (configs[3] / configs[4])")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the MULTI-RANK branch on a group of one rank: through torch.distributed.run, init_process_group('nccl', device_id=...), "
                         "rccl_view, SegmentTrace, HangWatchdog, the engine's multi-rank launch form (collectives between the graph replays -- identities "
                         "on one rank), run_legs with its fresh rendezvous and the closing collective flag check: what the 8-GPU node will run, executed "
                         "on a 1-GPU box")
    ap.add_argument("--no-adam-in-finish", action="store_true",
                    help="A/B: Adam on the dense parameters as a launch of its own instead of inside the step's finishing launch (the one-rank default)")
    ap.add_argument("--leg", action="store_true", help="internal: this process is one rank of a leg started by run_legs()")
    ap.add_argument("--fault-skip-collectives-on-rank", type=int, default=-1,
                    help="test hook: this rank skips its collectives in the timed region (its peers then wait for it forever): the hang "
                         "watchdog must turn that into exit code 124 within EBN_COLLECTIVE_TIMEOUT_S")
    ap.add_argument("--probe-launches", type=int, default=24, help="internal: launches of each roofline kernel in --kernel-probe")
    ap.add_argument("--kernel-probe", action="store_true", help="internal: launch the two roofline kernels a few times on the step's "
                                                                "buffers and exit (what probe_kernels() wraps rocprofv3 around)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = max(torch.cuda.device_count(), 1)
    local = local % n_dev  # ranks share GPUs only in the over-subscribed dry run
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = None
    multi = world > 1 or args.force_dist  # the multi-rank branch (a one-rank group under --force-dist)
    if args.legs is None:
        args.legs = "c4,c5" if multi else "c1,c3,c4,c5"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI; RCCL refuses two ranks on one GPU, so a box with fewer GPUs than ranks falls back to gloo
        backend = os.environ.get("EBN_DIST_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dfields = dist_fields(world, backend, n_dev, forced=args.force_dist)
    trace = watchdog = None
    if multi:
        from ebrec.models.newsrec._dist import SegmentTrace

        trace = SegmentTrace()
        watchdog = HangWatchdog(rank, trace.where)
        watchdog.arm("first collectives of the job (communicator set-up, the library's view of the ranks)")
        dfields["rccl_view"] = rccl_view(world, rank, device, backend)
        watchdog.disarm()

    def phase(what):  # N > 1: (re)start the hang watchdog's clock for the next stretch of the run
        if watchdog is not None:
            watchdog.arm(what)

    c = dict(CONFIGS[args.config])
    if args.batch:
        c["B"] = args.batch

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if args.config == "c3":
        phase("c3: the whole DocVec benchmark")
        bench_docvec(args, c, world, rank, device, sync, dfields, multi)
        if multi:
            dist.destroy_process_group()
        return
    from ebrec import _hip
    from ebrec.models.newsrec import NRMSModel

    sharded = bool(c.get("shard_table"))
    rng = np.random.default_rng(42)  # identical weights on every rank (data-parallel replicas)
    table = (rng.standard_normal((c["V"], c["D"]), dtype=np.float32) * 0.02) if not c["train_embedding"] else None
    def make_model(precision):
        return NRMSModel(make_hparams(c), word2vec_embedding=table, word_emb_dim=c["D"], vocab_size=c["V"], seed=42,
                         train_embedding=c["train_embedding"], device=device, shard_table=sharded, precision=precision)

    model = make_model(args.precision)
    eng = model._engine
    batches = synthetic_batches(c, 8, 123 + rank, device, args.ids)

    if args.kernel_probe:
        # the two roofline kernels, eagerly, on the step's own buffers: --probe-launches launches each (the gather cycling over the 8 batches' id sets)
        eng.train_step(*batches[0])  # allocates the step's buffers and fills X with gathered rows
        rk = eng.roofline_kernels(c["B"], c["C"])
        id_sets = [torch.cat([h_.reshape(-1), p_.reshape(-1)]).contiguous() for h_, p_, _ in batches]
        for i in range(args.probe_launches):  # each kernel back to back, as bench.py's own HIP-event timing runs them
            rk["qkv_gemm"]()
        for i in range(args.probe_launches):
            rk["gather"](id_sets[i % len(id_sets)])()
        sync()
        return

    eng.atomic_table_grad = bool(args.atomic_table_grad)
    if args.no_adam_in_finish:
        eng.adam_in_finish = False
    eng.enable_graphs(not args.no_graph)
    eng.trace = trace  # N > 1: an event behind every segment of every step, so that a hang can be named (None at N = 1: nothing recorded)
    if args.force_dist and world == 1:
        from ebrec.models.newsrec._dist import LockStepGuard

        eng.force_collectives = True
        eng.guard = LockStepGuard(force=True)
        eng.guard.enter("bench.py --force-dist (the guard's store rendezvous on a one-rank group)")
    if multi:
        dfields["guard"] = eng.guard.status() if eng.guard is not None else "disabled: the engine built none"
    eng.graph_collectives = bool(args.graph_collectives == "on" and backend == "nccl")
    if args.graph_collectives == "auto" and multi and not args.no_graph:
        phase("start-up self-check of the one-graph step (engine.verify_graph_collectives)")
        eng.verify_graph_collectives(*batches[0])
    if args.fault_skip_collectives_on_rank == rank:  # test hook: this rank leaves its peers alone in every collective of the step
        eng.skip_collectives = True
        if eng.exchange is not None:
            eng.exchange.skip = True
    phase(f"{args.config}: warm-up + timed region ({args.warmup} + {args.repeats} x {args.steps} steps)")
    times = timed_repeats(lambda k: eng.train_step(*batches[k % len(batches)]), args, sync, world, device)
    n_launches = launches_of_one_step(eng, lambda: eng.train_step(*batches[0]), sync) if not multi else None
    # Kernel-level rooflines: the Q|K|V projection GEMM and the embedding gather of THIS step (same buffers, same
    # arguments), each captured into a hipGraph of 10 launches and timed with HIP events on the launch stream.
    rk = eng.roofline_kernels(c["B"], c["C"])
    id_sets = [torch.cat([h.reshape(-1), p.reshape(-1)]).contiguous() for h, p, _ in batches]  # the 8 batches' token ids
    kt = calib = None
    if not args.no_roofline:
        kt = {"qkv_gemm": time_kernel(rk["qkv_gemm"], sync), "gather": time_kernel([rk["gather"](ids) for ids in id_sets], sync)}
        calib = hbm_calibration(sync, device)
    loss = float(eng.loss_dev.item())
    comm = None
    if multi:
        phase(f"{args.config}: the same steps re-timed without collectives (comm_exposed_us)")
        # what the collectives cost the step: the same steps timed again with every collective skipped (the results of those
        # steps are wrong and thrown away; all ranks skip together) -- exposed = with - without, after whatever the overlap hid
        eng.skip_collectives = True
        if eng.graph_collectives:
            eng._graphs.clear()  # the collectives are part of the captured step: re-capture without them
        if eng.exchange is not None:
            eng.exchange.skip = True
        t_no = timed_repeats(lambda k: eng.train_step(*batches[k % len(batches)]), argparse.Namespace(warmup=2, repeats=max(args.repeats, 3), steps=args.steps),
                             sync, world, device)
        eng.skip_collectives = False
        if eng.graph_collectives:
            eng._graphs.clear()
        if eng.exchange is not None:
            eng.exchange.skip = False
        ms_no = float(np.median([t / args.steps * 1e3 for t in t_no]))
        comm = {"ms_per_step_without_collectives": ms_no, "overlap": bool(eng.overlap_collectives), "collectives_in_graph": bool(eng.graph_collectives), "graph_collectives_mode": args.graph_collectives}
    phase(f"{args.config}: closing flag check (a collective)")
    eng.check_oob()  # sticky device flags of the whole run (ids out of range, exchange overflow, accumulator range): raise, don't report
    legs = None
    if multi and not args.leg and args.config == "c2" and args.legs:
        legs = run_legs(args, world, rank, device, watchdog)
    if watchdog is not None:
        watchdog.disarm()

    if rank == 0:
        n_tok = c["B"] * (c["H"] + c["C"]) * c["T"]
        E = c["h"] * c["d"]
        gemm_flops = 2.0 * n_tok * c["D"] * 3 * E
        bm, bn, sp = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _hip.call("ebn_gemm_plan", n_tok, 3 * E, c["D"], 0, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp))
        gemm_name = (f"gemm_f32_kernel<{bm.value}, {bn.value}, {4 if bm.value == 256 else 2}, false, false, true, 0>" if bm.value != 32
                     else "gemm_small_vec_kernel<false, false, 32>") + f" (ebn_gemm_plan: tile {bm.value}x{bn.value}, split-K {sp.value})"
        gather_bytes = n_tok * (4 + 2 * c["D"] * 4)  # id + row read + row write (materialising gather)
        probed = None if (args.no_probe or args.no_roofline or multi) else probe_kernels(args.config, args.batch, args.precision, args.ids)
        if probed and "traffic" in probed.get("qkv_gemm", {}) and "traffic" in probed.get("gather", {}):
            traffic = {k: v["traffic"] for k, v in probed.items()}
            traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                              "`bench.py --kernel-probe`, (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch")
            name_source = "observed in this run: rocprofv3 --kernel-trace --stats over `bench.py --kernel-probe`"
            gemm_name, gather_name = probed["qkv_gemm"]["name"], probed["gather"]["name"]
        else:
            traffic, traffic_source = static_traffic(args.config)
            name_source = "static: the template instantiation ebn_gemm_plan selects for this shape (not observed in a trace of this run)"
            gather_name = "gather_rows_vec4_kernel"
        cfg_idx = {"c1": 0, "c2": 1, "c4": 3, "c5": 4, "c5h50": 4}[args.config]
        ids0 = id_sets[0].cpu().numpy()
        id_note = {"ids": args.ids, "tokens_per_step": int(ids0.size), "distinct_rows_in_a_step": int(np.unique(ids0).size),
                   "tokens_on_row_0": int((ids0 == 0).sum()),
                   "what": "SURVEY.md 8(d) U: uniform ids in [0, V)" if args.ids == "uniform" else
                           "SURVEY.md 8(d) Z: Zipf(1.0) ids (id = frequency rank) + 15 % of the history slots all-zero titles (left-padded "
                           "histories, _behaviors.py:647-654) -- the realism check; the headline is U"}
        table_kind = "trainable" if c["train_embedding"] else "frozen lookup"
        if sharded:
            table_kind += f", row-sharded over {world} rank(s) (routed all-to-all of the distinct rows)"
        line = {
            "metric": "training impressions/sec", **timing_fields(times, args, world, c["B"]), "unit": "impressions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "launch": "eager" if (args.no_graph or not eng.graph_capable) else "hipGraph replay",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "exact" else "f32 (bf16x6 split, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"NRMS train step, BASELINE.json configs[{cfg_idx}] "
                                   f"({args.config}): table {c['V']}x{c['D']} {table_kind}, "
                                   f"history_size={c['H']} npratio={c['C'] - 1} title_len={c['T']} head={c['h']}x{c['d']} "
                                   f"att_hidden={c['A']} dropout=0.2 adam lr=1e-4 CE loss",
                       "global_batch": world * c["B"], "per_gpu_batch": c["B"], "parallelism": f"dp{world}",
                       "final_loss": loss, "id_distribution": id_note,
                       **({"table_gradient": "64-bit fixed-point atomics, one per element" if args.atomic_table_grad else "64-bit fixed-point atomics after combining the duplicate ids of every 64 tokens"}
                          if c["train_embedding"] else {})},
        }
        if args.precision == "split":
            line["precision_note"] = ("OPT-IN second precision, not the headline: the news encoder's projection GEMMs (forward Q|K|V and its "
                                      "weight gradient) split every fp32 operand exactly into three bf16 values and keep the six leading cross "
                                      "products (v_mfma_f32_32x32x16_bf16, fp32 accumulate); dropped terms < 2^-23 per product; every other kernel "
                                      "is the exact-fp32 one.  Same tolerances as the exact path in tests/test_hip_kernels.py")
        fl_step = step_flops(c)
        line["roofline_step"] = {"what": "the whole training step against the exact-fp32 MFMA peak: exact matmul FLOPs of one step "
                                         "(every GEMM and attention contraction, forward and backward) / ms_per_step",
                                 "bound": "mfma", "achieved": fl_step / (line["ms_per_step"] * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": fl_step / (line["ms_per_step"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                 "algorithmic_flops_per_step": fl_step,
                                 **({"launches_per_step": n_launches, "launches_note": "counted: ebn_launch_count around one eager step of this engine"} if n_launches is not None else {})}
        if args.precision == "split":  # part of the step runs on the bf16 pipe: a fraction of the fp32 peak would mean nothing
            line["roofline_step"].update({"what": "fp32-equivalent matmul rate of the whole step (exact matmul FLOPs of one step / ms_per_step); the "
                                                  "projection GEMMs execute 6 bf16 MFMA products per fp32 product on the bf16 pipe, so no single peak applies",
                                          "peak": None, "frac": None})
        if kt is not None and args.precision == "split":
            # the projection as the step runs it: two split passes + the bf16x6 GEMM (six bf16 MFMA products per fp32 product)
            line["roofline"] = {"kernel": gemm_name, "role": "news-encoder Q|K|V projection, forward, as split passes + bf16x6 GEMM "
                                "(timed together: three launches)", "kernel_name_source": name_source, "bound": "mfma",
                                "achieved": 6.0 * gemm_flops / kt["qkv_gemm"] / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": 6.0 * gemm_flops / kt["qkv_gemm"] / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                "fp32_equivalent_tflops": gemm_flops / kt["qkv_gemm"] / 1e12, "traffic": traffic.get("qkv_gemm"),
                                "traffic_source": traffic_source, "avg_launch_us": kt["qkv_gemm"] * 1e6,
                                "algorithmic_flops_per_launch": 6.0 * gemm_flops}
            line["roofline_gather"] = {"kernel": gather_name, "role": "title-token embedding gather + dropout", "kernel_name_source": name_source,
                                       "bound": "hbm", "achieved": gather_bytes / kt["gather"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": gather_bytes / kt["gather"] / 1e9 / HBM_PEAK_GBS, "traffic": traffic.get("gather"),
                                       "traffic_source": traffic_source, "avg_launch_us": kt["gather"] * 1e6,
                                       "algorithmic_bytes_per_launch": gather_bytes}
        elif kt is not None:
            line.update({
            "roofline": {"kernel": gemm_name, "role": "news-encoder Q|K|V projection, forward (the time-dominant kernel)",
                         "kernel_name_source": name_source, "bound": "mfma",
                         "achieved": gemm_flops / kt["qkv_gemm"] / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": gemm_flops / kt["qkv_gemm"] / 1e12 / MFMA_F32_PEAK_TFLOPS,
                         "traffic": traffic.get("qkv_gemm"), "traffic_source": traffic_source, "avg_launch_us": kt["qkv_gemm"] * 1e6,
                         "algorithmic_flops_per_launch": gemm_flops},
            "roofline_gather": {"kernel": gather_name, "role": "title-token embedding gather + dropout", "kernel_name_source": name_source,
                                "bound": "hbm",
                                "achieved": gather_bytes / kt["gather"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": gather_bytes / kt["gather"] / 1e9 / HBM_PEAK_GBS,
                                "traffic": traffic.get("gather"), "traffic_source": traffic_source, "avg_launch_us": kt["gather"] * 1e6,
                                "algorithmic_bytes_per_launch": gather_bytes}})
        if kt is not None and "roofline_gather" in line:
            rg = line["roofline_gather"]
            rg["achievable_gbs"] = calib["achievable_gbs"]
            rg["frac_of_achievable"] = rg["achieved"] / calib["achievable_gbs"]
            rg["calibration"] = calib
            table_mb = c["V"] * c["D"] * 4 / 1e6
            rg["table_residency"] = (f"table {table_mb:.0f} MB > the 256 MB memory-side cache and 8 id sets cycle: rows come from HBM" if table_mb > 256 else
                                     f"CACHE-RESIDENT: the {table_mb:.0f} MB table fits the 256 MB memory-side cache, so the row reads of this gather are "
                                     "served by the cache -- `achieved` is a cache + write-stream rate, not an HBM read rate")
            if args.ids == "zipf":
                rg["table_residency"] += ("; Zipf ids: the hot rows are re-read from L2 / the memory-side cache, `achieved` counts ALGORITHMIC bytes "
                                          "(every token reads its row), so it may exceed what HBM alone delivers")
        line.update(dfields)
        line["oracle_pin"] = oracle_pin_status()
        line["env"] = nondefault_env()
        if legs is not None:
            line["legs"] = legs
            line["legs_note"] = ("BASELINE.json configs[3] (c4) and configs[4] (c5) at their own per-rank shapes, each measured by its own set of "
                                 f"{world} child processes of this job (same ranks, same GPUs, new rendezvous) under its own timeout; `value` of the "
                                 "line stays the c2 headline")
        if sharded:
            line["exchange"] = eng.exchange.stats()
        if multi:
            line["allreduce_bytes_per_step"] = eng.allreduce_bytes(c["B"] * (c["H"] + c["C"]) * c["T"])
            line["comm_exposed_us"] = (line["ms_per_step"] - comm["ms_per_step_without_collectives"]) * 1e3
            line["comm"] = {**comm, "note": "comm_exposed_us = ms_per_step - the same steps with every collective skipped (measured after the timed region, same graphs).  "
                                            "The dense gradients travel as one flat bucket: with a trainable table it is started asynchronously after the dWqkv GEMM and "
                                            "runs under the dX GEMM and the table-gradient accumulation; with a frozen table nothing follows dWqkv"}
        if not multi and not args.no_split_leg and args.precision == "exact" and not sharded and not args.no_graph:
            line["split_precision"] = split_precision_leg(c, make_model, batches, args, sync, device, line["ms_per_step"])
        if not multi and not args.no_fit_loop and not sharded:
            line["fit_loop"] = fit_loop_leg(model, c)
            line["fit_loop"]["frac_of_value"] = line["fit_loop"]["value"] / line["value"]
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(c, steps=args.cpu_steps, ids=args.ids)
        if not multi and not args.leg and args.config == "c2" and args.legs and not args.batch:
            line["legs"], line["legs_note"] = run_legs_single_gpu(args)
        print("\n" + json.dumps(line), flush=True)  # (own line even when a library -- gloo -- has left an unterminated one on stdout)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
