"""Second look at prefetching step t+1's embedding gather (frozen table): tools/prefetch_probe.py ran the extra gather beside the WHOLE
step (+83 us).  Here it is a parallel branch of the step's own hipGraph, forked right before the user-level stage (the ~90 us chain of
small, latency-bound kernels that leaves most CUs idle) and joined right after it.  step(with the branch) - step(plain) = what the
gather costs there; a real prefetch would take the gather (32 us) out of the head of the step and pay that instead."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
import bench  # noqa: E402
from ebrec import _hip  # noqa: E402
from ebrec.models.newsrec import NRMSModel  # noqa: E402

c = dict(bench.CONFIGS["c2"])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(42)
table = rng.standard_normal((c["V"], c["D"]), dtype=np.float32) * 0.02
batches = bench.synthetic_batches(c, 8, 123, dev)
ids_next = torch.cat([batches[1][0].reshape(-1), batches[1][1].reshape(-1)]).contiguous()
n_tok = ids_next.numel()
X_alt = torch.empty(n_tok, c["D"], device=dev)
side = torch.cuda.Stream()
orig_call = _hip.call
state = {"on": False, "eng": None}


def call(name, *args):
    if name == "ebn_user_stage_train_f32" and state["on"]:
        eng = state["eng"]
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            orig_call("ebn_gather_rows_f32", _hip.ptr(ids_next), _hip.ptr(eng.table), _hip.ptr(X_alt), n_tok, c["D"], c["V"],
                      _hip.ptr(eng.state), 0, ctypes.c_float(0.2), None, _hip.stream_handle())
        r = orig_call(name, *args)
        main.wait_stream(side)
        return r
    return orig_call(name, *args)


_hip.call = call


def engine(on):
    state["on"] = on
    m = NRMSModel(bench.make_hparams(c), word2vec_embedding=table, seed=42, train_embedding=False, device=dev)
    e = m._engine
    state["eng"] = e
    e.enable_graphs()
    for k in range(10):
        e.train_step(*batches[k % 8])  # captures the graph with / without the branch
    torch.cuda.synchronize()
    return m, e


def window(e, steps=200):
    t0 = time.perf_counter()
    for k in range(steps):
        e.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


(ma, ea), (mb, eb) = engine(False), engine(True)
for rep in range(3):
    a, b = window(ea), window(eb)
    print(f"plain {a:.4f} ms   gather as a branch beside the user stage {b:.4f} ms   delta {1e3 * (b - a):+.1f} us  (gather alone ~32 us)")
