"""What would issuing step t+1's embedding gather under step t cost?  The c2 step is replayed as usual while an EXTRA gather of a
different batch runs on a second stream, forked at the start of every step and joined at its end.  step(with extra gather) -
step(plain) = what the overlapped gather costs the step; the gather alone takes ~32 us, so a real prefetch would save 32 - that."""
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
m = NRMSModel(bench.make_hparams(c), word2vec_embedding=table, seed=42, train_embedding=False, device=dev)
eng = m._engine
eng.enable_graphs()
batches = bench.synthetic_batches(c, 8, 123, dev)
ids = [torch.cat([h.reshape(-1), p.reshape(-1)]).contiguous() for h, p, _ in batches]
n_tok = ids[0].numel()
X_alt = torch.empty(n_tok, c["D"], device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(extra, steps=100):
    for k in range(10):
        eng.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if extra:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _hip.call("ebn_gather_rows_f32", _hip.ptr(ids[(k + 1) % 8]), _hip.ptr(eng.table), _hip.ptr(X_alt), n_tok, c["D"], c["V"],
                          _hip.ptr(eng.state), 0, ctypes.c_float(0.2), None, _hip.stream_handle())
        eng.train_step(*batches[k % 8])
        if extra:
            main.wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(3):
    a, b = run(False), run(True)
    print(f"plain {a:.4f} ms   with an extra concurrent gather {b:.4f} ms   delta {1e3 * (b - a):+.1f} us")
