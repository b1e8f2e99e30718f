"""A/B of one engine attribute on the bench step: tools/flag_ab.py <config> <attribute> [value_a value_b] [precision].
Each setting gets its own model (same table, same batches), 10 warm steps, then the median of three 200-step windows of graph
replays; the settings are interleaved A B A B so that clock drift shows up as a spread, not as a difference."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
import bench  # noqa: E402
from ebrec.models.newsrec import NRMSModel  # noqa: E402

cfg, attr = sys.argv[1], sys.argv[2]
va, vb = (eval(sys.argv[3]), eval(sys.argv[4])) if len(sys.argv) > 4 else (True, False)
precision = sys.argv[5] if len(sys.argv) > 5 else "exact"
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
rng = np.random.default_rng(42)
table = (rng.standard_normal((c["V"], c["D"]), dtype=np.float32) * 0.02) if not c["train_embedding"] else None
batches = bench.synthetic_batches(c, 8, 123, dev)


def engine(value):
    m = NRMSModel(bench.make_hparams(c), word2vec_embedding=table, word_emb_dim=c["D"], vocab_size=c["V"], seed=42,
                  train_embedding=c["train_embedding"], device=dev, precision=precision)
    e = m._engine
    assert hasattr(e, attr), attr
    setattr(e, attr, value)
    e.enable_graphs()
    for k in range(10):
        e.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    return m, e


def window(e, steps=200):
    t0 = time.perf_counter()
    for k in range(steps):
        e.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


(ma, ea), (mb, eb) = engine(va), engine(vb)
ta, tb = [], []
for _ in range(4):
    ta.append(window(ea))
    tb.append(window(eb))
print(f"{cfg} {precision} {attr}={va}: " + " ".join(f"{t:.4f}" for t in ta) + f" ms | {attr}={vb}: " + " ".join(f"{t:.4f}" for t in tb) + " ms")
