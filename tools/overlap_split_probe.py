"""Cost of cutting the multi-rank step into more graph segments: a single-rank RCCL group (every collective an identity), the engine
told it has two ranks, c2 shape; overlapped form (graph | async AR(A) | graph | async AR(B) | wait | graph) vs serial form
(graph | AR(A) | AR(B) | graph) vs the one-rank step (one graph)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import bench  # noqa: E402
from ebrec.models.newsrec import NRMSModel  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
c = dict(bench.CONFIGS[cfg])
rng = np.random.default_rng(42)
table = (rng.standard_normal((c["V"], c["D"]), dtype=np.float32) * 0.02) if not c["train_embedding"] else None
batches = bench.synthetic_batches(c, 8, 123, dev)


def run(world, overlap, steps=100, graph_collectives=False):
    m = NRMSModel(bench.make_hparams(c), word2vec_embedding=table, word_emb_dim=c["D"], vocab_size=c["V"], seed=42,
                  train_embedding=c["train_embedding"], device=dev, table_grad_exchange="dense")
    e = m._engine
    e.world, e.overlap_collectives, e.graph_collectives = world, overlap, graph_collectives
    e.enable_graphs()
    for k in range(10):
        e.train_step(*batches[k % 8])
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(steps):
            e.train_step(*batches[k % 8])
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(best)[1]


print(f"{cfg}: one rank {run(1, True):.4f} ms | two 'ranks', serial buckets {run(2, False):.4f} ms | overlapped buckets {run(2, True):.4f} ms")
try:
    print(f"{cfg}: collectives captured into the graph: serial {run(2, False, graph_collectives=True):.4f} ms | overlapped {run(2, True, graph_collectives=True):.4f} ms")
except Exception as ex:  # noqa: BLE001
    print("capturing the collectives failed:", repr(ex)[:300])
dist.destroy_process_group()
