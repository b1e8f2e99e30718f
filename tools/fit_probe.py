"""End-to-end model.fit() throughput on a synthetic train loader at c2 sizes (host loop + loader + device step).
usage: fit_probe.py [n_impressions]"""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec.models.newsrec import NRMSModel  # noqa: E402
from ebrec.models.newsrec.dataloader import NRMSDataLoader  # noqa: E402

n_imp = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
rng = np.random.default_rng(0)
V, D, T, H, n_art = 250002, 1024, 30, 20, 20000
hp = type("hp", (), dict(title_size=T, history_size=H, head_num=20, head_dim=20, attention_hidden_dim=200, optimizer="adam",
                         loss="cross_entropy_loss", dropout=0.2, learning_rate=1e-4, newsencoder_units_per_layer=None,
                         newsencoder_l2_regularization=1e-4))
art = np.arange(1000, 1000 + n_art)
mapping = {int(a): rng.integers(1, V, T).tolist() for a in art}
df = pd.DataFrame({"user_id": rng.integers(0, 1000, n_imp), "article_id_fixed": [rng.choice(art, H).tolist() for _ in range(n_imp)],
                   "article_ids_inview": [rng.choice(art, 5).tolist() for _ in range(n_imp)],
                   "labels": [np.eye(5, dtype=int)[rng.integers(0, 5)].tolist() for _ in range(n_imp)]})
loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros", batch_size=32)
table = (rng.standard_normal((V, D), dtype=np.float32) * 0.02)
m = NRMSModel(hp, word2vec_embedding=table, seed=1, train_embedding=False)
m._engine.enable_graphs("--no-graph" not in sys.argv)
m.model.fit(loader, epochs=1, verbose=0)  # warm-up: buffers, graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
m.model.fit(loader, epochs=1, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"fit(): {n_imp} impressions, {len(loader)} steps in {dt:.3f} s = {n_imp / dt:,.0f} impressions/s ({dt / len(loader) * 1e3:.3f} ms/step)")
