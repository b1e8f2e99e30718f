"""End-to-end model.fit() throughput on a synthetic train loader at c2 sizes (host loop + loader + device step), or with
--docvec at c3 sizes (NRMSDocVec on 768-d document vectors).   usage: fit_probe.py [n_impressions] [--no-graph] [--docvec] [--auc]"""
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
from ebrec.models.newsrec import NRMSDocVec, NRMSModel  # noqa: E402
from ebrec.models.newsrec.dataloader import NRMSDataLoader  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n_imp = int(pos[0]) if pos else 6400
docvec = "--docvec" in sys.argv
rng = np.random.default_rng(0)
V, D, T, H, n_art = 250002, 1024, 30, 20, 20000
hp = type("hp", (), dict(title_size=T, history_size=H, head_num=20, head_dim=20, attention_hidden_dim=200, optimizer="adam",
                         loss="cross_entropy_loss", dropout=0.2, learning_rate=1e-4, newsencoder_units_per_layer=None,
                         newsencoder_l2_regularization=1e-4))
art = np.arange(1000, 1000 + n_art)
mapping = {int(a): (rng.standard_normal(768).astype(np.float32) if docvec else rng.integers(1, V, T).tolist()) for a in art}
df = pd.DataFrame({"user_id": rng.integers(0, 1000, n_imp), "article_id_fixed": [rng.choice(art, H).tolist() for _ in range(n_imp)],
                   "article_ids_inview": [rng.choice(art, 5).tolist() for _ in range(n_imp)],
                   "labels": [np.eye(5, dtype=int)[rng.integers(0, 5)].tolist() for _ in range(n_imp)]})
loader = NRMSDataLoader(behaviors=df, article_dict=mapping, history_column="article_id_fixed", unknown_representation="zeros", batch_size=32)
if docvec:
    hp = type("hp", (), dict(title_size=768, history_size=H, head_num=16, head_dim=16, attention_hidden_dim=200, optimizer="adam",
                             loss="cross_entropy_loss", dropout=0.2, learning_rate=1e-4, newsencoder_units_per_layer=[512, 512, 512],
                             newsencoder_l2_regularization=1e-4))
    m = NRMSDocVec(hp, seed=1)
else:
    table = (rng.standard_normal((V, D), dtype=np.float32) * 0.02)
    m = NRMSModel(hp, word2vec_embedding=table, seed=1, train_embedding=False)
m._engine.enable_graphs("--no-graph" not in sys.argv)
if "--auc" in sys.argv:  # the reproducibility driver compiles with metrics=["AUC"]: streaming AUC over every batch
    m.model.compile(optimizer=m.model.optimizer, loss=m.model.loss, metrics=["AUC"])
m.model.fit(loader, epochs=1, verbose=0)  # warm-up: buffers, graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
m.model.fit(loader, epochs=1, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"fit(): {n_imp} impressions, {len(loader)} steps in {dt:.3f} s = {n_imp / dt:,.0f} impressions/s ({dt / len(loader) * 1e3:.3f} ms/step)")
