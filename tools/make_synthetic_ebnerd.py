"""Writes a small synthetic data tree with the EB-NeRD parquet layout and column names the NRMS drivers read
(articles.parquet, <split>/{train,validation}/{behaviors,history}.parquet, ebnerd_testset/test/...).
There is no network in the build environment, so this stands in for `ebnerd_demo` in tests and smoke runs.

    python tools/make_synthetic_ebnerd.py /tmp/ebnerd_data --split ebnerd_demo --impressions 600
"""
import argparse
import datetime as dt
from pathlib import Path

import numpy as np
import pandas as pd

WORDS = ("nyheder sport politik vejr krimi kendte biler bolig penge rejser mad film musik tv fodbold haandbold "
         "kongehuset sundhed skole job valg regering borgmester politi ulykke brand storm sommer vinter ferie").split()


def make(root, split="ebnerd_demo", n_articles=400, n_users=60, n_impressions=600, seed=0, doc_dim=None):
    rng = np.random.default_rng(seed)
    root = Path(root)
    ids = np.sort(rng.choice(np.arange(3_000_000, 9_900_000), n_articles, replace=False)).astype(np.int32)
    sent = lambda n: " ".join(rng.choice(WORDS, n))
    arts = pd.DataFrame({"article_id": ids, "title": [sent(6) for _ in ids], "subtitle": [sent(10) for _ in ids],
                         "body": [sent(40) for _ in ids], "category": rng.integers(1, 20, n_articles).astype(np.int16)})
    root.mkdir(parents=True, exist_ok=True)
    arts.to_parquet(root / "articles.parquet")
    if doc_dim:
        vec = rng.standard_normal((n_articles, doc_dim)).astype(np.float32)
        pd.DataFrame({"article_id": ids, "document_vector": list(vec)}).to_parquet(root / "document_vector.parquet")
    t0 = dt.datetime(2023, 5, 18)
    taste = rng.integers(1, 20, n_users)  # users click their favourite category more often: something to learn

    def part(path, n_imp, days, with_ba=False):
        path.mkdir(parents=True, exist_ok=True)
        users = rng.integers(0, n_users, n_imp)
        rows = []
        for k, u in enumerate(users):
            n_in = 250 if (with_ba and k % 25 == 0) else int(rng.integers(5, 16))
            inview = rng.choice(ids, size=min(n_in, n_articles), replace=False)
            cat = arts.set_index("article_id").loc[inview, "category"].to_numpy()
            w = np.where(cat == taste[u], 8.0, 1.0)
            clicked = rng.choice(inview, size=1, p=w / w.sum())
            rows.append({"impression_id": np.uint32(k + 1), "user_id": np.uint32(u + 10),
                         "impression_time": t0 + dt.timedelta(days=int(rng.integers(0, days)), seconds=int(rng.integers(0, 86000))),
                         "article_ids_inview": inview.astype(np.int32), "article_ids_clicked": clicked.astype(np.int32),
                         "is_beyond_accuracy": bool(with_ba and k % 25 == 0)})
        pd.DataFrame(rows).to_parquet(path / "behaviors.parquet")
        hist = []
        for u in range(n_users):
            n_h = int(rng.integers(3, 40))
            fav = arts.loc[arts["category"] == taste[u], "article_id"].to_numpy()
            pool = np.concatenate([fav, fav, ids]) if len(fav) else ids
            hist.append({"user_id": np.uint32(u + 10), "article_id_fixed": rng.choice(pool, n_h).astype(np.int32)})
        pd.DataFrame(hist).to_parquet(path / "history.parquet")

    part(root / split / "train", n_impressions, 6)
    part(root / split / "validation", max(n_impressions // 4, 40), 2)
    part(root / "ebnerd_testset" / "test", max(n_impressions // 3, 60), 2, with_ba=True)
    return root


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--split", default="ebnerd_demo")
    ap.add_argument("--articles", type=int, default=400)
    ap.add_argument("--users", type=int, default=60)
    ap.add_argument("--impressions", type=int, default=600)
    ap.add_argument("--doc_dim", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(make(a.root, a.split, a.articles, a.users, a.impressions, a.seed, a.doc_dim))
