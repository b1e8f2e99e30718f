"""A/B of the inference news encoder at the c2 table (250002 x 1024, frozen): the Embedding gather fused into the projection's
A-operand fetch (ebn_encoder_fwd_gather_f32: table rows -> LDS -> MFMA) against gather-into-X + projection.  Prints titles/s of
engine.encode_news over n_titles random titles and, from HIP events around the projection stage alone, the fused form's
algorithmic gather rate (n_tok * (4 + D * 4) bytes).  usage: eval_gather_ab.py [n_titles]"""
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "ebnerd-benchmark_amd")]
from ebrec import _hip  # noqa: E402
from ebrec.models.newsrec import NRMSModel  # noqa: E402

n_titles = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
V, D, T = 250002, 1024, 30
hp = type("hp", (), dict(title_size=T, history_size=20, head_num=20, head_dim=20, attention_hidden_dim=200, optimizer="adam",
                         loss="cross_entropy_loss", dropout=0.2, learning_rate=1e-4, newsencoder_units_per_layer=None,
                         newsencoder_l2_regularization=1e-4))
rng = np.random.default_rng(0)
table = rng.standard_normal((V, D), dtype=np.float32) * 0.02
m = NRMSModel(hp, word2vec_embedding=table, seed=1, train_embedding=False)
eng = m._engine
ids = torch.from_numpy(rng.integers(0, V, (n_titles, T)).astype(np.int32)).cuda()
out = {}
for fused in (True, False, True, False):
    eng.fuse_eval_gather = fused
    eng.encode_news(ids[:8192])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.encode_news(ids)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.setdefault("fused" if fused else "two_step", []).append(n_titles / dt)
# the projection stage alone on one 8192-title chunk (245760 token rows): gather + GEMM against the fused GEMM
b = eng._news_bufs(8192, False)
E3, R = 3 * eng.E, 8192 * T
S = _hip.stream_handle
pv = eng.params.view


def fused():
    _hip.call("ebn_gemm_f32_rowmap", _hip.ptr(b.ids), V, R, E3, D, _hip.ptr(eng.table), D, _hip.ptr(pv("n_Wqkv")), E3, _hip.ptr(b.QKV), E3,
              _hip.ptr(eng.oob_flag), S())


def two_step():
    _hip.call("ebn_gather_rows_f32", _hip.ptr(b.ids), _hip.ptr(eng.table), _hip.ptr(b.X), R, D, V, None, -1, ctypes.c_float(0.0),
              _hip.ptr(eng.oob_flag), S())
    _hip.call("ebn_gemm_f32", 0, 0, R, E3, D, ctypes.c_float(1.0), _hip.ptr(b.X), D, _hip.ptr(pv("n_Wqkv")), E3, ctypes.c_float(0.0),
              _hip.ptr(b.QKV), E3, S())


def gpu_us(fn, reps=20):
    chunks = [ids[i * 8192:(i + 1) * 8192].reshape(-1).contiguous() for i in range(n_titles // 8192)]
    for c in chunks[:2]:
        b.ids[: c.numel()].copy_(c)
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for r in range(reps):
        b.ids[:R].copy_(chunks[r % len(chunks)])  # other rows every launch: nothing comes out of the 256 MB memory-side cache
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1e3 / reps


tf, t2 = gpu_us(fused), gpu_us(two_step)
fl = 2.0 * R * D * E3
line = {"what": "inference news encoder, c2 table 250002 x 1024: Embedding gather fused into the Q|K|V projection vs gather + projection",
        "encode_news_titles_per_s": {k: [round(v) for v in vs] for k, vs in out.items()},
        "projection_stage_us_per_8192_titles": {"fused": tf, "two_step": t2},
        "fused": {"tflops": fl / tf / 1e6, "frac_of_fp32_mfma_peak": fl / tf / 1e6 / 157.3,
                  "gather_algorithmic_bytes": R * (4 + D * 4), "gather_GBps_fused_form": R * (4 + D * 4) / tf / 1e3,
                  "hbm_bytes_saved_vs_two_step": 2 * R * D * 4}}
print(json.dumps(line))
