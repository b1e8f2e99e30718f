"""Times the fused news-encoder launches of the c3 step in isolation (hipGraph replays): forward (4 launches) and backward (4 + the
weight-gradient group) of csrc/ebn_docvec.hip on the step's own buffers."""
import ctypes, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "ebnerd-benchmark_amd")]
import numpy as np, torch
from ebrec import _hip
from ebrec.models.newsrec._engine_docvec import DocVecEngine

def main():
    eng = DocVecEngine(768, [512, 512, 512], 20, 16, 16, 200, 0.2, 1e-4, "cross_entropy_loss", 1e-4, seed=1)
    B, C = 32, 5
    rng = np.random.default_rng(0)
    his = rng.standard_normal((B, 20, 768)).astype(np.float32); pred = rng.standard_normal((B, C, 768)).astype(np.float32)
    y = np.zeros((B, C), np.float32); y[:, 0] = 1
    eng.train_step(his, pred, y)
    mb = eng._bufs["mlp"]; a = mb["dvn_live"]; st = _hip.ptr(eng.state); S = _hip.stream_handle
    def fwd(): _hip.call("ebn_dvn_fwd_train_f32", ctypes.byref(a), st, S())
    def bwd(): _hip.call("ebn_dvn_bwd_f32", ctypes.byref(a), st, S())
    def grp(): _hip.call("ebn_gemm_tn_group_f32", a._probs, len(a._probs), S())
    def timeit(fns, reps=200):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with _hip.capture(g):
            for _ in range(10):
                for f in fns: f()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps // 10): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    print(f"fwd(4 launches) {timeit([fwd]):7.1f} us   bwd(4 launches) {timeit([bwd]):7.1f} us   fwd+bwd {timeit([fwd, bwd]):7.1f}   group {timeit([grp]):6.1f}", flush=True)
main()
