"""The multi-rank training step of the NRMS device engine: a step as an ordered list of SEGMENTS -- runs of
kernels (captured into hipGraphs) with the collectives between them launched eagerly -- for data parallel (one flat dense-gradient
bucket, dense or sparse table-gradient exchange) and for the row-sharded table (device-planned lookup, equal-split all-to-all);
the one-graph form behind its bit-for-bit self-check.  A mixin of NRMSEngine (_engine.py)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ebrec import _hip

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-7  # tf.keras.optimizers.Adam defaults (nrms.py:77)


def _named(kind, fn, what):
    """a step segment (kind, fn) whose fn carries a description (`SegmentTrace` / the hang watchdog name it)"""
    fn.what = what
    return kind, fn


class SegmentsMixin:
    # ---- device-planned row-sharded lookup, as ("k" kernels | "c" collective, fn) segments -------------------------
    def _lookup_segments(self, b, N):
        ex, n_tok = self.exchange, N * self.T

        def plan(ids, n, cap, ws, slot_rows, inv, counts):
            _hip.call("ebn_shard_plan_i32", _hip.ptr(ids), n, self.V, ex.world, 1 if ex.cyclic else 0, cap, _hip.ptr(ws),
                      _hip.ptr(slot_rows), _hip.ptr(inv), _hip.ptr(counts), _hip.stream_handle())

        def serve(local_rows, out):  # rows other ranks (and this one) asked of my shard; -1 padding gathers a zero row, no flag
            _hip.call("ebn_gather_rows_f32", _hip.ptr(local_rows), _hip.ptr(self.table), _hip.ptr(out), local_rows.numel(), self.D,
                      self.table.shape[0], None, -1, ctypes.c_float(0.0), None, _hip.stream_handle())

        return ex.lookup_segments(b.ids, n_tok, b.xb, plan, serve)

    def _table_grad_segments(self, b, N):
        """d(rows): one gradient row per requested slot locally, each slab sent to its owner, owners accumulate."""
        ex, n_tok = self.exchange, N * self.T
        site, p = (0, self.p) if self.p > 0 else (-1, 0.0)

        def reduce_local(inv, d_slot):
            d_slot.zero_()
            _hip.call("ebn_embedding_grad_scatter_f32", _hip.ptr(inv), _hip.ptr(b.dX), _hip.ptr(d_slot), n_tok, self.D,
                      d_slot.shape[0], _hip.ptr(self.state), site, ctypes.c_float(p), _hip.stream_handle())

        def accumulate(local_rows, grads):
            self.table_grad.zero_()
            _hip.call("ebn_embedding_grad_scatter_f32", _hip.ptr(local_rows), _hip.ptr(grads), _hip.ptr(self.table_grad),
                      local_rows.numel(), self.D, self.table.shape[0], None, -1, ctypes.c_float(0.0), _hip.stream_handle())

        return ex.grad_segments(n_tok, b.xb, reduce_local, accumulate)

    def _local_gather(self, local_rows: torch.Tensor) -> torch.Tensor:
        m = local_rows.numel()
        out = torch.empty(m, self.D, device=self.device)
        if m:
            _hip.call("ebn_gather_rows_f32", _hip.ptr(local_rows), _hip.ptr(self.table), _hip.ptr(out), m, self.D,
                      self.table.shape[0], None, -1, ctypes.c_float(0.0), _hip.ptr(self.oob_flag), _hip.stream_handle())
        return out

    def _local_scatter_add(self, local_rows: torch.Tensor, grads: torch.Tensor) -> None:
        m = local_rows.numel()
        if m:
            _hip.call("ebn_embedding_grad_scatter_f32", _hip.ptr(local_rows), _hip.ptr(grads), _hip.ptr(self.table_grad),
                      m, self.D, self.table.shape[0], None, -1, ctypes.c_float(0.0), _hip.stream_handle())

    # ------------------------------------------------------------------ one-graph multi-rank step: self-check
    def _state_tensors(self):
        ts = [self.params.data, self.params.grad, self.params.m, self.params.v, self.state, self.loss_dev]
        if self.train_embedding:
            ts += [self.table, self.table_grad, self.table_m, self.table_v] + ([self.table_acc] if self.deterministic else [])
        return ts

    def verify_graph_collectives(self, his, pred, y) -> bool:
        """Multi-rank: decide whether the WHOLE step -- kernels and collectives -- may run as one hipGraph (`graph_collectives`).
        Runs ONE training step from the current state twice, as the default form (hipGraph segments with eager collectives between
        the replays) and as the one-graph form (collectives captured: no eager launches, no cross-stream fork / join between
        replays), and compares every parameter, Adam moment, gradient and the loss BIT FOR BIT; the state (weights, moments, step
        counter, dropout keys) is restored in between and afterwards, so the check leaves no trace.  The verdict is MIN-reduced
        over the group: every rank adopts the one-graph form or none does.  Any exception while capturing or replaying the
        one-graph form counts as a failed check.  A collective (all ranks call it with their own batch of the same shape)."""
        if not (self.use_graph and self.graph_capable and self.multi):
            self.graph_collectives = False
            return False
        if torch.distributed.get_backend(self.pg) != "nccl":
            # only RCCL's collectives are stream operations that a hipGraph can hold; gloo's run on the host (attempting to capture
            # one invalidates the capture and leaves the stream unusable) -- nothing to try, the segment form stays
            self.graph_collectives = False
            return False
        torch.cuda.synchronize()
        snap = [t.clone() for t in self._state_tensors()]

        def restore():
            for t, s in zip(self._state_tensors(), snap):
                t.copy_(s)

        def one_step(flag):
            self.graph_collectives = flag
            self._graphs.clear()
            self.train_step(his, pred, y)   # captures
            restore()
            self.train_step(his, pred, y)   # replays from the same state
            torch.cuda.synchronize()
            out = [t.clone() for t in self._state_tensors()]
            restore()
            return out

        ok = True
        try:
            ref = one_step(False)
            got = one_step(True)
            ok = all(torch.equal(a, b) for a, b in zip(ref, got))
        except Exception:  # capture of a collective refused, a replay failed: keep the segment form
            ok = False
            restore()
        verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        torch.distributed.all_reduce(verdict, op=torch.distributed.ReduceOp.MIN, group=self.pg)
        self.graph_collectives = bool(int(verdict.item()))
        self._graphs.clear()
        return self.graph_collectives

    def _capture(self, B, C, advanced=False):
        """Runs of kernel-only segments become hipGraphs; collectives stay eager launches between the replays."""
        torch.cuda.synchronize()
        segs, run, pool, i = self._segments(B, C, advanced), [], None, 0
        desc = self.__dict__.setdefault("_graph_desc", {}).setdefault((B, C, advanced), [])
        desc.clear()
        if self.graph_collectives and self.multi:
            # the collectives are captured too (RCCL supports stream capture): the whole multi-rank step is ONE graph, no
            # eager launches and no cross-stream joins between replays
            g = torch.cuda.CUDAGraph()
            with _hip.capture(g):
                for _kind, fn in segs:
                    fn()
            self._graph_objs = getattr(self, "_graph_objs", []) + [g]
            self._graphs[(B, C, advanced)] = [g.replay]
            desc.append("ONE hipGraph: " + " | ".join(self.SEG_KINDS[k] for k, _fn in segs))
            return self._graphs[(B, C, advanced)]
        while i < len(segs):
            if segs[i][0] != "k":  # "c" | "a" | "w": collectives (and the wait for them) stay eager launches between the replays
                run.append(segs[i][1])
                desc.append(f"{self.SEG_KINDS[segs[i][0]]}: {getattr(segs[i][1], 'what', 'unnamed')} (segment {i} of the step's {len(segs)})")
                i += 1
                continue
            j = i
            while j < len(segs) and segs[j][0] == "k":
                j += 1
            g = torch.cuda.CUDAGraph()
            with _hip.capture(g, pool=pool):  # (thread_local error mode, garbage collector held off: see _hip.capture)
                for _kind, fn in segs[i:j]:
                    fn()
            pool = pool or g.pool()
            run.append(g.replay)
            desc.append(f"hipGraph replay of kernel segments {i}..{j - 1} of the step's {len(segs)}")
            self._graph_objs = getattr(self, "_graph_objs", []) + [g]
            i = j
        self._graphs[(B, C, advanced)] = run
        return run

    def _segments(self, B, C, advanced=False):
        """One training step as an ordered list of (kind, fn): "k" = kernels only (captured into hipGraphs), "c" = collective the
        step waits for, "a" = collective started asynchronously (RCCL runs it on its own stream after everything enqueued so far;
        the following kernels do not wait), "w" = wait for every started collective.

        Multi-rank data parallel: the dense gradients travel as ONE flat bucket.  With a trainable table it is started
        asynchronously after the dWqkv GEMM and runs under the dX GEMM and the table-gradient accumulation; with a frozen table
        nothing follows dWqkv and it is issued in place.  The same collective on the same buffer either way: the overlapped step
        is bit-identical to the serial one."""
        N = B * (self.H + C)
        nb, _ub = self._train_bufs(B, C)
        multi = self.multi
        segs = [] if advanced else [("k", lambda: _hip.call("ebn_step_advance", _hip.ptr(self.state), BETA1, BETA2, _hip.stream_handle()))]
        if self._planned:
            segs += self._lookup_segments(nb, N)
        sparse = self._sparse_dp(N * self.T)
        if multi and self.mlp is None and self.overlap_collectives and self.train_embedding:
            # trainable table: the flat bucket of dense gradients is complete after the dWqkv GEMM and travels under the dX GEMM and the
            # table-gradient accumulation (~430 us at c4).  ONE extra graph boundary: every asynchronous collective costs a
            # cross-stream fork / join between graph replays -- measured with identity collectives on a 1-rank RCCL group
            # (tools/overlap_split_probe.py): +24 us at c4 for this cut, +30-35 us more for a finer one that would also start
            # the gradients finished before the attention backward under it -- as much as that 2.6 MB all-reduce is expected to
            # take.  With a frozen table nothing follows dWqkv, so the step stays graph | bucket | graph.
            segs.append(("k", lambda: (self._fwd_bwd_kernels(B, C, sparse, part="a"), self._fwd_bwd_kernels(B, C, sparse, part="b"))))
            segs.append(_named("a", lambda: self._allreduce_async(self.params.grad), "all-reduce of the flat dense-gradient bucket"))
            segs.append(("k", lambda: self._fwd_bwd_kernels(B, C, sparse, part="c")))
            if self._planned:
                segs += self._table_grad_segments(nb, N)
            if self.exchange is None and not sparse:
                segs.append(_named("c", lambda: self._allreduce_table_grad(), "all-reduce of the dense (V, D) table gradient"))
            if sparse:
                segs += self._sparse_table_grad_segments(nb, N)
            segs.append(("w", self._wait_collectives))
        else:
            segs.append(("k", lambda: self._fwd_bwd_kernels(B, C, sparse)))
            if self._planned and self.train_embedding:
                segs += self._table_grad_segments(nb, N)
            if multi:
                segs.append(_named("c", lambda: self._allreduce_grads(dense_table=not sparse),
                                   "all-reduce of the flat dense-gradient bucket" + (" and of the dense table gradient" if (self.train_embedding and self.exchange is None and not sparse) else "")))
            if sparse:
                segs += self._sparse_table_grad_segments(nb, N)
        segs.append(("k", lambda: self._optimizer_kernels(from_acc=self._adam_from_acc or sparse)))
        return segs

    def _allreduce_async(self, t):
        if self.multi and not self.skip_collectives:
            self._pending.append(torch.distributed.all_reduce(t, group=self.pg, async_op=True))

    def _wait_collectives(self):
        for w in self._pending:
            w.wait()  # RCCL: the compute stream waits for the collective's stream (no host block); gloo: the host waits
        self._pending = []

    def _allreduce_table_grad(self):
        if self.multi and not self.skip_collectives:
            torch.distributed.all_reduce(self.table_grad, group=self.pg)

    def _sparse_table_grad_segments(self, nb, N):
        n_tok, W = N * self.T, self.world
        if getattr(nb, "ids_all", None) is None or nb.ids_all.numel() < W * n_tok:
            nb.ids_all = torch.empty(W * nb.n_seq * self.T, dtype=torch.int32, device=self.device)
            nb.dX_all = torch.empty(W * nb.n_seq * self.T, self.D, device=self.device)
        ids_all, dX_all = nb.ids_all[: W * n_tok], nb.dX_all[: W * n_tok]
        site, p = (0, self.p) if self.p > 0 else (-1, 0.0)

        def gather():
            if self.skip_collectives:
                return
            torch.distributed.all_gather_into_tensor(ids_all, nb.ids[:n_tok], group=self.pg)
            torch.distributed.all_gather_into_tensor(dX_all, nb.dX[:n_tok], group=self.pg)

        def accumulate():  # one launch per rank's slab: the dropout mask of d(X) is indexed by the position in THAT rank's batch
            for r in range(W):
                self._accumulate_fixed(ids_all[r * n_tok:], dX_all[r * n_tok:], n_tok, _hip.ptr(self.state), site, p)

        return [_named("c", gather, "all-gather of the per-token (id, gradient row) pairs (sparse table-gradient exchange)"), ("k", accumulate)]

    def _allreduce_grads(self, dense_table=True):
        """Data-parallel gradient all-reduce over RCCL (SUM; the 1/world is folded into Adam): one flat bucket."""
        if self.multi and not self.skip_collectives:
            torch.distributed.all_reduce(self.params.grad, group=self.pg)
            # (a sharded table's gradients already sit at their owner; the sparse exchange all-gathers token rows instead)
            if self.train_embedding and self.exchange is None and dense_table:
                torch.distributed.all_reduce(self.table_grad, group=self.pg)

