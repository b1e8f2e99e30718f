"""Multi-GPU plumbing of the NRMS path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two things shard:
  * impressions (data parallel): replicas all-reduce their flat gradient buckets once per step;
  * optionally the word-embedding table by ROWS (BASELINE.json config 5): rank r owns the contiguous
    block [r*ceil(V/W), (r+1)*ceil(V/W)).  A lookup is then
        dedup local token ids -> route each unique id to its owner (all-to-all of ids)
        -> owners gather their rows (the HIP gather kernel on the local shard)
        -> all-to-all of the rows back -> expand to token order (HIP gather again, dropout fused).
    Only the rows a rank actually needs cross xGMI (n_unique x D x 4 B, 7/8 of it remote), instead of
    the W-fold volume of all-gathering every rank's looked-up rows; the all-gather form is kept as
    ``mode="allgather"`` because it is the form BASELINE.json names and it validates the routed one.
    Backward runs the same routes in reverse and scatter-adds into the owner's shard, so table
    gradients are never all-reduced.

Nothing here computes on rows: the local gather / scatter-add are callables supplied by the engine
(HIP kernels); the CPU tests pass torch stand-ins to exercise the routing under gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


def world_info(group=None) -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def allreduce_sum_(tensors, group=None) -> None:
    """In-place SUM all-reduce of each bucket (the 1/world factor is folded into the Adam kernel)."""
    _, world = world_info(group)
    if world == 1:
        return
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def rows_per_rank(V: int, world: int) -> int:
    return -(-V // world)


def row_shard_range(V: int, world: int, rank: int) -> tuple[int, int]:
    per = rows_per_rank(V, world)
    return min(rank * per, V), min((rank + 1) * per, V)


@dataclass
class LookupPlan:
    uniq: torch.Tensor         # (n_u,) distinct global row ids this rank needs, ascending
    inv: torch.Tensor          # (n_tok,) int32: token -> position in uniq
    send_counts: list          # ids requested FROM each rank (uniq is sorted => already grouped by owner)
    recv_counts: list          # ids each rank requests from me
    recv_local: torch.Tensor   # (sum recv,) int32 LOCAL row numbers (global id - my first row) to serve


class ShardedTableExchange:
    def __init__(self, V: int, D: int, group=None, mode: str = "alltoall"):
        if mode not in ("alltoall", "allgather"):
            raise ValueError(f"unknown exchange mode {mode}")
        self.V, self.D, self.group, self.mode = int(V), int(D), group, mode
        self.rank, self.world = world_info(group)
        self.per = rows_per_rank(self.V, self.world)
        self.lo, self.hi = row_shard_range(self.V, self.world, self.rank)

    # ------------------------------------------------------------------ routing
    def plan(self, ids: torch.Tensor) -> LookupPlan:
        ids = ids.reshape(-1).to(torch.int64)
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.V):
            raise IndexError(f"token id out of range [0, {self.V}) for the embedding table")
        uniq, inv = torch.unique(ids, sorted=True, return_inverse=True)
        owner = torch.div(uniq, self.per, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        if self.world > 1:
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        else:
            recv_counts.copy_(send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        recv_ids = torch.empty(sum(rc), dtype=torch.int64, device=ids.device)
        if self.world > 1:
            dist.all_to_all_single(recv_ids, uniq.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        else:
            recv_ids.copy_(uniq)
        return LookupPlan(uniq, inv.to(torch.int32), sc, rc, (recv_ids - self.lo).to(torch.int32))

    # ------------------------------------------------------------------ forward: rows of plan.uniq
    def lookup(self, plan: LookupPlan, local_gather) -> torch.Tensor:
        """local_gather(local_rows int32 (m,)) -> (m, D) rows of this rank's shard.  Returns (n_u, D)."""
        if self.mode == "allgather":
            return self._lookup_allgather(plan, local_gather)
        served = local_gather(plan.recv_local)
        out = torch.empty(plan.uniq.numel(), self.D, dtype=served.dtype, device=served.device)
        if self.world > 1:
            dist.all_to_all_single(out, served.contiguous(), output_split_sizes=plan.send_counts,
                                   input_split_sizes=plan.recv_counts, group=self.group)
        else:
            out.copy_(served)
        return out

    def _lookup_allgather(self, plan: LookupPlan, local_gather) -> torch.Tensor:
        """Validation form: every rank all-gathers all requests, serves the rows it owns (zeros elsewhere),
        and a SUM all-reduce assembles them; each rank keeps its own slice."""
        n = torch.tensor([plan.uniq.numel()], dtype=torch.int64, device=plan.uniq.device)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(counts, n, group=self.group)
        else:
            counts[0].copy_(n)
        counts = [int(c.item()) for c in counts]
        padded = torch.full((max(counts),), -1, dtype=torch.int64, device=plan.uniq.device)
        padded[: plan.uniq.numel()] = plan.uniq
        gathered = [torch.empty_like(padded) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(gathered, padded, group=self.group)
        else:
            gathered[0].copy_(padded)
        allreq = torch.cat([g[:c] for g, c in zip(gathered, counts)])
        mine = (allreq >= self.lo) & (allreq < self.hi)
        rows = local_gather((allreq[mine] - self.lo).to(torch.int32))
        full = torch.zeros(allreq.numel(), self.D, dtype=rows.dtype, device=rows.device)
        full[mine] = rows
        if self.world > 1:
            dist.all_reduce(full, group=self.group)
        start = sum(counts[: self.rank])
        return full[start: start + counts[self.rank]].contiguous()

    # ------------------------------------------------------------------ backward: d(rows of plan.uniq)
    def scatter_grads(self, plan: LookupPlan, d_uniq: torch.Tensor, local_scatter_add) -> None:
        """Sends d_uniq (n_u, D) to the owners; local_scatter_add(local_rows int32 (m,), grads (m, D))
        accumulates into this rank's shard gradient (rows may repeat across requesting ranks)."""
        recv = torch.empty(sum(plan.recv_counts), self.D, dtype=d_uniq.dtype, device=d_uniq.device)
        if self.world > 1:
            dist.all_to_all_single(recv, d_uniq.contiguous(), output_split_sizes=plan.recv_counts,
                                   input_split_sizes=plan.send_counts, group=self.group)
        else:
            recv.copy_(d_uniq)
        local_scatter_add(plan.recv_local, recv)
