"""Multi-GPU plumbing of the NRMS path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two things shard:
  * impressions (data parallel): replicas all-reduce their flat gradient buckets once per step;
  * optionally the word-embedding table by ROWS (BASELINE.json config 5): rank r owns the contiguous
    block [r*ceil(V/W), (r+1)*ceil(V/W)) (``partition="block"``) or the ids = r mod W (``"cyclic"``, the default of
    the device-planned mode: tokenizer ids are roughly frequency-ordered, a block split would make rank 0 the owner of
    most of a batch and overflow its request lists).
    A lookup is then
        dedup local token ids -> route each distinct id to its owner (all-to-all of row numbers)
        -> owners gather their rows (the HIP gather kernel on the local shard)
        -> all-to-all of the rows back -> expand to token order (HIP gather again, dropout fused).
    Only the rows a rank actually needs cross xGMI (n_distinct x D x 4 B, 7/8 of it remote), instead of
    the W-fold volume of all-gathering every rank's looked-up rows.  Backward runs the same routes in
    reverse and scatter-adds into the owner's shard, so table gradients are never all-reduced.

    ``mode="alltoall"`` (default) plans the lookup ON THE DEVICE (csrc/ebn_shard.hip) into fixed-capacity
    per-owner request lists: both exchanges are equal-split all-to-alls whose sizes the host knows from the
    batch shape alone -- no host sync, nothing data-dependent on the host, the kernels between the collectives
    are hipGraph-capturable.  Capacity per (requester, owner) pair = ``capacity_factor`` x n_tok / W distinct
    rows (never less than min(n_tok, 1024)); an overflow drops ids, sets a device flag and raises at the next
    ``check()``.  ``mode="alltoall_exact"`` is the variable-size form planned with torch.unique on the host
    (three host syncs per lookup, eager only) and ``mode="allgather"`` the all-gather form BASELINE.json names;
    both are kept because they validate the planned one.

Nothing here computes on rows: the local gather / scatter-add are callables supplied by the engine
(HIP kernels); the CPU tests pass torch stand-ins to exercise the routing under gloo.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


def _named(kind, fn, what):
    fn.what = what
    return kind, fn


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


def group_timeout_s(group=None) -> float:
    """The process group's own collective timeout in seconds (what its watchdog aborts a stuck collective after): the
    backend options' timeout when readable, else torch's default for the backend (RCCL 600 s, gloo 1800 s)."""
    from torch.distributed import distributed_c10d as c10d

    try:
        pg = group if group is not None else c10d._get_default_group()
        backend = dist.get_backend(pg)
        try:
            dev = torch.device("cuda") if backend == "nccl" else torch.device("cpu")
            return float(pg._get_backend(dev).options._timeout.total_seconds())
        except Exception:
            return float(c10d._get_default_timeout(backend).total_seconds())
    except Exception:
        return 600.0


class LockStepGuard:
    """Turns "a collective API entered by only some ranks" from a hang into an error.

    `save_weights`, `evaluate` and `fit` are collectives when world > 1 (gradient / log / flag reductions, the table
    all-gather of a row-sharded engine).  A caller that enters one on rank 0 only -- the natural thing to write after
    `if rank == 0:` -- would block in RCCL until the watchdog aborts the job.  `enter(what)` runs a rendezvous with a
    deadline first: when a peer does not arrive within `timeout_s` it raises a RuntimeError naming the call and the
    missing ranks, on every rank that did arrive.

    The rendezvous lives on the job's c10d STORE (the TCP store behind `init_process_group`), not on a process group:
      * nothing is built at construction -- an engine made on only some ranks (a rank-0-only inference model while
        `torch.distributed` is initialised) costs nothing and never blocks; the first `enter()` does the first store access;
      * a rendezvous that timed out leaves no broken state behind: the ranks that gave up WITHDRAW their arrival marks and do
        not advance their call counter, so the next `enter()` made by every rank passes (round 4's gloo side group closed its
        connection pairs after one failed `monitored_barrier`, so every later guarded call raised too);
      * the caller's RCCL / gloo group is never touched.
    `timeout_s`: the argument, else EBN_COLLECTIVE_TIMEOUT_S, else the process group's OWN collective timeout (RCCL: 600 s) --
    the guard never fires earlier than the watchdog it pre-empts would have, so a rank that tokenizes for minutes before
    `fit()` does not turn a job that used to run into one that raises.  `disabled` holds the reason when no store is reachable
    (bench.py prints it as the `guard` field); a disabled guard lets every call through."""

    _calls = {}  # ranks of the group -> guarded calls passed so far.  Shared by every guard (= engine) of this process over that group:
    # what must be lock-step is the SEQUENCE of collective API calls, whichever model makes them; an engine built on one rank
    # only never enters a collective API, so it cannot put the ranks' counters out of step

    def __init__(self, group=None, timeout_s: float | None = None, force: bool = False):
        self.rank, self.world = world_info(group)
        self.force = bool(force)  # run the rendezvous on a ONE-rank group too (`bench.py --force-dist`: the store path, exercised)
        self.pg = group
        env = os.environ.get("EBN_COLLECTIVE_TIMEOUT_S")
        self.timeout_s = float(timeout_s) if timeout_s is not None else (float(env) if env else None)
        self.disabled = None
        self.poll_s = 0.02
        self._store = None

    def _resolve_timeout(self) -> float:
        if self.timeout_s is None:
            self.timeout_s = group_timeout_s(self.pg)
        return self.timeout_s

    def ensure(self) -> bool:
        """First use: find the store and this guard's key prefix.  Returns whether the guard is active."""
        if (self.world <= 1 and not self.force) or self.disabled is not None:
            return False
        if self._store is None:
            try:
                from torch.distributed import distributed_c10d as c10d

                ranks = dist.get_process_group_ranks(self.pg if self.pg is not None else dist.group.WORLD)
                self._ranks = list(ranks)
                self._gkey = tuple(ranks)
                self._prefix = f"ebn_guard/{'-'.join(map(str, ranks))}"
                store = c10d._get_default_store()
                store.add(f"{self._prefix}/probe", 0)  # a store that cannot be reached fails here, not in the first fit()
                self._store = store
            except Exception as e:  # a diagnostics aid must never keep a job from starting
                self.disabled = f"{type(e).__name__}: {e}"
                return False
        return True

    def status(self) -> str:
        if self.world <= 1 and not self.force:
            return "not needed (one rank)"
        if not self.ensure():
            return f"disabled: {self.disabled}"
        return f"active (c10d store rendezvous, timeout {self._resolve_timeout():.0f} s)"

    def enter(self, what: str) -> None:
        """Rendezvous of call number n of this group.  PASSING and GIVING UP are made consistent across the ranks by a shared
        `passed` counter: a rank passes only after it has (1) seen every rank's mark, (2) bumped `passed`, (3) seen every mark
        AGAIN; a rank that times out withdraws its mark and then reads `passed` -- if a peer has bumped it, everyone had arrived
        and the peer is about to pass (or is waiting in (3) for this rank's mark to come back), so this rank puts its mark back
        and goes on polling instead of raising.  Without that, a rank timing out in the instant a late peer arrives raised alone
        while the peer passed and advanced its call counter: every later guarded call of the two then met on different keys
        (round-5 ADVICE).  The poll backs off from 20 ms to 250 ms after the first second: `world` store reads per poll, for up
        to the group's own timeout, must not load the store."""
        if not self.ensure():
            return
        import time

        timeout = self._resolve_timeout()
        n = LockStepGuard._calls.get(self._gkey, 0)
        key = f"{self._prefix}/{n}"
        mine, passed = f"{key}/r{self.rank}", f"{key}/passed"
        marks = [f"{key}/r{r}" for r in self._ranks]
        st = self._store
        grace = max(1.0, 20 * self.poll_s)  # how long a rank in step (3) waits for a withdrawn mark to come back

        def all_here():
            here = [int(st.add(m, 0)) > 0 for m in marks]
            return all(here), here

        def done():
            if n:  # everyone is at call n, so everyone has left call n-1: its keys can go
                for k in (f"{self._prefix}/{n - 1}/r{self.rank}",) + ((f"{self._prefix}/{n - 1}/passed",) if self.rank == self._ranks[0] else ()):
                    try:
                        st.delete_key(k)
                    except Exception:
                        pass
            LockStepGuard._calls[self._gkey] = n + 1

        st.add(mine, 1)
        t0 = time.monotonic()
        deadline = t0 + timeout
        here = []
        while True:
            ok, here = all_here()
            if ok:
                st.add(passed, 1)
                t1 = time.monotonic()
                while True:  # (3): still everyone?  A mark missing now belongs to a rank that is giving up: it will see `passed`
                    ok2, here = all_here()
                    if ok2:
                        done()
                        return
                    if time.monotonic() - t1 >= grace:
                        break
                    time.sleep(self.poll_s)
                st.add(passed, -1)  # the peer gave up for good before it could see this rank's bump: this rank gives up as well
                break
            now = time.monotonic()
            if now >= deadline:
                st.add(mine, -1)  # withdraw ...
                if int(st.add(passed, 0)) > 0:  # ... unless a peer has seen everyone and is passing: stay in
                    st.add(mine, 1)
                    deadline = now + grace
                    continue
                mine = None
                break
            time.sleep(self.poll_s if now - t0 < 1.0 else max(self.poll_s, 0.25))
        if mine is not None:
            st.add(mine, -1)  # withdraw: the next enter() every rank makes starts from a clean slate
        missing = [r for r, h in zip(self._ranks, here) if not h]
        raise RuntimeError(f"{what} is a COLLECTIVE when world > 1 ({self.world} ranks): every rank of the group must call it at the "
                           f"same point.  Rank {self.rank} entered it, but rank(s) {missing} did not within {timeout:.0f} s -- a call "
                           f"guarded by `if rank == 0:`?  (rank-local inference: predict / scorer.predict / encode_news on a "
                           f"replicated table are not collectives.)")


class SegmentTrace:
    """Where a multi-rank training step is when it stops making progress (bench.py's hang watchdog, debugging).

    A step is a list of segments -- hipGraph replays of kernel runs and the collectives between them (`NRMSEngine._segments`).
    With a trace attached the engine records a HIP event behind every segment of every step; `where()` -- callable from another
    thread while the main one is blocked in a synchronize or a host-side collective -- reports the segment the HOST was
    launching and the first segment whose event the DEVICE has not reached.  (An RCCL collective that a peer never joins does
    not block the host at its launch: the host runs ahead and blocks at the next synchronize, so only the device-side position
    names the segment that hangs.)  Costs one event record per segment; off unless attached."""

    def __init__(self, keep: int = 4096):
        self.keep, self.step, self.host, self.desc, self.events = int(keep), 0, None, [], {}

    def run(self, fns, desc) -> None:
        self.desc, s = desc, self.step
        for i, fn in enumerate(fns):
            self.host = (s, i, "launching")
            fn()
            ev = torch.cuda.Event()
            ev.record()
            self.events[(s, i)] = ev
            self.host = (s, i, "launched")
        self.step += 1
        if len(self.events) > self.keep:
            for k in sorted(self.events)[: len(self.events) - self.keep]:
                del self.events[k]

    def _name(self, i) -> str:
        return self.desc[i] if i < len(self.desc) else "?"

    def where(self) -> str:
        host = "no step launched yet" if self.host is None else \
            f"host: step {self.host[0]}, segment {self.host[1]} [{self._name(self.host[1])}] {self.host[2]}"
        dev = "device: every launched segment has completed"
        for k in sorted(self.events):
            try:
                done = self.events[k].query()
            except Exception as e:  # a device in an error state: say so instead of dying in the watchdog
                dev = f"device: event query failed at step {k[0]}, segment {k[1]} [{self._name(k[1])}]: {type(e).__name__}: {e}"
                break
            if not done:
                dev = f"device: waiting in step {k[0]}, segment {k[1]} [{self._name(k[1])}] (first segment not completed)"
                break
        return f"{host}; {dev}"


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


class PlannedBuffers:
    """Device buffers of the fixed-capacity lookup for up to `n_tok` tokens (allocated once per shape: captured graphs
    and the collectives work on the same memory every step)."""

    def __init__(self, ex: "ShardedTableExchange", n_tok: int, device, need_grad: bool, ws_ints: int):
        W, D, cap = ex.world, ex.D, ex.capacity(n_tok)
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=device)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.n_tok, self.cap = n_tok, cap
        self.ws, self.slot_rows, self.inv, self.counts = i32(max(ws_ints, 1)), i32(W * cap), i32(n_tok), torch.zeros(W + 2, dtype=torch.int32, device=device)
        self.served = f32(W * cap, D)
        # one rank: "exchanging" with yourself is the identity, the receive buffers alias the send buffers
        self.recv_rows = i32(W * cap) if W > 1 else self.slot_rows
        self.rows = f32(W * cap, D) if W > 1 else self.served
        self.d_slot = f32(W * cap, D) if need_grad else None
        self.d_recv = (f32(W * cap, D) if W > 1 else self.d_slot) if need_grad else None


class ShardedTableExchange:
    def __init__(self, V: int, D: int, group=None, mode: str = "alltoall", partition: str | None = None, capacity_factor: float = 1.25):
        if mode not in ("alltoall", "alltoall_exact", "allgather"):
            raise ValueError(f"unknown exchange mode {mode}")
        if partition is None:
            # tokenizer ids are roughly frequency-ordered: under a block split rank 0 would own most of every batch and its
            # request lists would overflow the default capacity, so the device-planned mode spreads ids round-robin by default
            partition = "cyclic" if mode == "alltoall" else "block"
        if partition not in ("block", "cyclic"):
            raise ValueError(f"unknown table partition {partition}")
        if partition == "cyclic" and mode != "alltoall":
            raise ValueError("the cyclic partition is implemented by the device-planned exchange (mode='alltoall') only")
        self.V, self.D, self.group, self.mode, self.partition = int(V), int(D), group, mode, partition
        self.capacity_factor = float(capacity_factor)
        self.rank, self.world = world_info(group)
        self.per = rows_per_rank(self.V, self.world)
        self.lo, self.hi = row_shard_range(self.V, self.world, self.rank)
        self.bytes_sent = {"ids": 0, "rows": 0, "grads": 0}   # per lookup of the last shape, this rank, remote part only
        self.lookups = 0

    # ------------------------------------------------------------------ geometry
    @property
    def cyclic(self) -> bool:
        return self.partition == "cyclic"

    @property
    def n_local(self) -> int:
        """rows this rank owns"""
        if self.cyclic:
            return len(range(self.rank, self.V, self.world))
        return self.hi - self.lo

    def shard_of(self, table):
        """this rank's rows of a full (V, ...) array, in owner-local order"""
        return table[self.rank:: self.world] if self.cyclic else table[self.lo: self.hi]

    def unshard(self, parts):
        """inverse of shard_of over the list of every rank's shard, each padded to `per` rows: the full (V, ...) tensor"""
        if not self.cyclic:
            return torch.cat(parts)[: self.V]
        full = torch.stack(parts, dim=1)  # (per, W, ...): local row l of rank r is global row l*W + r
        return full.reshape(self.per * self.world, *parts[0].shape[1:])[: self.V]

    def capacity(self, n_tok: int) -> int:
        """distinct rows one rank may request from one owner in a lookup of n_tok tokens (a host-side function of the
        SHAPE only: every rank computes the same number without talking)"""
        if self.world == 1:
            return max(min(n_tok, self.V), 1)
        cap = max(int(-(-self.capacity_factor * n_tok // self.world)), min(n_tok, 1024))
        cap = min(cap, n_tok, self.per)
        return max(-(-cap // 64) * 64 if cap >= 64 else cap, 1)

    skip = False  # bench.py only: time a step without its exchanges (results are wrong)

    def _a2a(self, out, inp):
        if self.world > 1 and not self.skip:
            dist.all_to_all_single(out, inp, group=self.group)  # equal splits: sizes are a function of the shape only

    # ------------------------------------------------------------------ device-planned fixed-capacity lookup
    def lookup_segments(self, ids: torch.Tensor, n_tok: int, b: PlannedBuffers, plan_fn, gather_fn) -> list:
        """The lookup as an ordered list of ("k" = kernels only | "c" = collective, fn): the engine captures the "k" runs
        into hipGraphs and launches the "c" entries eagerly between the replays.  ids (>= n_tok,) int32 on the device.
        After the last segment b.rows[(W*cap), D] holds the distinct rows this rank asked for and b.inv[:n_tok] maps
        tokens to them.
          plan_fn(ids, n_tok, cap, ws, slot_rows, inv, counts)   -- ebn_shard_plan_i32
          gather_fn(local_rows (m,) int32, out (m, D))           -- out[i] = shard[local_rows[i]], zero row for -1"""
        W, cap = self.world, self.capacity(n_tok)
        n = W * cap

        def note():
            self.lookups += 1
            self.bytes_sent.update(ids=(W - 1) * cap * 4, rows=(W - 1) * cap * self.D * 4)

        segs = [("k", lambda: plan_fn(ids, n_tok, cap, b.ws, b.slot_rows, b.inv, b.counts))]
        if W > 1:
            segs.append(_named("c", lambda: (self._a2a(b.recv_rows[:n], b.slot_rows[:n]), note()), "all-to-all of the requested row numbers (row-sharded lookup)"))
        segs.append(("k", lambda: gather_fn(b.recv_rows[:n], b.served[:n])))
        if W > 1:
            segs.append(_named("c", lambda: self._a2a(b.rows[:n], b.served[:n]), "all-to-all of the served table rows (row-sharded lookup)"))
        return segs

    def grad_segments(self, n_tok: int, b: PlannedBuffers, reduce_fn, scatter_fn) -> list:
        """Backward of the lookup, same segment form.
          reduce_fn(inv (n_tok,), d_slot (W*cap, D))             -- d_slot = 0; d_slot[inv[t]] += d(token t)
          scatter_fn(local_rows (m,) int32, grads (m, D))        -- shard_grad = 0; shard_grad[local_rows[i]] += grads[i], -1 skipped"""
        W, cap = self.world, self.capacity(n_tok)
        n = W * cap
        segs = [("k", lambda: reduce_fn(b.inv[:n_tok], b.d_slot[:n]))]
        if W > 1:
            segs.append(_named("c", lambda: (self._a2a(b.d_recv[:n], b.d_slot[:n]), self.bytes_sent.update(grads=(W - 1) * cap * self.D * 4)),
                               "all-to-all of the row gradients to their owners (row-sharded table)"))
        segs.append(("k", lambda: scatter_fn(b.recv_rows[:n], b.d_recv[:n])))
        return segs

    def planned_lookup(self, ids, n_tok, b, plan_fn, gather_fn) -> None:
        for _kind, fn in self.lookup_segments(ids, n_tok, b, plan_fn, gather_fn):
            fn()

    def planned_scatter_grads(self, n_tok, b, reduce_fn, scatter_fn) -> None:
        for _kind, fn in self.grad_segments(n_tok, b, reduce_fn, scatter_fn):
            fn()

    def check(self, b: PlannedBuffers, what="embedding table", collective: bool = True) -> None:
        """One host read of the plan's STICKY flags (the engine calls this once per epoch, not per step; the plan kernel
        only ever raises them, this clears them).  A COLLECTIVE when world > 1 (unless collective=False): the flags are
        MAX-reduced over the group first, so that either every rank raises or none does -- a rank raising alone would leave
        the others blocked in the next step's all-to-all."""
        flags = b.counts[self.world:].clone()
        if self.world > 1 and collective:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        c = b.counts[: self.world].cpu().tolist() + flags.cpu().tolist()
        b.counts[self.world:].zero_()
        if c[self.world + 1]:
            raise IndexError(f"token id out of range [0, {self.V}) for the {what} (plan counters {c})")
        if c[self.world]:
            raise RuntimeError(f"row-sharded lookup overflowed its exchange capacity ({b.cap} distinct rows per owner, wanted up to "
                               f"{max(c[: self.world])}): rows were dropped.  Raise capacity_factor (now {self.capacity_factor}; "
                               f"{self.world} can never overflow) or use partition='cyclic' for frequency-ordered vocabularies")

    def stats(self) -> dict:
        return {"mode": self.mode, "partition": self.partition, "world": self.world, "capacity_factor": self.capacity_factor,
                "bytes_sent_per_lookup_remote": dict(self.bytes_sent),
                "note": "per rank and per step: row numbers out, rows back (and row gradients out when the table trains); "
                        "equal-split all-to-alls of fixed capacity, (world-1)/world of each buffer leaves the GPU"}

    # ------------------------------------------------------------------ host-planned exact routing (validation forms)
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
