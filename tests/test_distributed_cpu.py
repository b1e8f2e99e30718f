"""world_size-2 (and 3) gloo tests of the multi-GPU plumbing on CPU: the data-parallel
gradient all-reduce and the row-sharded-table routing.  The local gather / scatter-add that the engine does
with HIP kernels are replaced by torch stand-ins here -- only the routing is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ebrec.models.newsrec._dist import ShardedTableExchange, allreduce_sum_, row_shard_range, rows_per_rank, world_info


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_the_table():
    for V, W in ((250002, 8), (10, 3), (5, 8), (32000, 2)):
        spans = [row_shard_range(V, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == V
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) == rows_per_rank(V, W)
    assert world_info() == (0, 1)


def _sharded_lookup_worker(rank, world, mode, V, D):
    full = torch.from_numpy(np.random.default_rng(0).standard_normal((V, D)).astype(np.float32))
    ex = ShardedTableExchange(V, D, mode=mode)
    lo, hi = ex.lo, ex.hi
    shard = full[lo:hi].clone()
    g = torch.Generator().manual_seed(100 + rank)
    n_tok = 500 + 37 * rank  # ragged across ranks
    ids = torch.randint(0, V, (n_tok,), generator=g, dtype=torch.int32)
    ids[:50] = 0  # hot row owned by rank 0, requested by everyone
    if rank == world - 1:
        ids[-10:] = V - 1
    plan = ex.plan(ids)
    assert sum(plan.send_counts) == plan.uniq.numel() and plan.recv_local.numel() == sum(plan.recv_counts)
    assert plan.recv_local.numel() == 0 or (int(plan.recv_local.min()) >= 0 and int(plan.recv_local.max()) < hi - lo)
    rows = ex.lookup(plan, lambda loc: shard[loc.long()])
    assert torch.equal(rows, full[plan.uniq])  # bit-exact: rows only move
    assert torch.equal(rows[plan.inv.long()], full[ids.long()])
    # backward: every rank contributes d(rows); owners accumulate
    d_uniq = torch.ones(plan.uniq.numel(), D) * (rank + 1)
    shard_grad = torch.zeros(hi - lo, D)
    if mode == "alltoall_exact":
        ex.scatter_grads(plan, d_uniq, lambda loc, gr: shard_grad.index_add_(0, loc.long(), gr))
        dense = torch.zeros(V, D)
        dense[plan.uniq] = d_uniq
        dist.all_reduce(dense)  # reference: dense all-reduce of per-rank gradients
        assert torch.allclose(shard_grad, dense[lo:hi])


@pytest.mark.parametrize("mode", ["alltoall_exact", "allgather"])
@pytest.mark.parametrize("world,V", [(2, 1001), (3, 64)])
def test_row_sharded_lookup_routes_rows_exactly(mode, world, V):
    _run(_sharded_lookup_worker, world, mode, V, 12)


def torch_plan(ex, ids, n_tok, cap, ws, slot_rows, inv, counts):
    """torch stand-in for ebn_shard_plan_i32 (the contract of include/ebnerd_hip.h, restated with sorts)."""
    W, per, V = ex.world, ex.per, ex.V
    ids = ids[:n_tok].long()
    slot_rows.fill_(-1)
    inv[:n_tok] = -1
    counts[:W].zero_()  # the two flag words behind the per-owner counts are sticky: only ever raised, the caller clears them
    ok = (ids >= 0) & (ids < V)
    if (~ok).any():
        counts[W + 1] = 1
    owner = torch.where(ok, (ids % W) if ex.cyclic else (ids // per), torch.zeros_like(ids))
    local = torch.where(ok, (ids // W) if ex.cyclic else (ids - owner * per), torch.zeros_like(ids))
    for o in range(W):
        mine = ok & (owner == o)
        rows = torch.unique(local[mine], sorted=True)
        counts[o] = rows.numel()
        if rows.numel() > cap:
            counts[W] = 1
        kept = rows[:cap]
        slot_rows[o * cap: o * cap + kept.numel()] = kept.int()
        pos = torch.searchsorted(kept, local[mine]) if kept.numel() else torch.zeros(int(mine.sum()), dtype=torch.long)
        hit = (pos < kept.numel())
        hit[hit.clone()] = kept[pos[hit]] == local[mine][hit]
        tok = torch.nonzero(mine).reshape(-1)
        inv[tok[hit]] = (o * cap + pos[hit]).int()


def _planned_lookup_worker(rank, world, partition, V, D, factor):
    """Device-planned fixed-capacity exchange (mode 'alltoall') with torch stand-ins for the three HIP kernels: the
    routing, the equal-split all-to-alls and the backward must reproduce a plain table lookup / dense gradient."""
    from ebrec.models.newsrec._dist import PlannedBuffers

    full = torch.from_numpy(np.random.default_rng(0).standard_normal((V, D)).astype(np.float32))
    ex = ShardedTableExchange(V, D, partition=partition, capacity_factor=factor)
    shard = ex.shard_of(full).clone()
    assert shard.shape[0] == ex.n_local
    g = torch.Generator().manual_seed(100 + rank)
    n_tok = 400
    ids = torch.randint(0, V, (n_tok,), generator=g, dtype=torch.int32)
    ids[:50] = 0  # hot row owned by rank 0, requested by everyone
    ids[-7:] = V - 1
    b = PlannedBuffers(ex, n_tok, "cpu", need_grad=True, ws_ints=1)
    plan = lambda *a: torch_plan(ex, *a)

    def gather(local_rows, out):
        okr = local_rows >= 0
        out.zero_()
        out[okr] = shard[local_rows[okr].long()]

    ex.planned_lookup(ids, n_tok, b, plan, gather)
    ex.check(b)
    assert int(b.inv.min()) >= 0
    assert torch.equal(b.rows[b.inv.long()], full[ids.long()])  # bit-exact: rows only move
    # the request lists are distinct and ascending per owner, padded with -1
    cap = ex.capacity(n_tok)
    for o in range(world):
        lst = b.slot_rows[o * cap: (o + 1) * cap]
        k = int(b.counts[o])
        assert (lst[k:] == -1).all() and (lst[:k] >= 0).all() and (lst[1:k] > lst[: max(k - 1, 0)]).all()
    # backward: d(token) = rank+1 everywhere; owners accumulate what every rank sends
    shard_grad = torch.zeros_like(shard)

    def reduce_local(inv, d_slot):
        d_slot.zero_()
        d_slot.index_add_(0, inv.long(), torch.full((n_tok, D), float(rank + 1)))

    def accumulate(local_rows, grads):
        okr = local_rows >= 0
        shard_grad.zero_()
        shard_grad.index_add_(0, local_rows[okr].long(), grads[okr])

    ex.planned_scatter_grads(n_tok, b, reduce_local, accumulate)
    dense = torch.zeros(V, D)
    dense.index_add_(0, ids.long(), torch.full((n_tok, D), float(rank + 1)))
    dist.all_reduce(dense)  # reference: dense all-reduce of per-rank gradients
    assert torch.equal(shard_grad, ex.shard_of(dense))
    # unshard(shard_of(x)) == x
    pad = torch.zeros(ex.per, D)
    pad[: shard.shape[0]] = shard
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    assert torch.equal(ex.unshard(parts), full)
    st = ex.stats()
    assert st["bytes_sent_per_lookup_remote"]["rows"] == (world - 1) * cap * D * 4


@pytest.mark.parametrize("partition", ["block", "cyclic"])
@pytest.mark.parametrize("world,V", [(2, 1001), (3, 64)])
def test_device_planned_exchange_routes_rows_exactly(partition, world, V):
    _run(_planned_lookup_worker, world, partition, V, 12, float(world))


def _planned_overflow_worker(rank, world):
    from ebrec.models.newsrec._dist import PlannedBuffers

    V, D, n_tok = 100000, 4, 4096
    ids = torch.arange(n_tok, dtype=torch.int32)  # every id distinct and low: frequency-ordered tokenizer ids look like this
    # the DEFAULT partition of the device-planned mode is cyclic: low ids spread over the owners, nothing overflows
    ex_default = ShardedTableExchange(V, D, capacity_factor=1.0)
    assert ex_default.partition == "cyclic" and ShardedTableExchange(V, D, mode="alltoall_exact").partition == "block"
    bd = PlannedBuffers(ex_default, n_tok, "cpu", need_grad=False, ws_ints=1)
    ex_default.planned_lookup(ids, n_tok, bd, lambda *a: torch_plan(ex_default, *a), lambda rows, out: out.zero_())
    ex_default.check(bd)
    # block split: all of them are owned by rank 0: 4096 > capacity 2048
    ex = ShardedTableExchange(V, D, partition="block", capacity_factor=1.0)
    assert ex.capacity(n_tok) == 2048
    b = PlannedBuffers(ex, n_tok, "cpu", need_grad=False, ws_ints=1)
    first = ids if rank == 0 else ids % 1000  # rank 1's own requests fit (1000 distinct rows)
    ex.planned_lookup(first, n_tok, b, lambda *a: torch_plan(ex, *a), lambda rows, out: out.zero_())
    # the flags are STICKY: a later clean step (ids that fit) must not hide the overflow from the once-per-epoch check,
    # and check() is a collective -- rank 1 only ever sees clean steps here, yet must raise together with rank 0
    clean = torch.arange(1000, dtype=torch.int32) * 7
    ex.planned_lookup(clean, 1000, b, lambda *a: torch_plan(ex, *a), lambda rows, out: out.zero_())
    with pytest.raises(RuntimeError, match="overflowed"):
        ex.check(b)
    ex.check(b)  # cleared by the read
    ids[5] = V  # out of range beats overflow
    ex.planned_lookup(ids, n_tok, b, lambda *a: torch_plan(ex, *a), lambda rows, out: out.zero_())
    with pytest.raises(IndexError):
        ex.check(b)


def test_planned_exchange_flags_overflow_and_out_of_range_ids():
    _run(_planned_overflow_worker, 2)


def _out_of_range_worker(rank, world):
    ex = ShardedTableExchange(10, 4, mode="alltoall_exact")
    with pytest.raises(IndexError):
        ex.plan(torch.tensor([1, 10]))


def test_sharded_plan_rejects_out_of_range_ids():
    _run(_out_of_range_worker, 2)


def _data_parallel_worker(rank, world):
    """Each rank back-propagates ITS half of the batch (oracle gradients stand in for the HIP backward);
    all-reduce(SUM) with the 1/world scale must equal the full-batch gradient of the mean loss."""
    from oracle import nrms_numpy as on

    rng = np.random.default_rng(3)
    V, D, h, d, A, B, H, C, T = 30, 8, 2, 4, 5, 4, 3, 3, 4
    P = on.random_nrms_params(V, D, h, d, A, seed=2)
    his, pred = rng.integers(0, V, (B, H, T)), rng.integers(0, V, (B, C, T))
    y = np.eye(C)[rng.integers(0, C, B)]
    _, _, g_full = on.nrms_loss_and_grads(his, pred, y, P, h, d)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    _, _, g_loc = on.nrms_loss_and_grads(his[sl], pred[sl], y[sl], P, h, d)
    dense = torch.from_numpy(np.concatenate([g_loc[k].reshape(-1) for k in on.PARAM_ORDER[1:]]))
    table = torch.from_numpy(g_loc["emb"].copy())
    allreduce_sum_([dense, table])
    want = np.concatenate([g_full[k].reshape(-1) for k in on.PARAM_ORDER[1:]])
    np.testing.assert_allclose(dense.numpy() / world, want, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(table.numpy() / world, g_full["emb"], rtol=1e-10, atol=1e-14)


def test_data_parallel_allreduce_equals_full_batch_gradient():
    _run(_data_parallel_worker, 2)


# ---------------------------------------------------------------- lock-step guard of the collective model APIs
def _guard_worker(rank, world):
    import time

    from ebrec.models.newsrec._dist import LockStepGuard

    t0 = time.time()
    g = LockStepGuard(timeout_s=2.0)
    assert (g.rank, g.world) == (rank, world) and time.time() - t0 < 1.0  # construction builds nothing and talks to nobody
    if rank == 0:
        LockStepGuard()  # an engine (hence a guard) made on ONE rank only -- a rank-0 inference model -- must not disturb the others'
    assert g.status().startswith("active")
    g.enter("model.evaluate()")  # everyone arrives: passes
    g.enter("model.save_weights()")
    if rank == 0:  # the `if rank == 0: model.save_weights(...)` mistake: an error naming the call and the absent rank, not a hang
        t0 = time.time()
        with pytest.raises(RuntimeError, match=r"model\.save_weights\(\) is a COLLECTIVE.*rank\(s\) \[1\] did not"):
            g.enter("model.save_weights()")
        assert time.time() - t0 < 30
    else:
        time.sleep(4.0)
    # the failed rendezvous ran on the store: the caller's group is still usable after the error ...
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    assert float(t) == 3.0
    # ... and so is the guard (round-4 ADVICE: a gloo side group stayed broken after one timed-out barrier): the rank that gave
    # up withdrew its arrival mark, the next call every rank makes passes -- also through a second guard over the same group
    g.enter("model.fit()")
    LockStepGuard(timeout_s=2.0).enter("model.evaluate()")
    dist.barrier()


def _guard_default_timeout_worker(rank, world):
    from datetime import timedelta

    from ebrec.models.newsrec._dist import LockStepGuard, group_timeout_s

    os.environ.pop("EBN_COLLECTIVE_TIMEOUT_S", None)
    g = LockStepGuard()
    g.enter("model.fit()")
    # never stricter than the process group's own watchdog (gloo here: 30 min by default)
    assert g.timeout_s == group_timeout_s() >= 600.0
    sub = dist.new_group(ranks=[0, 1], backend="gloo", timeout=timedelta(seconds=77))
    assert group_timeout_s(sub) == 77.0


def _guard_race_worker(rank, world):
    """Round-5 ADVICE: a rank timing out in the instant its late peer arrives.  Whatever the timing, the ranks must AGREE (both pass
    or both raise) and stay in step: the next guarded call both make passes at once."""
    import time

    from ebrec.models.newsrec._dist import LockStepGuard

    g = LockStepGuard(timeout_s=1.0)
    g.enter("warm-up")  # store connection, key prefix
    outcomes = []
    for delay in (0.93, 0.97, 1.0, 1.03, 1.07, 1.4):
        dist.barrier()
        if rank == 1:
            time.sleep(delay)
        try:
            g.enter(f"model.fit() [late peer, {delay}]")
            mine = 1
        except RuntimeError:
            mine = 0
        both = [None, None]
        dist.all_gather_object(both, mine)
        assert both[0] == both[1], (delay, both)  # never one rank through and the other one raising
        outcomes.append(both[0])
        if rank == 0 and not mine:
            time.sleep(0.3)
        t0 = time.time()
        g.enter("the call after")  # counters in step: passes as soon as both are here
        assert time.time() - t0 < 0.9, (delay, time.time() - t0)
    assert outcomes[-1] == 0  # 1.4 s late against a 1 s deadline (+ nobody to have bumped `passed`): both raise


def test_a_rank_timing_out_while_its_peer_arrives_never_splits_the_ranks():
    _run(_guard_race_worker, 2)


def test_lock_step_guard_default_timeout_is_the_process_groups_own():
    _run(_guard_default_timeout_worker, 2)


def test_collective_api_entered_by_one_rank_raises_instead_of_hanging():
    _run(_guard_worker, 2)


def test_lock_step_guard_is_a_no_op_without_a_process_group():
    from ebrec.models.newsrec._dist import LockStepGuard

    g = LockStepGuard()
    assert g.world == 1 and g.status().startswith("not needed")
    g.enter("anything")


# ---------------------------------------------------------------- BatchNormalization moving statistics under data parallel
def _bn_sync_worker(rank, world):
    from ebrec.models.newsrec._mlp import MLPStack

    units = [24, 20, 7]
    mlp = MLPStack(None, "", 16, units, torch.device("cpu"), 0.0)
    assert all(float(m.abs().max()) == 0 for m in mlp.bn_mean) and all(float((v - 1).abs().max()) == 0 for v in mlp.bn_var)  # Keras' initial values
    def drifted(r):  # every rank's moving averages have drifted by ITS data
        rng = np.random.default_rng(1000 + r)
        return [(rng.standard_normal(u).astype(np.float32), (1 + rng.random(u)).astype(np.float32)) for u in units]

    for l, (m, v) in enumerate(drifted(rank)):
        mlp.bn_mean[l].copy_(torch.from_numpy(m))
        mlp.bn_var[l].copy_(torch.from_numpy(v))
    everyone = [drifted(r) for r in range(world)]
    mlp.sync_moving_statistics()
    for l, u in enumerate(units):
        for k, got in ((0, mlp.bn_mean[l]), (1, mlp.bn_var[l])):
            want = np.mean([everyone[r][l][k].astype(np.float64) for r in range(world)], axis=0)
            assert np.allclose(got.numpy(), want, rtol=1e-6, atol=1e-7)
    # the replicas now hold the SAME statistics (bit for bit: one all-reduce result, one scaling), and they are views of one flat buffer
    flat = mlp.bn_stats.clone()
    got = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    assert all(torch.equal(g, got[0]) for g in got)
    assert mlp.bn_mean[1].data_ptr() - mlp.bn_stats.data_ptr() == 4 * (64 + 64) and mlp.bn_var[2].numel() == 7


@pytest.mark.parametrize("world", [2, 3])
def test_batchnorm_moving_statistics_are_averaged_over_the_ranks(world):
    _run(_bn_sync_worker, world)
