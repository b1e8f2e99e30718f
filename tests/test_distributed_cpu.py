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
    if mode == "alltoall":
        ex.scatter_grads(plan, d_uniq, lambda loc, gr: shard_grad.index_add_(0, loc.long(), gr))
        dense = torch.zeros(V, D)
        dense[plan.uniq] = d_uniq
        dist.all_reduce(dense)  # reference: dense all-reduce of per-rank gradients
        assert torch.allclose(shard_grad, dense[lo:hi])


@pytest.mark.parametrize("mode", ["alltoall", "allgather"])
@pytest.mark.parametrize("world,V", [(2, 1001), (3, 64)])
def test_row_sharded_lookup_routes_rows_exactly(mode, world, V):
    _run(_sharded_lookup_worker, world, mode, V, 12)


def _out_of_range_worker(rank, world):
    ex = ShardedTableExchange(10, 4)
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
