"""BASELINE.json configs[3] (data parallel, H=50) and configs[4] (row-sharded table) on the GPU box (gpu-marked).

The box has ONE GPU, and RCCL refuses two ranks on one device, so the multi-rank tests run the REAL engine (HIP kernels
through the C ABI, hipGraph segments, the engine's own collectives) with two -- and, for the rank counts BASELINE.json
names, eight -- processes sharing cuda:0 over gloo.  What
is compared is arithmetic, not transport: the averaged update of the ranks must equal the single-process full-batch
step of the float64 oracle.  The 8-GPU RCCL run itself is the driver's (bench.py --gpus 8).
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import nrms_numpy as on
from tests.hip_testutil import assert_close
from tests.test_distributed_cpu import torch_plan
from tests.test_nrms_model import batch, make_hp, weight_list

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- c4: H=50 training parity at the per-GPU shape
@pytest.mark.parametrize("loss", ["cross_entropy_loss", "log_loss"])
def test_c4_shape_training_step_matches_oracle(hip, loss):
    """history_size=50, 20 heads x 20, trainable 300-d table (configs[3] per-GPU shape, smaller batch/vocabulary):
    loss, every dense gradient and the embedding gradient of one training step vs the float64 oracle, then two more
    steps of the Adam trajectory.  L=50 runs the user-level attention kernels that L=20 never touches."""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(history_size=50, dropout=0.2, learning_rate=1e-3, loss=loss)
    V, D, seed, B, C = 2000, 300, 5, 4, 5
    rng = np.random.default_rng(50)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=8)
    m = NRMSModel(hp, word2vec_embedding=P["emb"], seed=seed).from_keras_weight_list(weight_list(P))
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    P0 = {k: v.copy() for k, v in P.items()}
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    eng = m._engine
    eng.keep_table_grad = True  # (the fused accumulator->Adam sweep never writes the fp32 gradient)
    for t in range(1, 4):
        his, pred, y = batch(rng, B, 50, C, hp.title_size, V)
        L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, loss, on.Drop(0.2, seed, t))
        got_L = float(m.train_step(his, pred, y).item())
        assert abs(got_L - L) <= 2e-5 * max(1.0, abs(L)), (t, got_L, L)
        if t == 1:
            for pre in ("n", "u"):
                want = np.concatenate([g[f"{pre}_WQ"], g[f"{pre}_WK"], g[f"{pre}_WV"]], 1)
                assert_close(eng.params.g(f"{pre}_Wqkv").cpu().numpy(), want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what=f"{pre} dWqkv")
                for nm in ("W", "b", "q"):
                    want = g[f"{pre}_{nm}"].reshape(eng.params.shapes[f"{pre}_{nm}"])
                    assert_close(eng.params.g(f"{pre}_{nm}").cpu().numpy(), want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what=f"{pre} d{nm}")
            assert_close(eng.table_grad.cpu().numpy(), g["emb"], rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(g["emb"]).max(), what="dEmb")
        for k in P:
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=1e-3)
    got = dict(zip(on.PARAM_ORDER, m.model.get_weights()))
    for k in on.PARAM_ORDER:
        step = np.abs(P[k] - P0[k])
        assert_close(got[k].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"weights {k} after 3 steps")


# ---------------------------------------------------------------- c5: the device-side lookup plan
@pytest.mark.parametrize("world,cyclic", [(1, 0), (2, 0), (8, 0), (8, 1), (3, 1), (5, 0)])
@pytest.mark.parametrize("V,n_tok,cap", [(250002, 48000, 48000), (1001, 700, 64), (37, 1000, 37), (5000, 0, 8), (4099, 5000, 100)])
def test_shard_plan_kernel_is_the_reference_plan_bit_for_bit(hip, world, cyclic, V, n_tok, cap):
    """ebn_shard_plan_i32 for ANY world size runs on one GPU (the plan is local): request lists, token->slot map,
    per-owner counts, overflow and out-of-range flags vs the sort-based restatement in tests/test_distributed_cpu.py."""
    from ebrec.models.newsrec._dist import ShardedTableExchange

    P_, S_ = hip.ptr, hip.stream_handle
    rng = np.random.default_rng(world * 131 + V)
    ex = ShardedTableExchange.__new__(ShardedTableExchange)  # geometry only (no process group needed for a plan)
    ex.V, ex.world, ex.per, ex.partition = V, world, -(-V // world), "cyclic" if cyclic else "block"
    ids = rng.integers(0, V, n_tok).astype(np.int32)
    if n_tok > 100:
        ids[:40] = 0  # a hot row (padded titles)
        ids[40:60] = V - 1
        ids[77] = V  # one id out of range
        ids[78] = -3
    cap = min(cap, max(n_tok, 1))
    d_ids = torch.from_numpy(ids).cuda() if n_tok else torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(int(hip.lib().ebn_shard_plan_workspace_ints(V, world)), 1), dtype=torch.int32, device="cuda")
    slot = torch.empty(world * cap, dtype=torch.int32, device="cuda")
    inv = torch.full((max(n_tok, 1),), -7, dtype=torch.int32, device="cuda")
    counts = torch.full((world + 2,), -7, dtype=torch.int32, device="cuda")
    counts[world:] = 0  # the caller owns the two sticky flag words: zeroed before the first plan, cleared when read
    hip.call("ebn_shard_plan_i32", P_(d_ids), n_tok, V, world, cyclic, cap, P_(ws), P_(slot), P_(inv), P_(counts), S_())
    w_slot, w_inv, w_counts = torch.empty(world * cap, dtype=torch.int32), torch.full((max(n_tok, 1),), -7, dtype=torch.int32), torch.zeros(world + 2, dtype=torch.int32)
    torch_plan(ex, torch.from_numpy(ids), n_tok, cap, None, w_slot, w_inv, w_counts)
    assert torch.equal(counts.cpu(), w_counts), (counts.cpu().tolist(), w_counts.tolist())
    assert torch.equal(slot.cpu(), w_slot)
    assert torch.equal(inv.cpu()[:n_tok], w_inv[:n_tok])
    # replay on the same buffers (a captured graph does exactly this): the plan re-initialises everything it reads
    hip.call("ebn_shard_plan_i32", P_(d_ids), n_tok, V, world, cyclic, cap, P_(ws), P_(slot), P_(inv), P_(counts), S_())
    assert torch.equal(slot.cpu(), w_slot) and torch.equal(counts.cpu(), w_counts)
    if n_tok > 100:
        # the flags are sticky: a later, clean plan on the same buffers (the next step of an epoch) must not clear what an
        # earlier step raised -- check() reads them once per epoch
        flags = counts[world:].cpu().clone()
        clean = torch.from_numpy(np.clip(ids, 0, V - 1)[:min(n_tok, cap)].copy()).cuda()
        hip.call("ebn_shard_plan_i32", P_(clean), clean.numel(), V, world, cyclic, cap, P_(ws), P_(slot), P_(inv), P_(counts), S_())
        assert flags[1] == 1 and torch.equal(counts[world:].cpu(), flags)


@pytest.mark.parametrize("mode,partition,graph", [("alltoall", "block", False), ("alltoall", "cyclic", False), ("alltoall", "block", True),
                                                  ("alltoall_exact", "block", False), ("allgather", "block", False)])
@pytest.mark.parametrize("train_embedding", [False, True])
def test_every_row_sharded_form_on_one_rank_equals_the_replicated_table(hip, mode, partition, graph, train_embedding):
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(31)
    V = 257
    emb = rng.standard_normal((V, 64)).astype(np.float32)
    a = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=train_embedding, deterministic=False)
    b = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=train_embedding, shard_table=True, shard_mode=mode,
                  shard_partition=partition)
    assert b._engine.graph_capable == (mode == "alltoall")
    if graph:
        b._engine.enable_graphs()
    his, pred, y = batch(rng, 5, hp.history_size, 5, hp.title_size, V)
    assert np.array_equal(a.model.predict((his, pred)), b.model.predict((his, pred)))
    for _ in range(3):
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la))
    b._engine.check_oob()
    for wa, wb in zip(a.model.get_weights(), b.model.get_weights()):
        assert np.allclose(wa, wb, rtol=1e-4, atol=1e-6)
    if mode == "alltoall":
        his[0, 0, 0] = V  # out of range on a device-planned lookup: flagged by the plan, raised at the epoch check
        b._engine.train_step(torch.from_numpy(his.astype(np.int32)).cuda(), torch.from_numpy(pred.astype(np.int32)).cuda(),
                             torch.from_numpy(y.astype(np.float32)).cuda())
        with pytest.raises(IndexError):
            b._engine.check_oob()


@pytest.mark.parametrize("train_embedding,shard,graph", [(False, False, True), (True, False, True), (False, True, True), (True, True, False)])
def test_split_precision_steps_track_the_exact_steps_in_every_table_mode(hip, train_embedding, shard, graph):
    """precision="split" (bf16x6 projections; the gather writes bf16 planes in a training step) in the table modes the engine
    has: frozen / trainable, replicated / row-sharded with the device-planned lookup (the expanding gather then reads the
    exchanged rows), eager and graph-replayed -- three steps against the exact-precision engine on identical inputs:
    losses and weights agree to fp32 accumulation noise, and the frozen table stays bit-identical."""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    rng = np.random.default_rng(57)
    V = 300
    emb = rng.standard_normal((V, 64)).astype(np.float32)
    kw = dict(word2vec_embedding=emb, seed=3, train_embedding=train_embedding, shard_table=shard)
    a, b = NRMSModel(hp, **kw), NRMSModel(hp, precision="split", **kw)
    if graph:
        a._engine.enable_graphs()
        b._engine.enable_graphs()
    for t in range(3):
        his, pred, y = batch(rng, 5, hp.history_size, 5, hp.title_size, V)
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert abs(la - lb) <= 2e-5 * max(1.0, abs(la)), (t, la, lb)
    b._engine.check_oob()
    assert getattr(b._engine._bufs[("news", True)], "planes_rows", -1) > 0  # the training gather did write planes
    for i, (wa, wb) in enumerate(zip(a.model.get_weights(), b.model.get_weights())):
        if i == 0 and not train_embedding:
            assert np.array_equal(wa, wb)
        else:  # Adam turns fp32 noise in tiny gradients into O(lr) differences of single elements: compare like the oracle tests do
            assert np.abs(wa - wb).max() <= 2e-5 + 0.02 * 3e-3, (i, np.abs(wa - wb).max())
    assert np.allclose(a.model.predict((his, pred)), b.model.predict((his, pred)), atol=2e-5)


# ---------------------------------------------------------------- two ranks, one GPU, the real engine
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp

    port = _free_port()
    # every rank also runs the float64 oracle on the GLOBAL batch: keep `world` BLAS pools from oversubscribing the host
    old = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 8) // world))
    try:
        mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)
    finally:
        if old is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = old


def _entry(rank, world, port, fn, args):
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for p in (str(root), str(root / "ebnerd-benchmark_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    torch.cuda.set_device(0)  # both ranks on the one GPU: gloo moves the buffers, the kernels are the product's
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _dp_worker(rank, world, shard, partition, graph, train_embedding, H, table_grad_exchange="auto"):
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(history_size=H, dropout=0.0, learning_rate=1e-3)
    V, D, seed, B, C = 600, 64, 5, 3, 5  # B per rank
    rng = np.random.default_rng(77)
    P = on.random_nrms_params(V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=8)
    m = NRMSModel(hp, word2vec_embedding=P["emb"], seed=seed, train_embedding=train_embedding, shard_table=shard,
                  shard_partition=partition, shard_capacity_factor=float(world), deterministic=not shard,
                  table_grad_exchange=table_grad_exchange)
    m.from_keras_weight_list(weight_list(P))
    eng = m._engine
    assert eng.world == world
    assert eng._sparse_dp(B * (H + C) * hp.title_size) == (table_grad_exchange == "sparse")
    if graph:
        eng.enable_graphs()
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    P0 = {k: v.copy() for k, v in P.items()}
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    for t in range(1, 4):
        his, pred, y = batch(rng, B * world, H, C, hp.title_size, V)  # the GLOBAL batch, identical on every rank
        L, _, g = on.nrms_loss_and_grads(his, pred, y, P, hp.head_num, hp.head_dim, "cross_entropy_loss", None)
        sl = slice(rank * B, (rank + 1) * B)
        L_loc, _, _ = on.nrms_loss_and_grads(his[sl], pred[sl], y[sl], P, hp.head_num, hp.head_dim, "cross_entropy_loss", None)
        got = float(eng.train_step(his[sl], pred[sl], y[sl]).item())  # each rank steps on ITS rows
        assert abs(got - L_loc) <= 2e-5 * max(1.0, abs(L_loc)), (rank, t, got, L_loc)
        if t == 1 and not shard and table_grad_exchange != "sparse":
            # after the all-reduce (SUM) the gradient buffers hold world x the full-batch mean gradient
            want = np.concatenate([g["n_WQ"], g["n_WK"], g["n_WV"]], 1) * world
            assert_close(eng.params.g("n_Wqkv").cpu().numpy(), want, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want).max(), what="all-reduced dWqkv")
        for k in P:
            if k == "emb" and not train_embedding:
                continue
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=1e-3)
    eng.check_oob()
    got = dict(zip(on.PARAM_ORDER, m.model.get_weights()))  # (a collective with a sharded table: gathers the shards)
    for k in on.PARAM_ORDER:
        step = np.abs(P[k] - P0[k])
        assert_close(got[k].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"rank {rank}: weights {k} after 3 DP steps")
    if shard:
        st = eng.exchange.stats()
        assert st["world"] == world and st["bytes_sent_per_lookup_remote"]["rows"] > 0
        if train_embedding:
            assert st["bytes_sent_per_lookup_remote"]["grads"] > 0


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("H", [20, 50])
def test_two_rank_data_parallel_step_equals_the_full_batch_oracle_step(hip, graph, H):
    """configs[3]: the engine's own DP step (hipGraph(fwd+bwd) -> all-reduce -> hipGraph(Adam)) on two ranks; every rank
    must end on the weights of the float64 oracle stepping on the whole global batch."""
    _spawn(_dp_worker, 2, False, "block", graph, True, H)


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_sparse_table_gradient_exchange_equals_the_full_batch_oracle_step(hip, graph):
    """Data parallel with the (id, gradient row) all-gather instead of the dense (V, D) all-reduce: every rank accumulates
    every rank's token rows in the order-independent fixed-point accumulator -> identical replicas, oracle trajectory."""
    _spawn(_dp_worker, 2, False, "block", graph, True, 20, "sparse")


@pytest.mark.parametrize("partition,graph,train_embedding", [("block", False, True), ("cyclic", True, True), ("block", True, False)])
def test_two_rank_row_sharded_table_step_equals_the_full_batch_oracle_step(hip, partition, graph, train_embedding):
    """configs[4]: table rows split over two ranks, device-planned lookups, equal-split all-to-alls between the captured
    kernel segments, row gradients routed to their owners (never all-reduced)."""
    _spawn(_dp_worker, 2, True, partition, graph, train_embedding, 20)


def _overlap_worker(rank, world, H, train_embedding, graph, exchange):
    """The step with the dense-gradient bucket started under the rest of the backward (trainable table) against the step that waits for each
    collective where it is issued: same kernels, the same flat bucket -> bit-identical losses and replicas."""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(history_size=H, dropout=0.2, learning_rate=1e-3)
    V, D, seed, B, C = 600, 64, 5, 3, 5
    rng = np.random.default_rng(91)
    emb = rng.standard_normal((V, D)).astype(np.float32)
    ms = []
    for overlap in (True, False):
        m = NRMSModel(hp, word2vec_embedding=emb, seed=seed, train_embedding=train_embedding, table_grad_exchange=exchange)
        m._engine.overlap_collectives = overlap
        if graph:
            m._engine.enable_graphs()
        ms.append(m)
    kinds = [k for k, _ in ms[0]._engine._segments(B, C)]
    # a trainable table has work to hide the buckets under (dX GEMM + table-gradient accumulation); a frozen one does not
    assert (kinds.count("a") == 1 and "w" in kinds) if train_embedding else kinds.count("a") == 0
    assert [k for k, _ in ms[1]._engine._segments(B, C)].count("a") == 0
    for t in range(3):
        his, pred, y = batch(np.random.default_rng(1000 * t + rank), B, H, C, hp.title_size, V)
        l0, l1 = (float(m.train_step(his, pred, y).item()) for m in ms)
        assert l0 == l1, (rank, t, l0, l1)
    for a, b in zip(ms[0].model.get_weights(), ms[1].model.get_weights()):
        assert np.array_equal(a, b)
    import torch.distributed as dist

    mine = [w.tobytes() for w in ms[0].model.get_weights()]
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    assert all(p == parts[0] for p in parts)  # identical replicas on every rank


@pytest.mark.parametrize("world,H,train_embedding,graph,exchange", [(2, 20, False, True, "auto"), (2, 20, True, True, "dense"), (2, 50, True, False, "sparse"),
                                                                    (8, 20, True, True, "dense")])
def test_overlapped_gradient_buckets_are_bit_identical_to_the_serial_step(hip, world, H, train_embedding, graph, exchange):
    _spawn(_overlap_worker, world, H, train_embedding, graph, exchange)


# ---------------------------------------------------------------- world = 8 on the one GPU (BASELINE configs[3] / configs[4] rank count)
@pytest.mark.parametrize("table_grad_exchange", ["dense", "sparse"])
def test_eight_rank_c4_data_parallel_step_equals_the_full_batch_oracle_step(hip, table_grad_exchange):
    """configs[3] with its real rank count: 8 processes over gloo on cuda:0, history_size = 50, trainable table, hipGraph
    segments on; the dense all-reduce of the table gradient and the sparse (id, gradient row) all-gather with 8 slabs.
    Every rank must end on the float64 oracle's weights after 3 steps on the GLOBAL batch (reduced V)."""
    _spawn(_dp_worker, 8, False, "block", True, True, 50, table_grad_exchange)


@pytest.mark.parametrize("partition,train_embedding", [("block", True), ("cyclic", True), ("cyclic", False)])
def test_eight_rank_c5_row_sharded_table_step_equals_the_full_batch_oracle_step(hip, partition, train_embedding):
    """configs[4] with its real rank count: the table's rows split 8 ways (block and cyclic ownership; V = 600 is not a
    multiple of 8 for neither), capacity() at W = 8, the 8-way equal-split all-to-all buffer offsets inside captured graph
    segments, row gradients routed to 8 owners."""
    _spawn(_dp_worker, 8, True, partition, True, train_embedding, 20)


def _fit_worker(rank, world, shard, tmpdir, table_grad_exchange="dense"):
    """model.fit under two ranks whose shards have DIFFERENT numbers of batches (and ragged last batches): same number of
    steps everywhere, all-reduced epoch logs -> identical callback decisions, identical replicas, one checkpoint file."""
    import torch.distributed as dist
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.callbacks import EarlyStopping, ModelCheckpoint, ReduceLROnPlateau

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    V, D = 300, 32
    rng = np.random.default_rng(5)
    emb = rng.standard_normal((V, D)).astype(np.float32)
    m = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=True, shard_table=shard, shard_partition="cyclic",
                  shard_capacity_factor=float(world), deterministic=not shard, table_grad_exchange=table_grad_exchange)
    m._engine.enable_graphs()
    equal = shard or table_grad_exchange != "dense"  # collectives sized by the batch shape: only full batches may run
    assert m._engine.needs_equal_batches == equal
    n_rows = 70 if rank == 0 else 50  # batch 16: 5 batches (last one 6 rows) vs 4 batches (last one 2 rows)
    r2 = np.random.default_rng(100 + rank)
    his, pred, y = batch(r2, n_rows, hp.history_size, 5, hp.title_size, V)
    vhis, vpred, vy = batch(r2, 20 + 4 * rank, hp.history_size, 5, hp.title_size, V)
    ckpt = os.path.join(tmpdir, "weights")
    m.model.compile(optimizer=m.model.optimizer, loss=m.model.loss, metrics=["AUC"])
    cbs = [EarlyStopping(monitor="val_auc", mode="max", patience=1, restore_best_weights=True),
           ModelCheckpoint(filepath=ckpt, monitor="val_auc", mode="max", save_best_only=True, save_weights_only=True),
           ReduceLROnPlateau(monitor="val_auc", mode="max", factor=0.2, patience=1, min_lr=1e-6)]
    h = m.model.fit((his, pred), y.astype(np.float32), batch_size=16, epochs=3, verbose=0, callbacks=cbs,
                    validation_data=((vhis, vpred), vy.astype(np.float32)))
    # "auto" on a 300-row table picks the DENSE all-reduce (pinned by fit() from the full-batch shape): every batch trains, like the
    # reference; only a step whose collectives are sized by the batch shape is restricted to full batches
    equal_run = shard or table_grad_exchange == "sparse"
    assert m._engine.needs_equal_batches == equal_run
    per_epoch = 3 if equal_run else 4
    assert m._engine.read_state().step == per_epoch * len(h.history["loss"])
    hist = {k: [float(v) for v in vs] for k, vs in h.history.items()}
    parts = [None] * world
    dist.all_gather_object(parts, (hist, [w.tobytes() for w in m.model.get_weights()], float(m.model.optimizer.learning_rate)))
    assert parts[0][0] == parts[1][0], (parts[0][0], parts[1][0])  # identical epoch logs on both ranks
    assert parts[0][1] == parts[1][1]  # identical replicas (incl. the gathered table)
    assert parts[0][2] == parts[1][2]
    assert os.path.exists(ckpt)
    m.model.load_weights(ckpt)  # every rank can read rank 0's checkpoint


@pytest.mark.parametrize("shard,table_grad_exchange", [(False, "dense"), (True, "dense"), (False, "sparse"), (False, "auto")])
def test_two_rank_fit_keeps_ranks_in_lock_step(hip, shard, table_grad_exchange, tmp_path):
    """Shards of UNEQUAL length (5 vs 4 batches, ragged last ones).  With the sparse table-gradient exchange the all-gathers
    are sized by the local batch shape: a short last batch on one rank next to a full one on the other would hang or corrupt
    memory (round-2 ADVICE, high) -- fit() runs only full batches then."""
    _spawn(_fit_worker, 2, shard, str(tmp_path), table_grad_exchange)


def _one_rank_inference_worker(rank, world, tmpdir):
    """Round-3 ADVICE (high): with a REPLICATED table the inference entry points are rank-local -- the reproducibility driver runs
    `scorer.predict` on rank 0 while the other ranks have already returned (examples/reproducibility_scripts/ebnerd_nrms.py) -- so
    they must not contain a collective.  The collective APIs (save_weights) entered by one rank raise instead of hanging."""
    import time

    import torch.distributed as dist
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    V, D = 300, 32
    rng = np.random.default_rng(5)
    emb = rng.standard_normal((V, D)).astype(np.float32)
    m = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=True)
    assert m._engine.world == world and m._engine.guard is not None
    his, pred, y = batch(rng, 6, hp.history_size, 5, hp.title_size, V)
    m.train_step(his, pred, y)  # one lock-step training step (gradient all-reduce) on both ranks
    if rank == 0:
        probs = m.model.predict((his, pred))                      # forward() -> local flag read, no all-reduce
        sc = m.scorer.predict((his, pred[:, :1]))                 # encode_news / encode_users -> the same
        news = m.newsencoder.predict(his[0])
        assert probs.shape == (6, 5) and sc.shape == (6, 1) and news.shape == (hp.history_size, m._engine.E)
        bad = his.copy()
        bad[0, 0, 0] = V  # an id out of range on ONE rank still raises there, locally
        with pytest.raises(IndexError):
            m._engine.forward(torch.from_numpy(bad.astype(np.int32)).cuda(), torch.from_numpy(pred.astype(np.int32)).cuda())
        m._engine.guard.timeout_s = 2.0
        with pytest.raises(RuntimeError, match="COLLECTIVE"):
            m.model.save_weights(os.path.join(tmpdir, "w"))      # rank 0 only: an error after the timeout, not a hang
    else:
        time.sleep(6.0)
    dist.barrier()


def test_rank_local_inference_on_a_replicated_table_has_no_collective_and_lone_collective_calls_raise(hip, tmp_path):
    _spawn(_one_rank_inference_worker, 2, str(tmp_path))


def _c5_full_width_worker(rank, world, train_embedding):
    """configs[4] at its own table size on TWO ranks: 250002 x 1024 rows split cyclically, B = 32 per rank (global 64: the per-rank
    batch of c5), device-planned lookups with the default capacity, the equal-split all-to-all buffers at 4 KB rows
    (2 x 15 040 slots x 4 KB each way), hipGraph segments on; against the float64 oracle stepping on the GLOBAL batch with the
    compacted vocabulary."""
    from ebrec.models.newsrec import NRMSModel
    from tests.test_full_size_parity import _check_dense_grads, _dense_weights

    V, D, B, C, H, seed, lr = 250002, 1024, 32, 5, 20, 19, 1e-3
    hp = make_hp(history_size=H, dropout=0.0, learning_rate=lr)
    rng = np.random.default_rng(4242)  # identical streams on both ranks
    table = rng.standard_normal((V, D), dtype=np.float32) * np.float32(0.05)
    P = on.random_nrms_params(1, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim, seed=29)
    m = NRMSModel(hp, word2vec_embedding=table, seed=seed, train_embedding=train_embedding, shard_table=True, deterministic=False)
    m.from_keras_weight_list([table] + weight_list(P)[1:])
    eng = m._engine
    assert eng.world == world and eng.exchange.partition == "cyclic" and eng.table.shape[0] == len(range(rank, V, world))
    cap = eng.exchange.capacity(B * (H + C) * hp.title_size)
    assert cap < B * (H + C) * hp.title_size  # the DEFAULT capacity factor (1.25 x n_tok / W), not the can-never-overflow one
    eng.enable_graphs()
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    P0 = {k: v.copy() for k, v in P.items()}
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in P.items()}
    for t in range(1, 3):
        his, pred, y = batch(rng, B * world, H, C, hp.title_size, V)  # the GLOBAL batch
        uniq, inv = np.unique(np.concatenate([his.reshape(-1), pred.reshape(-1)]), return_inverse=True)
        P["emb"] = table[uniq].astype(np.float64)
        his_c, pred_c = inv[: his.size].reshape(his.shape), inv[his.size:].reshape(pred.shape)
        need = train_embedding and t == 1
        L, _, g = on.nrms_loss_and_grads(his_c, pred_c, y, P, hp.head_num, hp.head_dim, "cross_entropy_loss", None, need_emb_grad=need)
        sl = slice(rank * B, (rank + 1) * B)
        probs_loc = on.nrms_forward(his_c[sl], pred_c[sl], P, hp.head_num, hp.head_dim)[0]
        got = float(eng.train_step(his[sl], pred[sl], y[sl]).item())  # each rank steps on ITS rows
        want_loc = float(-np.log(probs_loc[y[sl].astype(bool)]).mean())
        assert abs(got - want_loc) <= 2e-5 * max(1.0, abs(want_loc)), (rank, t, got, want_loc)
        if t == 1:  # the all-reduced (SUM) dense gradients = world x the full-batch mean gradient
            _check_dense_grads(eng, {k: v * world for k, v in g.items() if k != "emb"})
            if train_embedding:  # my shard's gradient rows: the sum over BOTH ranks' requests, routed to me (never all-reduced)
                mine = uniq[uniq % world == rank]
                got_g = eng.table_grad[torch.from_numpy(mine // world).to(eng.device)].cpu().numpy()
                want_g = g["emb"][uniq % world == rank] * world
                assert_close(got_g, want_g, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(want_g).max(), what=f"rank {rank}: d(shard rows)")
        if train_embedding:
            break  # (a second float64 step would need the moved rows of a 1 GB table: the dense weights + shard gradient pin the path)
        for k in on.PARAM_ORDER[1:]:
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=lr)
    eng.check_oob()  # the default capacity held (uniform ids: ~11 000 distinct rows per owner against 15 040 slots)
    if not train_embedding:
        got = _dense_weights(eng)
        for k in on.PARAM_ORDER[1:]:
            step = np.abs(P[k] - P0[k])
            assert_close(got[k].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"rank {rank}: c5 weights {k}")
    st = eng.exchange.stats()["bytes_sent_per_lookup_remote"]
    assert st["rows"] == (world - 1) * cap * D * 4


@pytest.mark.parametrize("train_embedding", [False, True])
def test_two_rank_c5_full_width_row_sharded_step_equals_the_full_batch_oracle_step(hip, train_embedding):
    _spawn(_c5_full_width_worker, 2, train_embedding)


def _graph_collectives_check_worker(rank, world):
    """verify_graph_collectives under gloo: host-side collectives cannot be captured into a hipGraph (the check does not even try:
    a failed capture leaves the stream unusable), so it must come back False on every rank, leave weights / moments / step counter
    / dropout keys exactly as they were, and training must go on in the segment form -- bit-identical to a twin engine that never
    ran the check.  (The accepting side of the check runs on RCCL: tests/test_rccl_single_rank_gpu.py.)"""
    from ebrec.models.newsrec import NRMSModel

    hp = make_hp(dropout=0.2, learning_rate=1e-3)
    V, D, B, C = 300, 32, 3, 5
    emb = np.random.default_rng(5).standard_normal((V, D)).astype(np.float32)
    a = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=True, table_grad_exchange="dense")
    b = NRMSModel(hp, word2vec_embedding=emb, seed=3, train_embedding=True, table_grad_exchange="dense")
    a._engine.enable_graphs()
    b._engine.enable_graphs()
    rng = np.random.default_rng(100 + rank)
    his, pred, y = batch(rng, B, hp.history_size, C, hp.title_size, V)
    a.train_step(his, pred, y)
    b.train_step(his, pred, y)
    before = [t.clone() for t in a._engine._state_tensors()]
    assert a._engine.verify_graph_collectives(his, pred, y) is False and not a._engine.graph_collectives
    assert all(torch.equal(x, z) for x, z in zip(before, a._engine._state_tensors()))
    for t in range(2):
        his, pred, y = batch(rng, B, hp.history_size, C, hp.title_size, V)
        la, lb = float(a.train_step(his, pred, y).item()), float(b.train_step(his, pred, y).item())
        assert la == lb, (rank, t, la, lb)
    for wa, wb in zip(a.model.get_weights(), b.model.get_weights()):
        assert np.array_equal(wa, wb)


def test_graph_collectives_self_check_falls_back_cleanly_where_collectives_cannot_be_captured(hip):
    _spawn(_graph_collectives_check_worker, 2)


# ---------------------------------------------------------------- data-parallel NRMSDocVec / MLP-branch NRMS (BatchNormalization)
def _docvec_dp_worker(rank, world, graph, p, tmpdir):
    """NRMSDocVec under data parallel (nrms_docvec.py:116-124,139-188; verdict r4 item 2).  The contract: every rank normalises ITS rows
    with ITS batch statistics per call site (non-synchronised BatchNormalization, as tf.keras'), all gradients -- Dense, gamma / beta,
    user encoder -- are the MEAN of the ranks' gradients, the moving statistics drift per rank and are averaged over the ranks once
    per epoch / in front of evaluate / save_weights.  Oracle: the float64 step on each rank's rows, averaged."""
    import torch.distributed as dist
    from ebrec.models.newsrec import NRMSDocVec
    from tests import test_docvec_model as td

    hp = td.make_hp(title_size=64, newsencoder_units_per_layer=[48, 40], head_num=4, head_dim=8, attention_hidden_dim=12, history_size=7,
                    dropout=p, newsencoder_l2_regularization=1e-4, learning_rate=1e-3)
    seed, B, C, l2, lr = 5, 6, 5, 1e-4, 1e-3
    P = td.oracle_params(hp, 9)
    m = NRMSDocVec(hp, seed=seed)
    m.model.set_weights(td.weight_list(P))
    eng = m._engine
    assert eng.world == world
    if graph:
        eng.enable_graphs()
    train_keys = [k for k, v in P.items() if isinstance(v, np.ndarray) and not k.endswith(("_mean", "_var"))]
    P0 = {k: P[k].copy() for k in train_keys}
    mom = {k: (np.zeros_like(P[k]), np.zeros_like(P[k])) for k in train_keys}
    moving = [dict(P) for _ in range(world)]  # every rank's OWN moving statistics (float64 oracle)
    rng = np.random.default_rng(21)
    for t in range(1, 4):
        his, pred, y = td.data(rng, B * world, hp.history_size, C, hp.title_size)  # the GLOBAL batch, identical on every rank
        per_rank = []
        for r in range(world):
            sl = slice(r * B, (r + 1) * B)
            Pr = dict(P, **{k: moving[r][k] for k in moving[r] if k.endswith(("_mean", "_var"))})
            L, _, g, stats = on.docvec_loss_and_grads(his[sl].astype(np.float64), pred[sl].astype(np.float64), y[sl], Pr, hp.head_num, hp.head_dim,
                                                      l2=l2, training=True, drop=on.Drop(p, seed, t) if p > 0 else None)
            on.bn_update_moving(moving[r], stats)
            per_rank.append((L, g))
        sl = slice(rank * B, (rank + 1) * B)
        got = float(eng.train_step(his[sl], pred[sl], y[sl]).item())
        assert abs(got - per_rank[rank][0]) <= 3e-5 * max(1.0, abs(per_rank[rank][0])), (rank, t, got, per_rank[rank][0])
        g = {k: sum(pr[1][k] for pr in per_rank) / world for k in per_rank[0][1]}
        if t == 1:  # after the all-reduce (SUM) the gradient buffer holds world x the mean of the ranks' gradients
            for k in ("d0_W", "d1_W", "d1_b", "bn0_g", "bn1_b", "out_W", "out_b", "u_W", "u_q"):
                want = g[k].reshape(eng.params.shapes[k]) * world
                assert_close(eng.params.g(k).cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what=f"rank {rank}: all-reduced d{k}")
        for k in train_keys:
            on.adam_keras_step(P[k], g[k], mom[k][0], mom[k][1], t, lr=lr)
    # moving statistics: this rank's own so far ...
    for l in range(2):
        assert_close(eng.bn_mean[l].cpu().numpy(), moving[rank][f"bn{l}_mean"], rtol=1e-5, atol=1e-6, what=f"rank {rank}: own moving mean {l}")
        assert_close(eng.bn_var[l].cpu().numpy(), moving[rank][f"bn{l}_var"], rtol=1e-5, atol=1e-6, what=f"rank {rank}: own moving var {l}")
    assert not np.allclose(moving[0]["bn0_mean"], moving[1]["bn0_mean"], atol=1e-6)  # (they HAVE drifted apart)
    # ... then the mean over the ranks, the same bits on every rank
    eng.sync_moving_statistics()
    for l in range(2):
        for nm, got in (("mean", eng.bn_mean[l]), ("var", eng.bn_var[l])):
            want = sum(moving[r][f"bn{l}_{nm}"] for r in range(world)) / world
            assert_close(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6, what=f"rank {rank}: synchronised moving {nm} {l}")
    flat = eng.mlp.bn_stats.clone()
    everyone = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(everyone, flat)
    assert all(torch.equal(e, everyone[0]) for e in everyone)
    got = dict(zip([n for n in eng.weight_names()], m.model.get_weights()))
    name_of = {"d0_W": "news.dense0.kernel", "d1_W": "news.dense1.kernel", "bn0_g": "news.bn0.gamma", "bn1_b": "news.bn1.beta", "out_W": "news.out.kernel",
               "u_W": "user.att.W", "u_b": "user.att.b"}
    for k, nm in name_of.items():
        step = np.abs(P[k] - P0[k])
        assert_close(got[nm].reshape(P[k].shape), P[k], rtol=0, atol=2e-5 + 0.02 * float(step.max()), what=f"rank {rank}: {k} after 3 DP steps")
    # fit(): ranks train on DIFFERENT shards; afterwards weights, moving statistics and scores agree on every rank, and rank 0's
    # checkpoint (a collective save) reloads to the same scores
    rs = np.random.default_rng(100 + rank)
    his, pred, y = td.data(rs, 4 * B, hp.history_size, C, hp.title_size)
    vh, vp, vy = td.data(np.random.default_rng(7), 2 * B, hp.history_size, C, hp.title_size)
    m.model.compile(optimizer=m.model.optimizer, loss=m.model.loss, metrics=["AUC"])
    h = m.model.fit((his, pred), y, batch_size=B, epochs=2, verbose=0, validation_data=((vh, vp), vy))
    for w in m.model.get_weights():
        t_ = torch.from_numpy(np.ascontiguousarray(w)).cuda()
        ws = [torch.empty_like(t_) for _ in range(world)]
        dist.all_gather(ws, t_)
        assert all(torch.equal(a, ws[0]) for a in ws)
    logs = torch.tensor([h.history["val_loss"][-1], h.history["val_auc"][-1]], dtype=torch.float64)
    ls = [torch.empty_like(logs) for _ in range(world)]
    dist.all_gather(ls, logs)
    assert all(torch.equal(a, ls[0]) for a in ls)
    path = os.path.join(tmpdir, "docvec_dp.weights")
    m.model.save_weights(path)
    before = m.model.predict((vh, vp))
    m.model.load_weights(path)
    assert np.array_equal(m.model.predict((vh, vp)), before)


@pytest.mark.parametrize("graph,p", [(False, 0.0), (True, 0.2)])
def test_two_rank_docvec_step_is_the_mean_of_the_per_rank_oracle_steps_and_moving_statistics_agree_after_fit(hip, graph, p, tmp_path):
    _spawn(_docvec_dp_worker, 2, graph, p, str(tmp_path))


def _nrms_mlp_dp_worker(rank, world):
    """The NRMS news encoder's optional per-token Dense / BatchNormalization stack (nrms.py:143-152) under data parallel: same contract."""
    import torch.distributed as dist
    from ebrec.models.newsrec import NRMSModel
    from tests.test_nrms_model import _mlp_weight_list

    units = [48, 32]
    hp = make_hp(head_num=4, head_dim=8, attention_hidden_dim=10, history_size=5, title_size=6, dropout=0.0, learning_rate=1e-3,
                 newsencoder_units_per_layer=units, newsencoder_l2_regularization=1e-4)
    V, D, seed, B, C = 80, 20, 6, 4, 4
    P = on.random_nrms_params(V, D, 4, 8, 10, seed=3)
    on.add_mlp_params(P, units, 32, 10, seed=4)
    P = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in P.items()}
    m = NRMSModel(hp, word2vec_embedding=P["emb"], seed=seed, table_grad_exchange="dense")
    m.model.set_weights(_mlp_weight_list(P, units))
    eng = m._engine
    rng = np.random.default_rng(71)
    his, pred, y = batch(rng, B * world, 5, C, 6, V)
    per_rank, moving = [], []
    for r in range(world):
        sl = slice(r * B, (r + 1) * B)
        L, _, g, stats = on.nrms_mlp_loss_and_grads(his[sl], pred[sl], y[sl], P, 4, 8, l2=1e-4, training=True, drop=None)
        Pn = dict(P)
        on.bn_update_moving(Pn, stats, prefix="n_")
        per_rank.append((L, g))
        moving.append(Pn)
    sl = slice(rank * B, (rank + 1) * B)
    got = float(m.train_step(his[sl], pred[sl], y[sl]).item())
    assert abs(got - per_rank[rank][0]) <= 3e-5 * max(1.0, abs(per_rank[rank][0]))
    for k in ("n_d0_W", "n_bn0_g", "n_bn1_b", "n_d1_W", "n_W", "u_W"):
        want = sum(pr[1][k] for pr in per_rank).reshape(eng.params.shapes[k])
        assert_close(eng.params.g(k).cpu().numpy(), want, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(want).max(), what=f"rank {rank}: all-reduced d{k}")
    for l in range(2):
        assert_close(eng.mlp.bn_mean[l].cpu().numpy(), moving[rank][f"n_bn{l}_mean"], rtol=1e-5, atol=1e-6, what=f"own moving mean {l}")
    eng.sync_moving_statistics()
    for l in range(2):
        want = sum(mv[f"n_bn{l}_var"] for mv in moving) / world
        assert_close(eng.mlp.bn_var[l].cpu().numpy(), want, rtol=1e-5, atol=1e-6, what=f"synchronised moving var {l}")
    flat = eng.mlp.bn_stats.clone()
    everyone = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(everyone, flat)
    assert all(torch.equal(e, everyone[0]) for e in everyone)


def test_two_rank_nrms_mlp_branch_batchnorm_follows_the_docvec_contract(hip):
    _spawn(_nrms_mlp_dp_worker, 2)
