"""Device engine of the MI355X NRMS path: owns parameters, activations and optimizer state in
HBM and drives the HIP kernels through the C ABI (include/ebnerd_hip.h).

torch is plumbing only here (device memory, streams, torch.distributed over RCCL); every
arithmetic op of the reference hot path -- SURVEY.md section 8(a) rows a1..a10 -- is a call
into libebnerd_hip.so.  There is no CPU path: constructing an engine without the library or
without a GPU raises.

Data layout in HBM (all fp32 row-major):
  table        (V, D)            word-embedding rows (a1)
  dense        flat buffer       n_Wqkv (D,3E) | n_W (E,A) | n_b (A) | n_q (A) |
                                 u_Wqkv (E,3E) | u_W (E,A) | u_b (A) | u_q (A)
                                 -> one RCCL all-reduce and one Adam launch per step
  ids          (N*T,) int32      N = B*(H+C) titles: history titles first, then candidates
  Xd/QKV/Y/U/w per-token activations of the news encoder (kept for backward)
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from ebrec import _hip

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-7  # tf.keras.optimizers.Adam defaults (nrms.py:77)
_ALIGN = 64  # floats; keeps every parameter 256-byte aligned inside the flat buffer

from ._engine_segments import SegmentsMixin  # noqa: E402  (the multi-rank step: segments, collectives, the one-graph self-check)
from ._engine_staging import StagingMixin  # noqa: E402  (device-side batch assembly: article rows -> token ids, pinned staging)

LOSS_KIND = {"cross_entropy_loss": 0, "log_loss": 1}
BCE_ON = ("logits", "probs")


def loss_kind_of(loss: str, bce_on: str) -> int:
    """C-ABI loss_kind: 0 categorical CE on the softmax logits; 1 log_loss as sigmoid-CE on the logits Keras caches on the
    output of Activation("softmax") (bce_on="logits", the default); 2 log_loss as BCE on the clipped softmax outputs
    (bce_on="probs", SURVEY.md A.5's reading).  Which of the two TF 2.12-2.15 runs at nrms.py:54,61-62 is settled by the TF
    dump (tools/dump_tf_golden.py); until then both are implemented and tested, and switching is this one argument."""
    if bce_on not in BCE_ON:
        raise ValueError(f"bce_on must be one of {BCE_ON}, got {bce_on}")
    return 0 if loss == "cross_entropy_loss" else (1 if bce_on == "logits" else 2)


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("the MI355X NRMS path needs a visible GPU (no CPU fallback); "
                           "torch.cuda.is_available() is False")
    _hip.lib()
    return torch.device("cuda", torch.cuda.current_device())


def eval_loss_from_news(eng, news_all: torch.Tensor, his_idx, pred_idx, y):
    """Validation loss of one batch from CACHED news vectors (both engines): his_idx (B,H) / pred_idx (B,C) are rows of
    news_all (n_articles+1, E).  Same kernels as eval_loss after the news encoder: user encoder, scorer, compiled loss.
    Returns (loss[1], probs (B,C)) device tensors."""
    his_idx, pred_idx = np.asarray(his_idx), np.asarray(pred_idx)
    B, C = his_idx.shape[0], pred_idx.shape[1]
    H, E = his_idx.shape[1], eng.E
    idx = torch.from_numpy(np.ascontiguousarray(np.concatenate([his_idx.reshape(-1), pred_idx.reshape(-1)]), dtype=np.int32)).to(eng.device)
    NE = torch.empty(B * (H + C), E, device=eng.device)
    S = _hip.stream_handle
    _hip.call("ebn_gather_rows_f32", _hip.ptr(idx), _hip.ptr(news_all), _hip.ptr(NE), B * (H + C), E, news_all.shape[0], None, -1,
              ctypes.c_float(0.0), None, S())
    user = eng.encode_users_from_news(NE[: B * H].view(B, H, E))
    cand = NE[B * H:]
    scores, probs = torch.empty(B, C, device=eng.device), torch.empty(B, C, device=eng.device)
    _hip.call("ebn_score_fwd_f32", _hip.ptr(cand), _hip.ptr(user), _hip.ptr(scores), _hip.ptr(probs), B, C, E, 0, S())
    labels = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32))))
    labels = labels.to(device=eng.device, dtype=torch.float32).reshape(B, C).contiguous()
    rows, junk_c, junk_u = torch.empty(B, device=eng.device), torch.empty(B * C, E, device=eng.device), torch.empty(B, E, device=eng.device)
    loss = torch.empty(1, device=eng.device)
    _hip.call("ebn_score_loss_bwd_f32", _hip.ptr(cand), _hip.ptr(user), _hip.ptr(scores), _hip.ptr(labels), _hip.ptr(rows),
              _hip.ptr(junk_c), _hip.ptr(junk_u), B, C, E, eng.loss_kind, ctypes.c_float(1.0 / B), S())
    _hip.call("ebn_sum_f32", _hip.ptr(rows), B, ctypes.c_float(1.0), _hip.ptr(loss), 0, S())
    return loss, probs


def glorot_uniform_np(shape, seed):
    """[KERAS-SEMANTICS] GlorotUniform(seed): the same (seed, shape) gives the same draw, which is
    why WQ, WK and WV of one layer start identical in the reference (layers.py:155-172)."""
    rng = np.random.default_rng(seed)
    lim = math.sqrt(6.0 / (shape[0] + shape[-1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


class FlatParams:
    """Named fp32 parameters carved out of one flat device buffer (plus grad / Adam moments)."""

    def __init__(self, shapes: dict, device):
        self.shapes = dict(shapes)
        self.offsets = {}
        off = 0
        for name, shp in shapes.items():
            self.offsets[name] = off
            off += -(-int(np.prod(shp)) // _ALIGN) * _ALIGN
        self.numel = off
        self.data = torch.zeros(off, device=device)
        self.grad = torch.zeros(off, device=device)
        self.m = torch.zeros(off, device=device)
        self.v = torch.zeros(off, device=device)

    def view(self, name, buf=None):
        buf = self.data if buf is None else buf
        n = int(np.prod(self.shapes[name]))
        return buf[self.offsets[name]: self.offsets[name] + n].view(*self.shapes[name])

    def g(self, name):
        return self.view(name, self.grad)


class EncoderBuffers:
    """Activations + backward scratch of one SelfAttention+AttLayer2 stage for up to n_seq sequences."""

    def __init__(self, n_seq, L, Din, E, A, device, own_input, need_dx):
        R = n_seq * L
        self.n_seq, self.L, self.Din, self.E, self.A, self.R = n_seq, L, Din, E, A, R
        f = lambda *s: torch.empty(*s, device=device)
        self.X = f(R, Din) if own_input else None
        self.QKV, self.Y, self.U, self.w = f(R, 3 * E), f(R, E), f(R, A), f(R)
        self.out = f(n_seq, E)
        self.dY, self.dQKV, self.de = f(R, E), f(R, 3 * E), f(R)
        self.partials = f(int(_hip.lib().ebn_attpool_partials_len(R, A)))
        wsf = _hip.lib().ebn_gemm_workspace_floats  # split-K scratch for every GEMM shape of the stage (fwd and bwd)
        ws = max(int(wsf(Din, 3 * E, R)), int(wsf(E, A, R)), int(wsf(R, 3 * E, Din)), int(wsf(R, A, E)), int(wsf(R, E, A)),
                 int(wsf(R, Din, 3 * E)), 1)
        self.ws = f(ws)
        self.fwd_scratch = _hip.EncoderScratch(None, None, None, None, self.ws.data_ptr(), self.ws.numel())
        self.dX = f(R, Din) if need_dx else None


class NRMSEngine(StagingMixin, SegmentsMixin):
    def __init__(self, table: np.ndarray, title_size: int, history_size: int, head_num: int, head_dim: int,
                 attention_hidden_dim: int, dropout: float, learning_rate: float, loss: str, seed=None,
                 train_embedding: bool = True, device=None, process_group=None, shard_table: bool = False,
                 shard_mode: str = "alltoall", deterministic: bool = True, units=None, l2: float = 0.0,
                 shard_partition: str | None = None, shard_capacity_factor: float = 1.25, table_grad_exchange: str = "auto",
                 bce_on: str = "logits", precision: str = "exact"):
        self.device = require_gpu() if device is None else torch.device(device)
        if precision not in ("exact", "split"):
            raise ValueError(f"precision must be 'exact' (fp32 MFMA, the default) or 'split' (bf16x6 split on the bf16 matrix pipe, "
                             f"fp32-accurate, fp32 accumulate), got {precision}")
        self.precision = precision
        if loss not in LOSS_KIND:
            raise ValueError(f"this loss not defined {loss}")
        loss_kind_of(loss, bce_on)
        self.bce_on = bce_on
        self.T, self.H = int(title_size), int(history_size)
        self.h, self.d, self.A = int(head_num), int(head_dim), int(attention_hidden_dim)
        self.E = self.h * self.d
        self.p = float(dropout)
        self.loss = loss
        self.seed = seed
        self.train_embedding = bool(train_embedding)
        if table_grad_exchange not in ("auto", "dense", "sparse"):
            raise ValueError(f"table_grad_exchange must be auto | dense | sparse, got {table_grad_exchange}")
        self.table_grad_exchange = table_grad_exchange
        self._sparse_pin = None  # the "auto" decision, taken once (pin_table_grad_exchange)
        self.pg = process_group
        table = np.ascontiguousarray(table, dtype=np.float32)  # copied, like weights=[...] (nrms.py:128)
        self.V, self.D = table.shape
        self.exchange = None
        if shard_table:
            # BASELINE config 5: each rank keeps only its contiguous block of rows in HBM
            from ._dist import ShardedTableExchange

            self.exchange = ShardedTableExchange(self.V, self.D, group=process_group, mode=shard_mode, partition=shard_partition,
                                                 capacity_factor=shard_capacity_factor)
            table = self.exchange.shard_of(table)
        self.table = torch.from_numpy(np.ascontiguousarray(table)).to(self.device)
        D, E, A = self.D, self.E, self.A
        # optional per-token [Dense-ReLU -> BatchNorm -> Dropout] stack between self-attention and AttLayer2 (nrms.py:142-152)
        self.units = [int(u) for u in units] if units else []
        if self.units and self.precision == "split":
            raise ValueError("precision='split' covers the news encoder without the optional per-token Dense stack "
                             "(newsencoder_units_per_layer); use precision='exact' with it")
        if self.units and self.units[-1] != E:
            raise ValueError(f"newsencoder_units_per_layer must end with head_num*head_dim = {E} (the news vector is dotted with "
                             f"the {E}-wide user vector, nrms.py:201), got {self.units}")
        shapes = {"n_Wqkv": (D, 3 * E)}
        if self.units:
            from ._mlp import MLPStack

            shapes.update(MLPStack.shapes("n_", E, self.units))
        shapes.update({"n_W": (E, A), "n_b": (A,), "n_q": (A,), "u_Wqkv": (E, 3 * E), "u_W": (E, A), "u_b": (A,), "u_q": (A,)})
        self.params = FlatParams(shapes, self.device)
        self._graphs = {}
        self.mlp = MLPStack(self.params, "n_", E, self.units, self.device, l2, on_realloc=lambda: self._graphs.clear()) if self.units else None
        self._init_weights(seed)
        self.deterministic = bool(deterministic)
        if self.train_embedding:
            self.table_grad = torch.zeros_like(self.table)
            self.table_m = torch.zeros_like(self.table)
            self.table_v = torch.zeros_like(self.table)
            if self.deterministic:  # order-independent fixed-point accumulator of the embedding gradient
                self.table_acc = torch.zeros(self.table.shape, dtype=torch.int64, device=self.device)
        st = _hip.StepState()
        st.step, st.seed, st.lr, st.adam_alpha = 0, (0 if seed is None else int(seed)) & 0xFFFFFFFF, learning_rate, 0.0
        self.state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(self.device)
        self._lr = float(learning_rate)
        self._bufs = {}
        self.oob_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.keep_table_grad = False  # True: always materialise the fp32 dense table gradient (inspection / tests)
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=self.device)  # fixed-point gradient accumulator left its range
        self.loss_dev = torch.zeros(1, device=self.device)
        self.use_graph = False  # capture the per-shape kernel sequence into hipGraphs (enable_graphs())
        self.fuse_user_head = True  # False: the per-impression head of a step as its six separate launches (validation)
        self.fuse_eval_gather = True  # False: inference gathers the token rows into X first, like a training step (validation / A-B)
        # True: the four small finishing passes of the backward (split-K sums of dW and dWqkv, AttLayer2 d(q) / d(b) column sums, the
        # per-impression head's d(q) / d(b) / loss sums) run as ONE launch at the end of the backward (ebn_grad_finish_f32) instead
        # of four launches of the dependent chain; same summation orders, same bits (False: the stand-alone passes -- validation)
        self.defer_finish = True
        self.atomic_table_grad = False  # True: one 64-bit atomic per gradient element instead of combining the duplicates of every 64
        # consecutive tokens first (same bits; the validation / A-B form: 2.8-3.1x slower on Zipf ids, profiles/r04_*zipf*)
        self.graph_collectives = False  # multi-rank: capture the collectives into the step's hipGraph as well (experimental)
        self.overlap_collectives = True  # multi-rank: start the dense-gradient buckets under the rest of the backward (see _segments)
        self.skip_collectives = False    # bench.py only: time the step without its collectives (-> comm_exposed_us); results are wrong
        self._pending = []
        self.defer_flag_checks = False  # set by the lock-step callers around their inference calls (see _check_oob)
        self.trace = None  # a _dist.SegmentTrace: event behind every segment of every step (bench.py's hang watchdog); None = off
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # world > 1: save_weights / evaluate / fit raise instead of hanging when only some ranks call them (nothing is built here: the
        # guard's store rendezvous starts at its first enter(); `guard.status()` says "disabled: ..." when no store is reachable)
        self.guard = None
        if self.world > 1:
            from ._dist import LockStepGuard

            self.guard = LockStepGuard(process_group)
        # True: launch the multi-rank form of the step (collectives between the graph replays, the multi-rank Adam inputs) on a group of
        # ONE rank, where every collective is an identity -- `bench.py --force-dist` runs the RCCL branch end to end on a 1-GPU box
        self.force_collectives = False
        # True (one rank, the standard configuration of `_deferred`): Keras-form Adam on the dense parameters runs INSIDE the step's
        # finishing launch (ebn_grad_finish_adam_f32: each gradient element gets its update from the thread that has just summed it,
        # the user encoder's kernels -- whose gradients earlier launches wrote -- in blocks behind) instead of as a launch of its own.
        # With a trainable table the finishing launch then runs AFTER the input-gradient GEMM (which reads Wqkv).  False: the two
        # launches (what a multi-rank step runs: the all-reduce sits between them; the validation form)
        self.adam_in_finish = True
        self._adam_done_in_finish = False  # set by the step's backward when its finishing launch has applied the dense Adam

    @property
    def multi(self) -> bool:
        """the step runs its multi-rank launch form (world > 1, or forced on a one-rank group: see `force_collectives`)"""
        return self.world > 1 or self.force_collectives

    @property
    def loss_kind(self) -> int:
        return loss_kind_of(self.loss, self.bce_on)

    # ------------------------------------------------------------------ parameters
    def _init_weights(self, seed):
        D, E, A = self.D, self.E, self.A
        s = (lambda: seed) if seed is not None else (lambda: None)
        pv = self.params.view
        with torch.no_grad():
            for pre, din in (("n", D), ("u", E)):
                # three separate Keras initialisers with the same seed and shape -> identical draws
                w = np.concatenate([glorot_uniform_np((din, E), s()) for _ in range(3)], axis=1)
                pv(f"{pre}_Wqkv").copy_(torch.from_numpy(w))
                pv(f"{pre}_W").copy_(torch.from_numpy(glorot_uniform_np((E, A), s())))
                pv(f"{pre}_b").zero_()
                pv(f"{pre}_q").copy_(torch.from_numpy(glorot_uniform_np((A, 1), s())[:, 0]))
        if self.mlp is not None:
            self.mlp.init_weights((lambda k: None) if seed is None else (lambda k: int(seed) * 1000 + k), glorot_uniform_np)

    def weight_names(self):
        base = ["news.emb", "news.attn.WQ", "news.attn.WK", "news.attn.WV"]
        if self.mlp is not None:
            base += self.mlp.weight_names("news")
        return base + ["news.att.W", "news.att.b", "news.att.q", "user.attn.WQ", "user.attn.WK", "user.attn.WV", "user.att.W",
                       "user.att.b", "user.att.q"]

    def get_weights(self):
        """13 arrays in the Keras creation order of SURVEY.md A.6."""
        E = self.E
        out = [self._full_table().cpu().numpy()]
        for pre in ("n", "u"):
            w = self.params.view(f"{pre}_Wqkv").cpu().numpy()
            out += [w[:, :E].copy(), w[:, E:2 * E].copy(), w[:, 2 * E:].copy()]
            if pre == "n" and self.mlp is not None:  # Keras creation order: Dense/BN layers sit before AttLayer2
                out += self.mlp.get_weights()
            out += [self.params.view(f"{pre}_W").cpu().numpy(), self.params.view(f"{pre}_b").cpu().numpy(),
                    self.params.view(f"{pre}_q").cpu().numpy().reshape(-1, 1)]
        return out

    def set_weights(self, weights):
        n_mlp = 6 * len(self.units)
        if len(weights) != 13 + n_mlp:
            raise ValueError(f"expected {13 + n_mlp} weight arrays (emb, 2x[WQ,WK,WV,W,b,q]{' + 6 per Dense/BN layer' if n_mlp else ''}), "
                             f"got {len(weights)}")
        w = [np.asarray(a, dtype=np.float32) for a in weights]
        if n_mlp:
            self.mlp.set_weights([torch.from_numpy(np.ascontiguousarray(a)) for a in w[4:4 + n_mlp]])
            w = w[:4] + w[4 + n_mlp:]
        if w[0].shape != (self.V, self.D):
            raise ValueError(f"embedding shape {w[0].shape} != {(self.V, self.D)}")
        with torch.no_grad():
            t0 = w[0] if self.exchange is None else self.exchange.shard_of(w[0])
            self.table.copy_(torch.from_numpy(np.ascontiguousarray(t0)))
            i = 1
            for pre in ("n", "u"):
                self.params.view(f"{pre}_Wqkv").copy_(torch.from_numpy(np.concatenate(w[i:i + 3], axis=1)))
                self.params.view(f"{pre}_W").copy_(torch.from_numpy(w[i + 3]))
                self.params.view(f"{pre}_b").copy_(torch.from_numpy(w[i + 4].reshape(-1)))
                self.params.view(f"{pre}_q").copy_(torch.from_numpy(w[i + 5].reshape(-1)))
                i += 6

    def _full_table(self) -> torch.Tensor:
        """The whole (V, D) table; with a row-sharded table the shards are all-gathered (save / export only)."""
        if self.exchange is None or self.exchange.world == 1:
            return self.table
        per = self.exchange.per
        mine = torch.zeros(per, self.D, device=self.device)
        mine[: self.table.shape[0]] = self.table
        parts = [torch.empty_like(mine) for _ in range(self.exchange.world)]
        torch.distributed.all_gather(parts, mine, group=self.pg)
        return self.exchange.unshard(parts)

    def count_params(self):
        return self.V * self.D + sum(int(np.prod(s)) for s in self.params.shapes.values()) + 2 * sum(self.units)

    @property
    def _planned(self) -> bool:
        """row-sharded table with the device-planned fixed-capacity exchange (the graph-capturable form)"""
        return self.exchange is not None and self.exchange.mode == "alltoall"

    @property
    def _adam_from_acc(self) -> bool:
        """trainable replicated table on ONE rank with the deterministic accumulator: Adam reads it directly"""
        return (self.train_embedding and self.deterministic and self.exchange is None and not self.multi
                and not self.keep_table_grad)

    def _sparse_dp(self, n_tok: int) -> bool:
        """Data parallel + trainable replicated table: exchange the per-token (id, gradient row) pairs (all-gather, then every
        rank accumulates all of them in the same order-independent fixed-point accumulator and Adam reads it) instead of
        all-reducing the dense (V, D) gradient.  "auto": when the gathered rows are fewer than the table's -- a 250 002 x 1024
        table is a 1 GB all-reduce per step, 8 x 24 000 token rows are 0.8 GB of all-gather traffic and need no fp32 copy of
        the gradient; a 32 000 x 300 table (38 MB) stays dense.  The "auto" choice is PINNED the first time it is asked
        (`pin_table_grad_exchange`: fit() asks with the loader's full-batch shape before the first step), so that a rank's
        short last batch can never pick a different collective than its peers' full ones."""
        if not (self.multi and self.train_embedding and self.exchange is None and self.deterministic and not self.keep_table_grad):
            return False
        if self.table_grad_exchange != "auto":
            return self.table_grad_exchange == "sparse"
        return self.pin_table_grad_exchange(n_tok)

    def pin_table_grad_exchange(self, n_tok_full: int) -> bool:
        """Decide table_grad_exchange="auto" ONCE from the FULL-batch token count (the same number on every rank: batch size and
        title shapes are configuration).  Returns True for the sparse exchange."""
        if self._sparse_pin is None:
            self._sparse_pin = bool(self.world * int(n_tok_full) < self.V)
        return self._sparse_pin

    @property
    def needs_equal_batches(self) -> bool:
        """True when the step's collectives are sized by the local batch SHAPE, so every rank must run the same shape in
        every step (fit() then leaves a shard's short last batch out): the row-sharded lookup's equal-split all-to-alls,
        and the sparse table-gradient exchange of a replicated trainable table (all-gathers of n_tok ids / rows).  The dense
        all-reduces are shape-independent: a 32 000 x 300 table under "auto" trains on every batch, like the reference."""
        if self.world <= 1:
            return False
        if self.exchange is not None:
            return True
        if not (self.train_embedding and self.deterministic and not self.keep_table_grad):
            return False
        if self.table_grad_exchange != "auto":
            return self.table_grad_exchange == "sparse"
        return True if self._sparse_pin is None else self._sparse_pin  # not decided yet: the conservative answer

    @property
    def graph_capable(self) -> bool:
        return self.exchange is None or self._planned

    def allreduce_bytes(self, n_tok: int = 0) -> int:
        """bytes each rank contributes to the per-step gradient all-reduce(s) (all-gather for the sparse table exchange)"""
        n = self.params.numel * 4
        if self.train_embedding and self.exchange is None:
            n += n_tok * (4 + self.D * 4) if self._sparse_dp(n_tok) else self.table_grad.numel() * 4
        return n

    def enable_graphs(self, flag=True):
        self.use_graph = bool(flag)
        if not flag:
            self._graphs = {}
        return self

    # ------------------------------------------------------------------ optimizer state
    @property
    def learning_rate(self):
        return self._lr

    @learning_rate.setter
    def learning_rate(self, lr):
        self._lr = float(lr)
        st = self.read_state()
        st.lr = self._lr
        self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))

    def read_state(self):
        return _hip.StepState.from_buffer_copy(self.state.cpu().numpy().tobytes())

    # ------------------------------------------------------------------ buffers
    def _news_bufs(self, N, train):
        key = ("news", train)
        b = self._bufs.get(key)
        if b is None or b.n_seq < N:
            b = EncoderBuffers(N, self.T, self.D, self.E, self.A, self.device, own_input=True,
                               need_dx=train and self.train_embedding)
            b.ids = torch.empty(N * self.T, dtype=torch.int32, device=self.device)
            if self._planned:
                from ._dist import PlannedBuffers

                ws = int(_hip.lib().ebn_shard_plan_workspace_ints(self.V, self.exchange.world))
                b.xb = PlannedBuffers(self.exchange, N * self.T, self.device, need_grad=train and self.train_embedding, ws_ints=ws)
            self._bufs[key] = b
            if train:
                self._graphs.clear()  # captured training graphs hold raw pointers into the old buffers (inference is never captured)
        return b

    def _user_bufs(self, B, train, H=None):
        """H: history length of an INFERENCE pass that differs from hparams.history_size (the reference's history-length sweep,
        ebnerd_nrms_doc_hist.py:270-300, scores one trained model on histories truncated to 1..50); training always runs
        hparams.history_size."""
        H = self.H if H is None else int(H)
        key = ("user", train) if H == self.H else ("user", train, H)
        b = self._bufs.get(key)
        if b is None or b.n_seq < B:
            b = EncoderBuffers(B, H, self.E, self.E, self.A, self.device, own_input=False, need_dx=False)
            self._bufs[key] = b
            if train and H == self.H:  # only the training step's own buffers are referenced by captured graphs: a history-length
                self._graphs.clear()   # sweep or a truncated-history predict between epochs must not force a re-capture
        return b

    # ------------------------------------------------------------------ C-ABI plumbing
    def _enc_structs(self, pre, b: EncoderBuffers, n_seq, X, drop_site, drop_p):
        pv = self.params.view
        dims = _hip.EncoderDims(n_seq, b.L, b.Din, self.h, self.d, self.A, drop_site, drop_p)
        params = _hip.EncoderParams(pv(f"{pre}_Wqkv").data_ptr(), pv(f"{pre}_W").data_ptr(),
                                    pv(f"{pre}_b").data_ptr(), pv(f"{pre}_q").data_ptr())
        acts = _hip.EncoderActs(X.data_ptr(), b.QKV.data_ptr(), b.Y.data_ptr(), b.U.data_ptr(), b.w.data_ptr(),
                                b.out.data_ptr())
        return dims, params, acts

    def _encoder_fwd(self, pre, b, n_seq, X, train, n_first=None):
        site, p = (1, self.p) if (train and pre == "n" and self.p > 0) else (-1, 0.0)
        st = _hip.ptr(self.state) if train else None
        if pre == "n" and self.mlp is not None:
            return self._news_encoder_fwd_mlp(b, n_seq, X, train, n_seq if n_first is None else n_first)
        if pre == "n" and self.precision == "split":
            return self._news_encoder_fwd_split(b, n_seq, X, st, site, p)
        dims, params, acts = self._enc_structs(pre, b, n_seq, X, site, p)
        _hip.call("ebn_encoder_fwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts),
                  ctypes.byref(self._fwd_scratch(b)), st, _hip.stream_handle())

    @staticmethod
    def _fwd_scratch(b):
        return b.fwd_scratch

    def _news_encoder_fwd_mlp(self, b, n_seq, X, train, n_first):
        """News encoder with the per-token Dense/BN stack: QKV -> attention (NO dropout here, nrms.py:142-152) ->
        stack (call sites: titles [0,n_first) and [n_first,n_seq)) -> AttLayer2."""
        S, E, A, T = _hip.stream_handle, self.E, self.A, b.L
        R = n_seq * T
        pv = self.params.view
        _hip.call("ebn_gemm_f32_site", 0, 0, R, 3 * E, b.Din, ctypes.c_float(1.0), _hip.ptr(X), b.Din, _hip.ptr(pv("n_Wqkv")), 3 * E,
                  ctypes.c_float(0.0), _hip.ptr(b.QKV), 3 * E, _hip.ptr(b.ws), b.ws.numel(), 1, S())
        _hip.call("ebn_attn_fwd_f32", _hip.ptr(b.QKV), 3 * E, _hip.ptr(b.Y), E, n_seq, T, self.h, self.d, None, -1, ctypes.c_float(0.0), S())
        b.Z = self.mlp.forward(b.Y, n_first * T, (n_seq - n_first) * T, train, self.state, self.p)
        _hip.call("ebn_gemm_f32_ws", 0, 0, R, A, E, ctypes.c_float(1.0), _hip.ptr(b.Z), E, _hip.ptr(pv("n_W")), A, ctypes.c_float(0.0),
                  _hip.ptr(b.U), A, _hip.ptr(b.ws), b.ws.numel(), S())
        _hip.call("ebn_attpool_fwd_f32", _hip.ptr(b.U), _hip.ptr(pv("n_b")), _hip.ptr(pv("n_q")), _hip.ptr(b.Z), _hip.ptr(b.out),
                  _hip.ptr(b.w), n_seq, T, E, A, S())

    def _news_encoder_bwd_mlp(self, b, n_seq, X, dout, dX, n_first):
        S, E, A, T = _hip.stream_handle, self.E, self.A, b.L
        R = n_seq * T
        pv, g = self.params.view, self.params.g
        ws, wsn = _hip.ptr(b.ws), b.ws.numel()
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        _hip.call("ebn_attpool_bwd_pool_f32", _hip.ptr(b.Z), _hip.ptr(b.w), _hip.ptr(dout), _hip.ptr(b.dY), _hip.ptr(b.de), n_seq, T, E, S())
        _hip.call("ebn_attpool_bwd_dpre_f32", _hip.ptr(b.U), _hip.ptr(pv("n_q")), _hip.ptr(b.de), _hip.ptr(g("n_q")), _hip.ptr(g("n_b")),
                  _hip.ptr(b.partials), R, A, 0, S())
        _hip.call("ebn_gemm_f32_ws", 1, 0, E, A, R, one, _hip.ptr(b.Z), E, _hip.ptr(b.U), A, zero, _hip.ptr(g("n_W")), A, ws, wsn, S())
        _hip.call("ebn_gemm_f32_ws", 0, 1, R, E, A, one, _hip.ptr(b.U), A, _hip.ptr(pv("n_W")), A, one, _hip.ptr(b.dY), E, ws, wsn, S())
        dYattn = self.mlp.backward(b.dY, b.Y, n_first * T, (n_seq - n_first) * T, self.state, self.p, need_dx0=True,
                                   loss_dev=self.loss_dev)
        _hip.call("ebn_attn_bwd_f32", _hip.ptr(b.QKV), 3 * E, _hip.ptr(dYattn), E, _hip.ptr(b.dQKV), 3 * E, n_seq, T, self.h, self.d,
                  None, -1, ctypes.c_float(0.0), S())
        _hip.call("ebn_gemm_f32_ws", 1, 0, b.Din, 3 * E, R, one, _hip.ptr(X), b.Din, _hip.ptr(b.dQKV), 3 * E, zero, _hip.ptr(g("n_Wqkv")),
                  3 * E, ws, wsn, S())
        if dX is not None:
            _hip.call("ebn_gemm_f32_ws", 0, 1, R, b.Din, 3 * E, one, _hip.ptr(b.dQKV), 3 * E, _hip.ptr(pv("n_Wqkv")), 3 * E, zero,
                      _hip.ptr(dX), b.Din, ws, wsn, S())

    # ---- precision = "split": the news encoder's two (three, with a trainable table) big projection GEMMs run as bf16x6 split
    # products on the bf16 matrix pipe (ebn_gemm_f32_prec, precision 1); everything else is the same kernels in the same
    # order as ebn_encoder_fwd_f32 / ebn_encoder_bwd_f32 (csrc/ebn_encoder.hip)
    def _split_ws(self, b):
        if getattr(b, "split_ws", None) is None:
            f = _hip.lib().ebn_gemm_prec_workspace_bytes
            R, E3, D = b.R, 3 * b.E, b.Din
            nbytes = max(int(f(R, E3, D, 1)), int(f(D, E3, R, 1)), int(f(R, D, E3, 1)) if b.dX is not None else 0)
            b.split_ws = torch.empty(nbytes // 4 + 64, device=self.device)
            b.split_ws_bytes = nbytes
        return b.split_ws

    def _split_bufs(self, b):
        """Plane sets of a training step in split precision (allocated once per buffer set): X in both orientations (written by
        the gather), Wqkv^T and dQKV^T (split passes), and the split-K partials of the weight-gradient GEMM."""
        sb = getattr(b, "split_bufs", None)
        if sb is None:
            L = _hip.lib()
            R, E3, D = b.R, 3 * b.E, b.Din
            u8 = lambda n: torch.empty(int(n) + 64, dtype=torch.uint8, device=self.device)
            sb = b.split_bufs = {"XN": u8(L.ebn_planes_bytes(R, D)), "XT": u8(L.ebn_planes_bytes(D, R)), "Wp": u8(L.ebn_planes_bytes(E3, D)),
                                 "dQp": u8(L.ebn_planes_bytes(E3, R)),
                                 "part": torch.empty(max(int(L.ebn_gemm_planes_workspace_floats(D, E3, R)), int(L.ebn_gemm_planes_workspace_floats(R, E3, D)), 1),
                                                     device=self.device)}
        return sb

    def _gemm_prec(self, b, tA, tB, M, N, K, A, lda, Bm, ldb, C, ldc):
        ws = self._split_ws(b)
        _hip.call("ebn_gemm_f32_prec", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), lda, _hip.ptr(Bm), ldb, ctypes.c_float(0.0),
                  _hip.ptr(C), ldc, _hip.ptr(ws), b.split_ws_bytes, 1, _hip.stream_handle())

    def _news_encoder_fwd_split_gemm(self, b, R):
        S, E = _hip.stream_handle, self.E
        sb = self._split_bufs(b)
        _hip.call("ebn_split_planes_f32", _hip.ptr(self.params.view("n_Wqkv")), 3 * E, 3 * E, b.Din, 1, _hip.ptr(sb["Wp"]), S())
        _hip.call("ebn_gemm_planes_f32", _hip.ptr(sb["XN"]), R, _hip.ptr(sb["Wp"]), 3 * E, b.Din, ctypes.c_float(1.0), ctypes.c_float(0.0),
                  _hip.ptr(b.QKV), 3 * E, _hip.ptr(sb["part"]), sb["part"].numel(), S())

    def _news_encoder_fwd_split(self, b, n_seq, X, st, site, p):
        S, E, A, T = _hip.stream_handle, self.E, self.A, b.L
        R = n_seq * T
        pv = self.params.view
        if getattr(b, "planes_rows", -1) == R:   # layers.py:214,220,226 on the planes the gather wrote
            self._news_encoder_fwd_split_gemm(b, R)
        else:
            self._gemm_prec(b, 0, 0, R, 3 * E, b.Din, X, b.Din, pv("n_Wqkv"), 3 * E, b.QKV, 3 * E)
        _hip.call("ebn_attn_fwd_f32", _hip.ptr(b.QKV), 3 * E, _hip.ptr(b.Y), E, n_seq, T, self.h, self.d, st, site, ctypes.c_float(p), S())
        _hip.call("ebn_gemm_f32_ws", 0, 0, R, A, E, ctypes.c_float(1.0), _hip.ptr(b.Y), E, _hip.ptr(pv("n_W")), A, ctypes.c_float(0.0),
                  _hip.ptr(b.U), A, _hip.ptr(b.ws), b.ws.numel(), S())
        _hip.call("ebn_attpool_fwd_f32", _hip.ptr(b.U), _hip.ptr(pv("n_b")), _hip.ptr(pv("n_q")), _hip.ptr(b.Y), _hip.ptr(b.out),
                  _hip.ptr(b.w), n_seq, T, E, A, S())

    def _news_encoder_bwd_split(self, b, n_seq, X, dout, dX, site, p):
        self._news_bwd_p1(b, n_seq, dout)
        self._news_bwd_p2(b, n_seq, X, dout, site, p)
        if dX is not None:
            self._news_bwd_p3(b, n_seq, dX)

    # The news encoder's backward as three runs of kernels -- the kernels of ebn_encoder_bwd_f32 (csrc/ebn_encoder.hip) in its
    # order, cut where a multi-rank step can start a collective on what has just been written:
    #   p1  AttLayer2 backward: de; dq, db, d(pre-tanh); dW = Y^T.dpre | dY = dpre.W^T           -> every dense gradient but dWqkv
    #   p2  self-attention core backward (pooling term folded in where supported); dWqkv = X^T.dQKV   -> the last dense gradient
    #   p3  dX = dQKV.Wqkv^T (trainable table only)
    def _news_bwd_p1(self, b, n_seq, dout):
        S, E, A, T = _hip.stream_handle, self.E, self.A, b.L
        R = n_seq * T
        pv, g = self.params.view, self.params.g
        ws, wsn = _hip.ptr(b.ws), b.ws.numel()
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        _hip.call("ebn_attpool_bwd_pool_f32", _hip.ptr(b.Y), _hip.ptr(b.w), _hip.ptr(dout), None, _hip.ptr(b.de), n_seq, T, E, S())
        _hip.call("ebn_attpool_bwd_dpre_f32", _hip.ptr(b.U), _hip.ptr(pv("n_q")), _hip.ptr(b.de), _hip.ptr(g("n_q")), _hip.ptr(g("n_b")),
                  _hip.ptr(b.partials), R, A, 0, S())
        if self._fold_pooling(T):
            _hip.call("ebn_dense_bwd_pair_f32", R, E, A, _hip.ptr(b.Y), E, _hip.ptr(b.U), A, _hip.ptr(pv("n_W")), A, zero, _hip.ptr(g("n_W")), A,
                      _hip.ptr(b.dY), E, ws, wsn, S())
        else:
            _hip.call("ebn_gemm_f32_ws", 1, 0, E, A, R, one, _hip.ptr(b.Y), E, _hip.ptr(b.U), A, zero, _hip.ptr(g("n_W")), A, ws, wsn, S())
            _hip.call("ebn_gemm_f32_rank1", R, E, A, one, _hip.ptr(b.U), A, _hip.ptr(pv("n_W")), A, _hip.ptr(b.dY), E, _hip.ptr(b.w), _hip.ptr(dout), E,
                      T, ws, wsn, S())

    def _fold_pooling(self, T) -> bool:
        return int(_hip.lib().ebn_attn_bwd_pooled_supported(T, self.d)) != 0 and self.E % 4 == 0

    def _news_bwd_p2(self, b, n_seq, X, dout, site, p):
        S, E, T = _hip.stream_handle, self.E, b.L
        R = n_seq * T
        g = self.params.g
        st = _hip.ptr(self.state)
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        if self._fold_pooling(T):
            _hip.call("ebn_attn_bwd_pooled_f32", _hip.ptr(b.QKV), 3 * E, _hip.ptr(b.dY), E, _hip.ptr(b.w), _hip.ptr(dout), E, _hip.ptr(b.dQKV), 3 * E,
                      n_seq, T, self.h, self.d, st, site, ctypes.c_float(p), S())
        else:
            _hip.call("ebn_attn_bwd_f32", _hip.ptr(b.QKV), 3 * E, _hip.ptr(b.dY), E, _hip.ptr(b.dQKV), 3 * E, n_seq, T, self.h, self.d, st, site,
                      ctypes.c_float(p), S())
        if self.precision != "split":                       # dWqkv = X^T . dQKV
            _hip.call("ebn_gemm_f32_ws", 1, 0, b.Din, 3 * E, R, one, _hip.ptr(X), b.Din, _hip.ptr(b.dQKV), 3 * E, zero, _hip.ptr(g("n_Wqkv")), 3 * E,
                      _hip.ptr(b.ws), b.ws.numel(), S())
        elif getattr(b, "planes_rows", -1) == R:            # ... on the transposed planes the gather wrote
            sb = self._split_bufs(b)
            _hip.call("ebn_split_planes_f32", _hip.ptr(b.dQKV), 3 * E, 3 * E, R, 1, _hip.ptr(sb["dQp"]), S())
            _hip.call("ebn_gemm_planes_f32", _hip.ptr(sb["XT"]), b.Din, _hip.ptr(sb["dQp"]), 3 * E, R, one, zero, _hip.ptr(g("n_Wqkv")), 3 * E,
                      _hip.ptr(sb["part"]), sb["part"].numel(), S())
        else:
            self._gemm_prec(b, 1, 0, b.Din, 3 * E, R, X, b.Din, b.dQKV, 3 * E, g("n_Wqkv"), 3 * E)

    def _news_bwd_p3(self, b, n_seq, dX):
        E, R = self.E, n_seq * b.L
        pv = self.params.view
        if self.precision != "split":                       # dX = dQKV . Wqkv^T
            _hip.call("ebn_gemm_f32_ws", 0, 1, R, b.Din, 3 * E, ctypes.c_float(1.0), _hip.ptr(b.dQKV), 3 * E, _hip.ptr(pv("n_Wqkv")), 3 * E,
                      ctypes.c_float(0.0), _hip.ptr(dX), b.Din, _hip.ptr(b.ws), b.ws.numel(), _hip.stream_handle())
        else:
            self._gemm_prec(b, 0, 1, R, b.Din, 3 * E, b.dQKV, 3 * E, pv("n_Wqkv"), 3 * E, dX, b.Din)

    def roofline_kernels(self, B, C):
        """Launchers of single kernels of the training step at batch shape (B, C), on the step's own buffers and arguments
        (bench.py captures them into hipGraphs and times them with HIP events on the launch stream): "qkv_gemm" = zero-
        argument launcher of the news encoder's Q|K|V projection; "gather" = factory ids -> launcher of the title-token
        embedding gather (+ dropout) for that id set."""
        N = B * (self.H + C)
        nb, _ub = self._train_bufs(B, C)
        S, E, R = _hip.stream_handle, self.E, N * self.T
        pv = self.params.view
        site, p = (0, self.p) if self.p > 0 else (-1, 0.0)
        st = _hip.ptr(self.state)

        def qkv_gemm():
            if self.precision == "split":  # as the step runs it: the weight split pass + the bf16x6 GEMM on the gather's planes
                nb.planes_rows = R  # (a training step has filled them)
                return self._news_encoder_fwd_split_gemm(nb, R)
            _hip.call("ebn_gemm_f32_site", 0, 0, R, 3 * E, nb.Din, ctypes.c_float(1.0), _hip.ptr(nb.X), nb.Din,
                      _hip.ptr(pv("n_Wqkv")), 3 * E, ctypes.c_float(0.0), _hip.ptr(nb.QKV), 3 * E, _hip.ptr(nb.ws), nb.ws.numel(), 1, S())

        def make_gather(ids):
            # `ids`: (R,) int32 token ids on the device.  bench.py passes the id sets of SEVERAL batches and cycles through
            # them, so that consecutive launches read different table rows (re-reading one batch's rows would be served by
            # the 256 MB memory-side cache, not HBM).  With a row-sharded table the ids are folded into this rank's shard.
            table_rows = self.table.shape[0]
            src = ids if table_rows == self.V else torch.remainder(ids, table_rows).to(torch.int32)

            def gather():
                self._gather_tokens(nb, src, self.table, table_rows, R, st, site, p, True)

            return gather

        return {"qkv_gemm": qkv_gemm, "gather": make_gather}

    def _encoder_bwd(self, pre, b, n_seq, X, dout, dX, n_first=None):
        if pre == "n" and self.mlp is not None:
            return self._news_encoder_bwd_mlp(b, n_seq, X, dout, dX, n_seq if n_first is None else n_first)
        site, p = (1, self.p) if (pre == "n" and self.p > 0) else (-1, 0.0)
        if pre == "n" and self.precision == "split":
            return self._news_encoder_bwd_split(b, n_seq, X, dout, dX, site, p)
        dims, params, acts = self._enc_structs(pre, b, n_seq, X, site, p)
        g = self.params.g
        grads = _hip.EncoderGrads(g(f"{pre}_Wqkv").data_ptr(), g(f"{pre}_W").data_ptr(), g(f"{pre}_b").data_ptr(),
                                  g(f"{pre}_q").data_ptr())
        scratch = _hip.EncoderScratch(b.dY.data_ptr(), b.dQKV.data_ptr(), b.de.data_ptr(), b.partials.data_ptr(),
                                      b.ws.data_ptr(), b.ws.numel())
        _hip.call("ebn_encoder_bwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), _hip.ptr(dout),
                  ctypes.byref(grads), ctypes.byref(scratch), _hip.ptr(dX), 0, _hip.ptr(self.state),
                  _hip.stream_handle())

    def _news_forward(self, b, N, train, n_first=None, looked_up=False):
        """a1..a4 for N titles whose ids are already in b.ids -> b.out[:N].  n_first = titles of the first
        TimeDistributed call site (history); only the BatchNorm of the optional Dense stack cares.
        looked_up: the row-sharded lookup of these ids already ran (the train step interleaves it with collectives)."""
        site, p = (0, self.p) if (train and self.p > 0) else (-1, 0.0)
        st = _hip.ptr(self.state) if train else None
        n_tok = N * self.T
        if self._planned:
            # row-sharded table: the device plans the routes, two equal-split all-to-alls fetch the distinct rows, then
            # the same gather kernel expands them to token order (ids = slots of the received buffer), dropout fused
            if not looked_up:
                for _kind, fn in self._lookup_segments(b, N):
                    fn()
            xb = b.xb
            self._gather_tokens(b, xb.inv, xb.rows, self.exchange.world * self.exchange.capacity(n_tok), n_tok, st, site, p, train)
            return self._encoder_fwd("n", b, N, b.X, train, n_first)
        if self.exchange is not None:
            # validation forms (host-planned, eager only): route the distinct ids to their owners, fetch the rows over
            # RCCL, then expand them to token order (ids = positions in the unique list) with dropout fused
            b.plan = self.exchange.plan(b.ids[: n_tok])
            b.rows_uniq = self.exchange.lookup(b.plan, self._local_gather)
            self._gather_tokens(b, b.plan.inv, b.rows_uniq, b.rows_uniq.shape[0], n_tok, st, site, p, train)
            return self._encoder_fwd("n", b, N, b.X, train, n_first)
        if not train and self.fuse_eval_gather and self.mlp is None and self.precision == "exact":
            # inference: no Dropout between Embedding and the projection (nrms.py:136 is training-only), so the gather rides in the
            # projection's A-operand fetch -- table rows -> LDS -> MFMA, the (n_tok, D) activations are never written or re-read
            dims, params, acts = self._enc_structs("n", b, N, b.X, -1, 0.0)
            rc = _hip.lib().ebn_encoder_fwd_gather_f32(ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), ctypes.byref(self._fwd_scratch(b)),
                                                       _hip.ptr(b.ids), _hip.ptr(self.table), self.V, _hip.ptr(self.oob_flag), _hip.stream_handle())
            if rc == 0:
                return
            if rc != -2:  # EBN_ERR_UNSUPPORTED: a shape the fused kernel does not take (fewer than 256 token rows, ...): two steps
                raise _hip.HipError(f"ebn_encoder_fwd_gather_f32 failed with code {rc}")
        self._gather_tokens(b, b.ids, self.table, self.V, n_tok, st, site, p, train)
        self._encoder_fwd("n", b, N, b.X, train, n_first)

    def _gather_tokens(self, b, ids, table, table_rows, n_tok, st, site, p, train):
        """a1: the title-token embedding gather (+ dropout) of n_tok tokens.  Exact precision: fp32 rows into b.X.  Split
        precision, training step: the rows go out as bf16 planes in the two orientations the projection GEMMs contract over
        (b.XN: tokens x D, b.XT: D x tokens) -- nothing else reads the fp32 X, so it is not written at all."""
        if self.precision == "split" and train and self.mlp is None and n_tok <= b.R:
            sb = self._split_bufs(b)
            _hip.call("ebn_gather_split_planes_f32", _hip.ptr(ids), _hip.ptr(table), n_tok, self.D, table_rows, st, site, ctypes.c_float(p),
                      _hip.ptr(self.oob_flag), _hip.ptr(sb["XN"]), _hip.ptr(sb["XT"]), _hip.stream_handle())
            b.planes_rows = n_tok
            return
        b.planes_rows = -1
        _hip.call("ebn_gather_rows_f32", _hip.ptr(ids), _hip.ptr(table), _hip.ptr(b.X), n_tok, self.D, table_rows, st, site,
                  ctypes.c_float(p), _hip.ptr(self.oob_flag), _hip.stream_handle())

    # ------------------------------------------------------------------ public compute
    def encode_news(self, ids, chunk=8192) -> torch.Tensor:
        """newsencoder: (N,T) ids -> (N,E) device tensor (inference mode, nrms.py:116-159)."""
        ids = ids if isinstance(ids, torch.Tensor) else np.asarray(ids)
        N = ids.shape[0]
        if ids.ndim != 2 or ids.shape[1] != self.T:
            raise ValueError(f"expected ids of shape (N, {self.T}), got {tuple(ids.shape)}")
        out = torch.empty(N, self.E, device=self.device)
        for s in range(0, N, chunk):
            n = min(chunk, N - s)
            b = self._news_bufs(min(chunk, N), False)
            self._upload_ids(b.ids, ids[s:s + n])
            self._news_forward(b, n, False)
            out[s:s + n].copy_(b.out[:n])
        self._check_oob()  # local read with a replicated table (callers run inference on one rank), collective when row-sharded
        return out

    def encode_users_from_news(self, NEh: torch.Tensor) -> torch.Tensor:
        """user encoder on already-encoded history (B,H,E) -> (B,E)  (nrms.py:108-111).  H is taken from the input: the
        layers have no weight that depends on it (layers.py:200-254, 55-81)."""
        B, H = NEh.shape[0], NEh.shape[1]
        X = NEh.reshape(B * H, self.E).contiguous()
        b = self._user_bufs(B, False, H)
        self._encoder_fwd("u", b, B, X, False)
        return b.out[:B].clone()

    def encode_users(self, his) -> torch.Tensor:
        his = his if isinstance(his, torch.Tensor) else np.asarray(his)
        B, H = his.shape[0], his.shape[1]
        NEh = self.encode_news(his.reshape(B * H, self.T))
        return self.encode_users_from_news(NEh.view(B, H, self.E))

    def forward(self, his, pred, mode="softmax"):
        """(B,H,T),(B,C,T) ids -> (probs (B,C), scores (B,C)) device tensors, inference mode."""
        his = his if isinstance(his, torch.Tensor) else np.asarray(his)
        pred = pred if isinstance(pred, torch.Tensor) else np.asarray(pred)
        B, C = his.shape[0], pred.shape[1]
        self._check_shapes(his, pred)
        N = B * (self.H + C)
        nb = self._news_bufs(N, False)
        self._upload_ids(nb.ids, his, pred)
        self._news_forward(nb, N, False, B * self.H)
        ub = self._user_bufs(B, False)
        self._encoder_fwd("u", ub, B, nb.out, False)
        scores = torch.empty(B, C, device=self.device)
        probs = torch.empty(B, C, device=self.device)
        cand = nb.out[B * self.H:]
        _hip.call("ebn_score_fwd_f32", _hip.ptr(cand), _hip.ptr(ub.out), _hip.ptr(scores), _hip.ptr(probs), B, C,
                  self.E, 0 if mode == "softmax" else 1, _hip.stream_handle())
        self._check_oob()
        return probs, scores

    def eval_loss(self, his, pred, y):
        """Inference-mode forward + the compiled loss (validation pass of fit): (loss[1], probs (B,C))."""
        probs, scores = self.forward(his, pred, mode="softmax")
        B, C = scores.shape
        labels = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32))))
        labels = labels.to(device=self.device, dtype=torch.float32).reshape(B, C).contiguous()
        nb, ub = self._news_bufs(B * (self.H + C), False), self._user_bufs(B, False)
        rows = torch.empty(B, device=self.device)
        junk_c = torch.empty(B * C, self.E, device=self.device)
        junk_u = torch.empty(B, self.E, device=self.device)
        loss = torch.empty(1, device=self.device)
        _hip.call("ebn_score_loss_bwd_f32", _hip.ptr(nb.out[B * self.H:]), _hip.ptr(ub.out), _hip.ptr(scores),
                  _hip.ptr(labels), _hip.ptr(rows), _hip.ptr(junk_c), _hip.ptr(junk_u), B, C, self.E,
                  self.loss_kind, ctypes.c_float(1.0 / B), _hip.stream_handle())
        _hip.call("ebn_sum_f32", _hip.ptr(rows), B, ctypes.c_float(1.0), _hip.ptr(loss), 0, _hip.stream_handle())
        return loss, probs

    def pair_scores(self, user, news, u_idx, n_idx, sigmoid=True):
        n = u_idx.numel()
        out = torch.empty(n, device=self.device)
        _hip.call("ebn_pair_score_f32", _hip.ptr(user), _hip.ptr(news), _hip.ptr(u_idx), _hip.ptr(n_idx),
                  _hip.ptr(out), n, self.E, 1 if sigmoid else 0, _hip.stream_handle())
        return out

    def _check_shapes(self, his, pred):
        if his.ndim != 3 or his.shape[1] != self.H or his.shape[2] != self.T:
            raise ValueError(f"his_input_title must be (B, {self.H}, {self.T}), got {tuple(his.shape)}")
        if pred.ndim != 3 or pred.shape[0] != his.shape[0] or pred.shape[2] != self.T:
            raise ValueError(f"pred_input_title must be (B, C, {self.T}), got {tuple(pred.shape)}")

    def _check_oob(self, collective=None):
        """One host read of the device flags (ids out of range, fixed-point accumulator range, exchange overflow).

        collective=True: the flags are MAX-reduced over the group first, so that every rank raises together -- a rank raising
        alone would leave the others blocked in the next step's collectives.  Only the LOCK-STEP callers ask for this: the
        once-per-epoch `check_oob()` of fit() / evaluate() and bench.py.  collective=False: a local read, no communication --
        what the inference entry points (`encode_news`, `forward`, hence `predict` / `scorer.predict`) use with a REPLICATED
        table, because callers run those on one rank only (the reproducibility driver predicts on rank 0 while the other ranks
        have returned: examples/reproducibility_scripts/ebnerd_nrms.py).  Default (None): collective exactly when the table is
        row-sharded -- there `encode_news` is itself a collective (the lookup's all-to-alls), every rank is in the call."""
        if collective is None:
            if self.defer_flag_checks:
                # inside a lock-step caller (evaluate() / fit() with world > 1): the sticky flags are left alone -- neither read
                # nor cleared nor raised on -- and the caller's closing check_oob() MAX-reduces them, so every rank raises
                # together.  A rank raising here alone (its own range_flag, say) would leave its peers in evaluate()'s closing
                # all-reduce with nobody to meet (round-4 ADVICE)
                return
            collective = self.exchange is not None
        collective = bool(collective) and self.multi
        err = None
        if self._planned:  # the plan's own flags first: an overflowed exchange also shows up as zero rows in the gather
            for key in sorted((k for k, b in self._bufs.items() if hasattr(b, "xb")), key=repr):
                try:
                    self.exchange.check(self._bufs[key].xb, collective=collective)
                except Exception as e:  # keep reading: every sticky flag is cleared by this one check, whatever is raised
                    err = err or e
        flags = torch.cat([self.oob_flag, self.range_flag])
        if collective:
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        oob, rng_bad = (int(v) for v in flags.cpu().tolist())
        if oob or rng_bad or err is not None:  # the values are in `flags`: clear BOTH words before raising anything, so that no
            self.oob_flag.zero_()              # stale flag raises a second, spurious error at the next check
            self.range_flag.zero_()
        if err is not None:
            raise err
        if oob != 0:
            raise IndexError(f"token id out of range [0, {self.V}) for the embedding table")
        if rng_bad != 0:
            raise FloatingPointError("embedding gradient left the range of the deterministic fixed-point accumulator (|sum| >= 2^22 "
                                     "or NaN): the run has diverged; deterministic=False accumulates in fp32 instead")

    def check_oob(self):
        """The lock-step form: fit() / evaluate() call this once per epoch on EVERY rank (device-resident batches are not
        range-checked on the host), bench.py after its timed region."""
        self._check_oob(collective=True)

    def sync_moving_statistics(self) -> None:
        """world > 1 with BatchNormalization layers: average the moving mean / variance over the ranks (a COLLECTIVE; fit() calls it
        at the end of every epoch, evaluate() and save_weights() on entry -- see MLPStack.sync_moving_statistics for the contract)."""
        if self.world > 1 and self.mlp is not None:
            self.mlp.sync_moving_statistics(self.pg)

    def l2_penalty(self) -> float:
        """lambda * sum(W^2) over the regularised Dense kernels (0 without the optional per-token stack)."""
        return self.mlp.l2_penalty() if self.mlp is not None else 0.0

    def train_step(self, his, pred, y, return_probs=False, indexed=False):
        """One optimizer step (forward, loss, backward, gradient all-reduce, Keras Adam).
        Returns the batch loss as a 1-element device tensor (no host sync).  With
        ``use_graph`` the kernel sequence of a (B, C) shape is captured once into hipGraphs and
        replayed: the ~30 launches of a step become graph launches with only the collectives of a
        multi-rank step between them (step-dependent scalars live in the device ebn_step_state, so
        the replay sees fresh dropout keys / Adam step sizes)."""
        his = his if isinstance(his, torch.Tensor) else np.asarray(his)
        pred = pred if isinstance(pred, torch.Tensor) else np.asarray(pred)
        B, C = his.shape[0], pred.shape[1]
        advanced = False  # True once a staging launch has advanced the step state
        if indexed:  # (B,H) / (B,C) article-row numbers of the matrix given to set_article_matrix()
            if his.ndim != 2 or his.shape[1] != self.H or pred.ndim != 2 or pred.shape[0] != B:
                raise ValueError(f"indexed batches must be (B, {self.H}) and (B, C), got {tuple(his.shape)} {tuple(pred.shape)}")
            nb, ub = self._train_bufs(B, C)
            y, advanced = self._stage_indexed(nb, his, pred, y)
        else:
            self._check_shapes(his, pred)
            nb, ub = self._train_bufs(B, C)
            if self._device_batch(his, pred, y):  # batch already in HBM in the step's dtypes: one copy launch, not three
                nh = his.numel()              # (and the step-state advance rides in it instead of being a launch of its own)
                _hip.call("ebn_copy3_advance", _hip.ptr(his), _hip.ptr(nb.ids), nh * 4, _hip.ptr(pred), _hip.ptr(nb.ids[nh:]),
                          pred.numel() * 4, _hip.ptr(y), _hip.ptr(nb.labels), y.numel() * 4, _hip.ptr(self.state), BETA1, BETA2,
                          _hip.stream_handle())
                y, advanced = None, True
            elif not isinstance(his, torch.Tensor) and not isinstance(pred, torch.Tensor) and not isinstance(y, torch.Tensor):
                for a in (his, pred):  # ids outside [0, V) raise like TF-CPU's Embedding does
                    if a.size and (a.min() < 0 or a.max() >= self.V):
                        raise IndexError(f"token id out of range [0, {self.V}) for the embedding table")
                self._stage_host([his, pred], nb.ids, y, nb.labels)
                y, advanced = None, True
            else:
                self._upload_ids(nb.ids, his, pred)
        if y is not None:
            labels = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32))))
            nb.labels[: B * C].copy_(labels.reshape(-1).to(device=self.device, dtype=torch.float32), non_blocking=True)
        if self.use_graph and self.graph_capable:
            run = self._graphs.get((B, C, advanced))
            if run is None:
                run = self._capture(B, C, advanced)
            if self.trace is not None:
                self.trace.run(run, self._graph_desc.get((B, C, advanced), []))
            else:
                for fn in run:
                    fn()
        elif self.trace is not None:
            segs = self._segments(B, C, advanced)
            self.trace.run([fn for _kind, fn in segs], [self.SEG_KINDS.get(kind, kind) + (f": {fn.what}" if hasattr(fn, "what") else "") for kind, fn in segs])
        else:
            for _kind, fn in self._segments(B, C, advanced):
                fn()
        if return_probs:  # (loss, probabilities, the labels as staged on the device: the streaming AUC needs no second upload)
            return self.loss_dev, nb.probs[: B * C].view(B, C), nb.labels[: B * C].view(B, C)
        return self.loss_dev

    def _train_bufs(self, B, C):
        N, E = B * (self.H + C), self.E
        nb = self._news_bufs(N, True)
        ub = self._user_bufs(B, True)
        if not hasattr(nb, "dNE"):
            nb.dNE = torch.empty(nb.n_seq, E, device=self.device)
            nb.scores = torch.empty(nb.n_seq, device=self.device)
            nb.probs = torch.empty(nb.n_seq, device=self.device)
            nb.labels = torch.empty(nb.n_seq, device=self.device)
        if self._deferred(C):
            self._defer_bufs(nb)  # (outside any capture)
        if not hasattr(ub, "duser"):
            ub.duser = torch.empty(ub.n_seq, E, device=self.device)
            ub.loss_rows = torch.empty(ub.n_seq, device=self.device)
            ub.head_partials = torch.empty(max(int(_hip.lib().ebn_user_head_partials_len(ub.n_seq, self.A)), 1), device=self.device)
        return nb, ub

    SEG_KINDS = {"k": "kernels", "c": "collective", "a": "collective started asynchronously", "w": "wait for the started collectives"}

    def _fwd_bwd_kernels(self, B, C, sparse_table_grads=False, part="all"):
        """part: "all" = the whole forward + backward (one rank: the stage calls run the news backward as one C call);
        "a" | "b" | "c" = the same kernels cut in three for a multi-rank step (see _segments): a = forward, user stage, news
        AttLayer2 backward; b = attention-core backward + dWqkv; c = dX + the table-gradient accumulation."""
        H, E = self.H, self.E
        N = B * (H + C)
        S = _hip.stream_handle
        nb, ub = self._train_bufs(B, C)
        st = _hip.ptr(self.state)
        site1, p1 = (1, self.p) if self.p > 0 else (-1, 0.0)
        deferred = self._deferred(C)
        if part in ("all", "a"):
            self._adam_done_in_finish = False
        if part == "b":
            if deferred:
                return self._news_bwd_deferred(nb, ub, B, N, "p2")
            return self._news_bwd_p2(nb, N, nb.X, nb.dNE, site1, p1)
        if part in ("all", "a"):
            if deferred:
                self._news_forward(nb, N, True, B * H, looked_up=True)
                self._user_stage_deferred(B, C, nb, ub)
            else:
                self._fwd_user_stage_kernels(B, C, nb, ub)
        if part == "a":
            if deferred:
                return self._news_bwd_deferred(nb, ub, B, N, "p1")
            return self._news_bwd_p1(nb, N, nb.dNE)
        if part == "all" and deferred:
            self._news_bwd_deferred(nb, ub, B, N, "p1")
            self._adam_done_in_finish = self._adam_fused(C)
            if self._adam_done_in_finish:  # the finishing launch carries the optimizer: it runs after the last reader of the weights
                self._news_bwd_deferred(nb, ub, B, N, "p2a")
                if nb.dX is not None:
                    self._news_bwd_p3(nb, N, nb.dX)
                self._news_bwd_deferred(nb, ub, B, N, "p2b")
            else:
                self._news_bwd_deferred(nb, ub, B, N, "p2")
                if nb.dX is not None:
                    self._news_bwd_p3(nb, N, nb.dX)
        elif part == "all":
            self._encoder_bwd("n", nb, N, nb.X, nb.dNE, nb.dX, B * H)
        elif nb.dX is not None:  # part "c"
            self._news_bwd_p3(nb, N, nb.dX)
        self._table_grad_kernels(nb, N, sparse_table_grads)

    def _deferred(self, C) -> bool:
        """whether this step leaves its finishing passes to ONE ebn_grad_finish_f32 launch (the standard configuration: no
        per-token Dense stack, exact precision, the one-launch head and the pooling term folded into the attention backward)"""
        L = _hip.lib()
        return bool(self.defer_finish and self.mlp is None and self.precision == "exact" and self.fuse_user_head
                    and self._fold_pooling(self.T) and self._fold_pooling(self.H)
                    and int(L.ebn_user_head_supported(self.H, C, self.E, self.A)) != 0)

    def _adam_fused(self, C) -> bool:
        """this step's finishing launch applies Adam to the dense parameters (one rank, the deferred launch form)"""
        return bool(self.adam_in_finish and not self.multi and self._deferred(C))

    def _adam_flat_args(self):
        """ebn_adam_flat over this engine's flat buffers; `rest` = the parameters no finishing job produces: the user encoder's kernels"""
        a = getattr(self, "_adam_flat", None)
        if a is None:
            P = self.params
            a = _hip.AdamFlat()
            a.theta, a.grad, a.m, a.v, a.numel = P.data.data_ptr(), P.grad.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.numel
            a.beta1, a.beta2, a.eps, a.grad_scale = BETA1, BETA2, ADAM_EPS, 1.0
            rest = sorted((P.offsets[k], int(np.prod(P.shapes[k]))) for k in ("u_Wqkv", "u_W"))
            a.n_rest = len(rest)
            for i, (off, n) in enumerate(rest):
                a.rest_off[i], a.rest_len[i] = off, n
            self._adam_flat = a
        return a

    def _defer_bufs(self, nb):
        if getattr(nb, "ws_dw", None) is None:
            L = _hip.lib()
            nb.ws_dw = torch.empty(max(int(L.ebn_gemm_partials_workspace_floats(self.E, self.A, nb.R)), 1), device=self.device)
            nb.ws_dwqkv = torch.empty(max(int(L.ebn_gemm_partials_workspace_floats(self.D, 3 * self.E, nb.R)), 1), device=self.device)
            nb.finish_jobs = {}
        return nb

    def _user_stage_deferred(self, B, C, nb, ub):
        """ebn_user_stage_train_f32's fused branch (csrc/ebn_encoder.hip), kernel by kernel, with the head's reduction launch left
        out: user encoder forward up to the AttLayer2 matmul, the one-launch head, the two gradient-GEMM pairs around the
        attention backward.  Writes d(cand), d(history news vectors) = dNE[:B*H], the user encoder's weight gradients except
        d(q) / d(b), and the per-impression partials / loss rows the finishing pass sums."""
        S, H, E, A = _hip.stream_handle, self.H, self.E, self.A
        R = B * H
        pv, g = self.params.view, self.params.g
        ws, wsn = _hip.ptr(ub.ws), ub.ws.numel()
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        X, cand, dcand = nb.out, nb.out[B * H:], nb.dNE[B * H:]
        _hip.call("ebn_gemm_f32_site", 0, 0, R, 3 * E, E, one, _hip.ptr(X), E, _hip.ptr(pv("u_Wqkv")), 3 * E, zero, _hip.ptr(ub.QKV), 3 * E, ws, wsn, 1, S())
        _hip.call("ebn_attn_fwd_f32", _hip.ptr(ub.QKV), 3 * E, _hip.ptr(ub.Y), E, B, H, self.h, self.d, None, -1, zero, S())
        _hip.call("ebn_gemm_f32_ws", 0, 0, R, A, E, one, _hip.ptr(ub.Y), E, _hip.ptr(pv("u_W")), A, zero, _hip.ptr(ub.U), A, ws, wsn, S())
        _hip.call("ebn_user_head_train_f32", _hip.ptr(ub.U), _hip.ptr(pv("u_b")), _hip.ptr(pv("u_q")), _hip.ptr(ub.Y), _hip.ptr(cand), _hip.ptr(nb.labels),
                  _hip.ptr(ub.w), _hip.ptr(ub.out), _hip.ptr(nb.scores), _hip.ptr(nb.probs), _hip.ptr(ub.loss_rows), _hip.ptr(self.loss_dev), _hip.ptr(dcand),
                  _hip.ptr(ub.duser), _hip.ptr(ub.de), None, None, _hip.ptr(ub.head_partials), B, H, C, E, A, self.loss_kind, ctypes.c_float(1.0 / B), S())
        _hip.call("ebn_dense_bwd_pair_f32", R, E, A, _hip.ptr(ub.Y), E, _hip.ptr(ub.U), A, _hip.ptr(pv("u_W")), A, zero, _hip.ptr(g("u_W")), A,
                  _hip.ptr(ub.dY), E, ws, wsn, S())
        _hip.call("ebn_attn_bwd_pooled_f32", _hip.ptr(ub.QKV), 3 * E, _hip.ptr(ub.dY), E, _hip.ptr(ub.w), _hip.ptr(ub.duser), E, _hip.ptr(ub.dQKV), 3 * E,
                  B, H, self.h, self.d, None, -1, zero, S())
        _hip.call("ebn_dense_bwd_pair_f32", R, E, 3 * E, _hip.ptr(X), E, _hip.ptr(ub.dQKV), 3 * E, _hip.ptr(pv("u_Wqkv")), 3 * E, zero, _hip.ptr(g("u_Wqkv")),
                  3 * E, _hip.ptr(nb.dNE), E, ws, wsn, S())

    def _news_bwd_deferred(self, nb, ub, B, N, part):
        """The news encoder's backward (the kernels of ebn_encoder_bwd_f32 in its order) with the combining passes of its three
        reductions left out, then -- after the last weight gradient -- ONE finishing launch for them and for the user head's.
        part: "p1" AttLayer2 backward | "p2" attention core + dWqkv + the finishing launch | "p2a" / "p2b": p2's two halves (the
        finishing launch with the optimizer inside runs after the input-gradient GEMM of a trainable table)."""
        S, E, A, T, D = _hip.stream_handle, self.E, self.A, self.T, self.D
        R = N * T
        pv, g = self.params.view, self.params.g
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        L = _hip.lib()
        self._defer_bufs(nb)
        if part == "p1":
            _hip.call("ebn_attpool_bwd_pool_f32", _hip.ptr(nb.Y), _hip.ptr(nb.w), _hip.ptr(nb.dNE), None, _hip.ptr(nb.de), N, T, E, S())
            _hip.call("ebn_attpool_bwd_dpre_f32", _hip.ptr(nb.U), _hip.ptr(pv("n_q")), _hip.ptr(nb.de), None, None, _hip.ptr(nb.partials), R, A, 0, S())
            n = ctypes.c_int32(0)
            _hip.call("ebn_gemm_f32_partials", 1, 0, E, A, R, one, _hip.ptr(nb.Y), E, _hip.ptr(nb.U), A, _hip.ptr(nb.ws_dw), nb.ws_dw.numel(), ctypes.byref(n), S())
            nb.finish_jobs["dW"] = int(n.value)
            _hip.call("ebn_gemm_f32_ws", 0, 1, R, E, A, one, _hip.ptr(nb.U), A, _hip.ptr(pv("n_W")), A, zero, _hip.ptr(nb.dY), E, _hip.ptr(nb.ws), nb.ws.numel(), S())
            return
        site, p = (1, self.p) if self.p > 0 else (-1, 0.0)
        if part in ("p2", "p2a"):
            _hip.call("ebn_attn_bwd_pooled_f32", _hip.ptr(nb.QKV), 3 * E, _hip.ptr(nb.dY), E, _hip.ptr(nb.w), _hip.ptr(nb.dNE), E, _hip.ptr(nb.dQKV), 3 * E,
                      N, T, self.h, self.d, _hip.ptr(self.state), site, ctypes.c_float(p), S())
            n = ctypes.c_int32(0)
            _hip.call("ebn_gemm_f32_partials", 1, 0, D, 3 * E, R, one, _hip.ptr(nb.X), D, _hip.ptr(nb.dQKV), 3 * E, _hip.ptr(nb.ws_dwqkv), nb.ws_dwqkv.numel(),
                      ctypes.byref(n), S())
            nb.finish_jobs["dWqkv"] = int(n.value)
            if part == "p2a":
                return
        # the latency-bound jobs FIRST (a few blocks each walking a long chain of partials): blocks are dispatched in order, and behind
        # the 1200 blocks of the dWqkv sum they would only start when it is nearly done (measured: 16 us that way, the sum alone 8)
        jobs = (_hip.FinishJob * 4)()
        jobs[0] = _hip.FinishJob(_hip.FINISH_COLRED, int(L.ebn_attpool_partials_len(R, A)) // (2 * A), 1, A, nb.partials.data_ptr(), g("n_q").data_ptr(),
                                 g("n_b").data_ptr(), A, 0.0, 1.0, None, None)
        jobs[1] = _hip.FinishJob(_hip.FINISH_HEAD, 1, B, A, ub.head_partials.data_ptr(), g("u_q").data_ptr(), g("u_b").data_ptr(), A, 0.0, 1.0,
                                 ub.loss_rows.data_ptr(), self.loss_dev.data_ptr())
        jobs[2] = _hip.FinishJob(_hip.FINISH_SPLITK, nb.finish_jobs["dW"], E, A, nb.ws_dw.data_ptr(), g("n_W").data_ptr(), None, A, 0.0, 1.0, None, None)
        jobs[3] = _hip.FinishJob(_hip.FINISH_SPLITK, nb.finish_jobs["dWqkv"], D, 3 * E, nb.ws_dwqkv.data_ptr(), g("n_Wqkv").data_ptr(), None, 3 * E, 0.0, 1.0, None, None)
        if part == "p2b":  # the optimizer rides in the finishing launch (one rank)
            _hip.call("ebn_grad_finish_adam_f32", jobs, 4, ctypes.byref(self._adam_flat_args()), _hip.ptr(self.state), S())
        else:
            _hip.call("ebn_grad_finish_f32", jobs, 4, S())

    def _fwd_user_stage_kernels(self, B, C, nb, ub):
        H, E = self.H, self.E
        N = B * (H + C)
        S = _hip.stream_handle
        st = _hip.ptr(self.state)
        # ---- forward
        self._news_forward(nb, N, True, B * H, looked_up=True)
        cand, dcand = nb.out[B * H:], nb.dNE[B * H:]
        # user encoder (its input: the first B*H rows of the news vectors, no copy) + scorer + compiled loss, forward and
        # backward, as one stage call: the per-impression middle of it is ONE launch where it fits (ebn_user_head_train_f32);
        # writes the batch loss, d(cand) and d(history news vectors) = dNE[:B*H]
        dims, params, acts = self._enc_structs("u", ub, B, nb.out, -1, 0.0)
        g = self.params.g
        grads = _hip.EncoderGrads(g("u_Wqkv").data_ptr(), g("u_W").data_ptr(), g("u_b").data_ptr(), g("u_q").data_ptr())
        scratch = _hip.EncoderScratch(ub.dY.data_ptr(), ub.dQKV.data_ptr(), ub.de.data_ptr(), ub.partials.data_ptr(),
                                      ub.ws.data_ptr(), ub.ws.numel())
        _hip.call("ebn_user_stage_train_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), _hip.ptr(cand),
                  _hip.ptr(nb.labels), _hip.ptr(nb.scores), _hip.ptr(nb.probs), _hip.ptr(ub.loss_rows), _hip.ptr(self.loss_dev),
                  _hip.ptr(dcand), _hip.ptr(ub.duser), ctypes.byref(grads), ctypes.byref(scratch),
                  _hip.ptr(ub.head_partials) if self.fuse_user_head else None, _hip.ptr(nb.dNE), C, self.loss_kind,
                  ctypes.c_float(1.0 / B), st, S())

    def _accumulate_fixed(self, ids, dX, n_tok, st, site, p):
        """table_acc += the (id, gradient row) pairs of n_tok tokens, in the order-independent fixed-point accumulator.  Duplicate
        ids among every 64 consecutive tokens are combined in registers first (one atomic per distinct id and column: left-padded
        history slots and Zipfian tokens put ~20 % of a real batch on table row 0); `atomic_table_grad=True`: one 64-bit integer
        atomic per element.  Bit-identical accumulators either way."""
        _hip.call("ebn_embedding_grad_scatter_fixed_atomic" if self.atomic_table_grad else "ebn_embedding_grad_scatter_fixed", _hip.ptr(ids),
                  _hip.ptr(dX), _hip.ptr(self.table_acc), n_tok, self.D, self.V, st, site, ctypes.c_float(p), _hip.ptr(self.range_flag),
                  _hip.stream_handle())

    def _table_grad_kernels(self, nb, N, sparse_table_grads):
        S = _hip.stream_handle
        st = _hip.ptr(self.state)
        if self.train_embedding and not self._planned and not sparse_table_grads:
            if self.exchange is not None or not self.deterministic:
                self.table_grad.zero_()
            site, p = (0, self.p) if self.p > 0 else (-1, 0.0)
            if self.exchange is not None:
                # validation forms: one gradient row per distinct id locally, sent to its owner, owners accumulate
                d_uniq = torch.zeros_like(nb.rows_uniq)
                _hip.call("ebn_embedding_grad_scatter_f32", _hip.ptr(nb.plan.inv), _hip.ptr(nb.dX), _hip.ptr(d_uniq),
                          N * self.T, self.D, d_uniq.shape[0], st, site, ctypes.c_float(p), S())
                self.exchange.scatter_grads(nb.plan, d_uniq, self._local_scatter_add)
            elif self.deterministic:
                self._accumulate_fixed(nb.ids, nb.dX, N * self.T, st, site, p)
                if not self._adam_from_acc:  # data parallel: the fp32 dense gradient is what gets all-reduced
                    _hip.call("ebn_fixed_to_f32", _hip.ptr(self.table_acc), _hip.ptr(self.table_grad), self.table.numel(),
                              _hip.ptr(self.range_flag), S())
            else:
                _hip.call("ebn_embedding_grad_scatter_f32", _hip.ptr(nb.ids), _hip.ptr(nb.dX), _hip.ptr(self.table_grad),
                          N * self.T, self.D, self.V, st, site, ctypes.c_float(p), S())

    def _optimizer_kernels(self, from_acc=None):
        from_acc = self._adam_from_acc if from_acc is None else from_acc
        S = _hip.stream_handle
        st = _hip.ptr(self.state)
        gs = ctypes.c_float(1.0 / self.world)
        P = self.params
        if not self._adam_done_in_finish:
            _hip.call("ebn_adam_keras_step_f32", _hip.ptr(P.data), _hip.ptr(P.grad), _hip.ptr(P.m), _hip.ptr(P.v), P.numel,
                      st, BETA1, BETA2, ADAM_EPS, gs, S())
        if self.train_embedding and from_acc:
            # the gradient goes from the fixed-point accumulator into Adam in a single sweep over the table
            _hip.call("ebn_adam_keras_step_fixed_f32", _hip.ptr(self.table), _hip.ptr(self.table_acc), _hip.ptr(self.table_m),
                      _hip.ptr(self.table_v), self.table.numel(), st, BETA1, BETA2, ADAM_EPS, gs, _hip.ptr(self.range_flag), S())
        elif self.train_embedding:
            _hip.call("ebn_adam_keras_step_f32", _hip.ptr(self.table), _hip.ptr(self.table_grad), _hip.ptr(self.table_m),
                      _hip.ptr(self.table_v), self.table.numel(), st, BETA1, BETA2, ADAM_EPS, gs, S())
