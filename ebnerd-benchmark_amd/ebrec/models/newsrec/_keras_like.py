"""The Keras ``Model`` surface the reference's callers use, on top of a device engine.

Callers of the reference touch ``model.model.fit / predict / compile / summary / save_weights /
load_weights / count_params / variables / optimizer / loss`` and ``model.scorer.predict``
(ebnerd_nrms.py:244-260,302; nrms_dummy.py:46-47).  This module keeps
those call shapes; the arithmetic happens in the engine's HIP kernels.
"""
from __future__ import annotations

import time
from types import SimpleNamespace

import numpy as np
import torch

from ._engine import eval_loss_from_news
from .callbacks import Callback, History, StreamingAUC


def _dist_of(eng):
    """(rank, world, group) of the engine's data-parallel group; (0, 1, None) for a single process."""
    world = int(getattr(eng, "world", 1))
    if world <= 1:
        return 0, 1, None
    group = getattr(eng, "pg", None)
    return torch.distributed.get_rank(group), world, group


def _allreduce_host(values, eng, op="sum"):
    """All-reduce a small list of host floats over the engine's group (device round trip: RCCL needs device buffers).
    Every rank gets the same result, so callbacks driven by the reduced logs take the same decisions everywhere."""
    _, world, group = _dist_of(eng)
    if world == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=eng.device)
    torch.distributed.all_reduce(t, op={"sum": torch.distributed.ReduceOp.SUM, "min": torch.distributed.ReduceOp.MIN}[op], group=group)
    return t.cpu().tolist()


def _enter_collective(eng, what: str) -> None:
    """world > 1: raise (instead of hanging in RCCL) when `what` is entered by only some ranks of the group."""
    guard = getattr(eng, "guard", None)
    if guard is not None:
        guard.enter(what)


def _sync_moving_statistics(eng) -> None:
    fn = getattr(eng, "sync_moving_statistics", None)
    if fn is not None and int(getattr(eng, "world", 1)) > 1:
        fn()


def _full_batches(data) -> int:
    """number of leading batches of `data` that have the full batch size (all but possibly the last one)"""
    bs = getattr(data, "batch_size", None) or getattr(data, "bs", None)
    rows = getattr(data, "X", None)
    rows = rows if rows is not None else getattr(data, "his", None)
    if bs is None or rows is None:
        return len(data)
    return len(rows) // int(bs)


def _is_loader(x):
    return hasattr(x, "__len__") and hasattr(x, "__getitem__") and not isinstance(x, (tuple, list, np.ndarray))


class _ArrayBatches:
    """(his, pred), y arrays sliced like a Keras Sequence."""

    def __init__(self, x, y, batch_size):
        self.his, self.pred = x
        self.y = y
        self.bs = int(batch_size or 32)

    def __len__(self):
        return int(np.ceil(len(self.his) / self.bs))

    def __getitem__(self, i):
        s = slice(i * self.bs, (i + 1) * self.bs)
        return (self.his[s], self.pred[s]), (None if self.y is None else self.y[s])


class _Optimizer:
    """``model.optimizer``: exposes learning_rate like tf.keras.optimizers.Adam (nrms.py:77)."""

    def __init__(self, engine, name="adam"):
        self._engine, self.name = engine, name

    @property
    def learning_rate(self):
        return self._engine.learning_rate

    @learning_rate.setter
    def learning_rate(self, lr):
        self._engine.learning_rate = lr

    lr = learning_rate

    def get_config(self):
        return {"name": "Adam", "learning_rate": self.learning_rate, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7}


class _Variable(SimpleNamespace):
    pass


class TrainModel:
    """``NRMSModel.model``: inputs [his (B,H,T), pred (B,C,T)] -> softmax probabilities (B,C)."""

    cache_articles = True  # evaluate(): encode the loader's article matrix once (set False to force per-batch encoding)

    def __init__(self, owner, names):
        self._owner = owner
        self._names = names
        self.stop_training = False
        self.metrics_names = []
        self.history = None

    @property
    def _engine(self):
        return self._owner._engine

    # ---- compile / introspection -------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, **_):
        """Re-compile (ebnerd_nrms.py:244-248): optimizer/loss objects coming from this model's own
        getters are accepted as-is; a loss given by name switches the engine's loss."""
        if isinstance(loss, str):
            loss = {"categorical_crossentropy": "cross_entropy_loss", "binary_crossentropy": "log_loss"}.get(loss, loss)
            self._owner._set_loss(loss)
        if isinstance(optimizer, str) and optimizer.lower() != "adam":
            raise ValueError(f"this optimizer not defined {optimizer}")
        self.metrics_names = [str(m).lower() for m in (metrics or [])]

    @property
    def optimizer(self):
        return _Optimizer(self._engine)

    @property
    def loss(self):
        return {"cross_entropy_loss": "categorical_crossentropy", "log_loss": "binary_crossentropy"}[self._engine.loss]

    def count_params(self):
        return self._engine.count_params()

    @property
    def variables(self):
        dev = str(self._engine.device)
        return [_Variable(name=n, device=dev, shape=w.shape) for n, w in zip(self._names, self._engine.get_weights())]

    trainable_variables = variables

    def summary(self, print_fn=print):
        e = self._engine
        print_fn(f'Model: "{type(self._owner).__name__}" on {e.device} (HIP/gfx950 kernels)')
        print_fn(f"{'variable':<28}{'shape':<20}{'params':>12}")
        for n, w in zip(self._names, e.get_weights()):
            print_fn(f"{n:<28}{str(tuple(w.shape)):<20}{w.size:>12,}")
        print_fn(f"Total params: {e.count_params():,}")

    def get_weights(self):
        return self._engine.get_weights()

    def set_weights(self, weights):
        self._engine.set_weights(weights)

    def save_weights(self, filepath, **_):
        """Named tensors (SURVEY.md A.6 order) in a torch file at exactly `filepath`.
        With world > 1 this is a COLLECTIVE: every rank must call it (rank 0 writes; the others take part in the table
        all-gather of a row-sharded engine and in the closing barrier).  Calling it on rank 0 only RAISES after
        EBN_COLLECTIVE_TIMEOUT_S (`_dist.LockStepGuard`) instead of hanging."""
        rank, world, group = _dist_of(self._engine)
        _enter_collective(self._engine, "model.save_weights()")
        _sync_moving_statistics(self._engine)  # BatchNormalization moving averages: the checkpoint carries the mean over the ranks
        # get_weights() is a collective when the table is row-sharded: every rank calls it, rank 0 writes the file
        state = {n: torch.from_numpy(np.ascontiguousarray(w)) for n, w in zip(self._names, self._engine.get_weights())}
        extra = getattr(self._engine, "extra_state", lambda: {})()
        if rank == 0:
            torch.save({"format": "ebnerd-mi355x-weights-v1", "weights": state, "extra": extra}, str(filepath))
        if world > 1:
            torch.distributed.barrier(group=group)  # the file exists on return, on every rank

    def load_weights(self, filepath, **_):
        blob = torch.load(str(filepath), map_location="cpu", weights_only=True)
        if not isinstance(blob, dict) or blob.get("format") != "ebnerd-mi355x-weights-v1":
            raise ValueError(f"{filepath} is not an ebnerd-mi355x weight file")
        self._engine.set_weights([blob["weights"][n].numpy() for n in self._names])
        if blob.get("extra") and hasattr(self._engine, "load_extra_state"):
            self._engine.load_extra_state(blob["extra"])

    # ---- training ------------------------------------------------------------------
    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, initial_epoch=0, **_):
        data = x if _is_loader(x) else _ArrayBatches(x, y, batch_size)
        val = None
        if validation_data is not None:
            val = validation_data if _is_loader(validation_data) else _ArrayBatches(validation_data[0], validation_data[1], batch_size)
        hist = History()
        cbs = [hist] + list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
        self.stop_training = False
        want_auc = "auc" in self.metrics_names
        eng = self._engine
        _enter_collective(eng, "model.fit()")
        rng = np.random.default_rng(self._owner.seed)
        rank, world, _group = _dist_of(eng)
        # data parallel: every step ends in a gradient all-reduce, so every rank must run the SAME number of steps per
        # epoch -- the shortest shard decides (the surplus batches of longer shards rotate in through the shuffle)
        n_batches = len(data)
        if world > 1 and hasattr(eng, "pin_table_grad_exchange") and n_batches:
            # table_grad_exchange="auto" is decided ONCE, here, from the FULL-batch shape (configuration: the same on every rank) --
            # never per step from a rank's local, possibly short, batch
            # (round-4 ADVICE: a rank whose loader has no batch_size and whose first batch is short must not pin a different
            # collective than its peers -> the count is MAX-reduced over the group; a later fit() with another batch size must
            # not keep a stale verdict -> the pin is reset here)
            (h0, p0), _y0 = data.index_batch(0) if hasattr(data, "index_batch") and not getattr(data, "eval_mode", False) else data[0]
            bs = int(getattr(data, "batch_size", None) or getattr(data, "bs", None) or len(h0))
            n_tok_full = bs * (np.shape(h0)[1] + np.shape(p0)[1]) * eng.T
            t = torch.tensor([float(n_tok_full)], dtype=torch.float64, device=eng.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=_group)
            before, eng._sparse_pin = eng._sparse_pin, None
            if eng.pin_table_grad_exchange(int(t.item())) != before and before is not None:
                eng._graphs.clear()  # the step's collectives changed: captured segment lists are stale
        if world > 1 and getattr(eng, "needs_equal_batches", getattr(eng, "exchange", None) is not None):
            # row-sharded table / sparse table-gradient exchange: the step's collectives are sized by the batch SHAPE, which
            # must therefore be the same on every rank in every step -- a shard's short last batch is left out (it is always
            # the last index).  Shards of unequal length would otherwise put one rank's short batch next to a full one.
            n_batches = _full_batches(data)
        n_steps = int(_allreduce_host([n_batches], eng, "min")[0]) if world > 1 else n_batches
        for cb in cbs:
            cb.on_train_begin()
        for epoch in range(initial_epoch, epochs):
            for cb in cbs:
                cb.on_epoch_begin(epoch)
            t0 = time.time()
            order = rng.permutation(n_batches) if shuffle else np.arange(n_batches)  # Keras shuffles batch ORDER only
            order = order[:n_steps]
            loss_sum = torch.zeros(1, device=eng.device)
            n_rows = 0
            auc = StreamingAUC() if want_auc else None
            # loaders of this repo + an engine that can hold the token matrix in HBM: ship article-row numbers only
            indexed = hasattr(data, "index_batch") and hasattr(eng, "set_article_matrix") and not getattr(data, "eval_mode", False)
            if indexed and getattr(eng, "_article_matrix_src", None) is not data.lookup_article_matrix:
                try:
                    eng.set_article_matrix(data.lookup_article_matrix)
                except ValueError:  # not this engine's kind of matrix (token ids vs document vectors): host-gathered batches
                    indexed = False
            for step, idx in enumerate(order):
                (his, pred), yb = data.index_batch(int(idx)) if indexed else data[int(idx)]
                nb = len(his)
                if want_auc:
                    loss, probs, labels_dev = eng.train_step(his, pred, yb, return_probs=True, **({"indexed": True} if indexed else {}))
                    auc.update_device(labels_dev, probs)
                else:
                    loss = eng.train_step(his, pred, yb, **({"indexed": True} if indexed else {}))
                loss_sum += loss * nb
                n_rows += nb
                for cb in cbs:
                    cb.on_train_batch_end(step)
            ls, nr = _allreduce_host([float(loss_sum.item()), n_rows], eng)  # epoch logs are GLOBAL: identical on every rank
            logs = {"loss": ls / max(nr, 1)}
            if hasattr(eng, "check_oob"):
                eng.check_oob()  # ids outside the table raise (as TF-CPU's Embedding does) at the epoch's one host sync
            _sync_moving_statistics(eng)  # data parallel: once per epoch the replicas' BatchNormalization moving averages are averaged
            if want_auc:
                logs["auc"] = auc.result(eng)
            if val is not None:
                vl = self.evaluate(val, verbose=0, return_dict=True)
                logs.update({f"val_{k}": v for k, v in vl.items()})
            if hasattr(data, "on_epoch_end"):
                data.on_epoch_end()
            if verbose:
                dt = time.time() - t0
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in logs.items())
                if rank == 0:
                    print(f"Epoch {epoch + 1}/{epochs} - {len(order)} steps - {dt:.1f}s - {nr / max(dt, 1e-9):.0f} impressions/s - {msg}")
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end()
        self.history = hist
        return hist

    def evaluate(self, x=None, y=None, batch_size=None, verbose=0, return_dict=False, **_):
        """With world > 1 this is a COLLECTIVE: every rank evaluates ITS shard and must call this at the same point; loss,
        AUC histograms and the device error flags are reduced over the group, so the result (and any raised error) is the
        same on every rank.  Entered by only some ranks it raises after EBN_COLLECTIVE_TIMEOUT_S instead of hanging.  For a
        rank-local evaluation build the model without a process group."""
        data = x if _is_loader(x) else _ArrayBatches(x, y, batch_size)
        eng = self._engine
        _enter_collective(eng, "model.evaluate()")
        _sync_moving_statistics(eng)  # every rank scores with the same BatchNormalization moving averages
        auc = StreamingAUC() if "auc" in self.metrics_names else None
        loss_sum, n_rows = torch.zeros(1, device=eng.device), 0
        # loaders of this repo: encode every article of the lookup matrix ONCE (the weights are fixed during evaluate) and
        # run each batch from the cached news vectors -- the validation pass of fit() costs user encoders only
        cached = (self.cache_articles and hasattr(data, "index_batch") and hasattr(self._owner, "_encode_article_matrix")
                  and not getattr(data, "eval_mode", False))
        lockstep = int(getattr(eng, "world", 1)) > 1 and hasattr(eng, "defer_flag_checks")
        if lockstep:  # the inference calls below must not raise on one rank alone: the closing check_oob() reports for everyone
            eng.defer_flag_checks = True
        try:
            news_all = self._owner._encode_article_matrix(data.lookup_article_matrix) if cached else None
            for i in range(len(data)):
                if cached:
                    (his, pred), yb = data.index_batch(i)
                    loss, probs = eval_loss_from_news(eng, news_all, his, pred, yb)
                else:
                    (his, pred), yb = data[i]
                    loss, probs = eng.eval_loss(his, pred, yb)
                loss_sum += loss * len(his)  # stays on the device: one host sync per evaluate(), not per batch
                n_rows += len(his)
                if auc is not None:
                    auc.update_device(torch.as_tensor(np.asarray(yb, dtype=np.float32)).to(eng.device).reshape(probs.shape).contiguous(), probs)
        finally:
            if lockstep:
                eng.defer_flag_checks = False
        ls, nr = _allreduce_host([float(loss_sum.item()), n_rows], eng)  # every rank evaluates its shard; the result is global
        out = {"loss": ls / max(nr, 1)}
        if hasattr(eng, "check_oob"):
            eng.check_oob()
        if hasattr(eng, "l2_penalty"):
            out["loss"] += eng.l2_penalty()  # Keras adds the kernel_regularizer terms to the evaluated loss too
        if auc is not None:
            out["auc"] = auc.result(eng)
        return out if return_dict else ([out["loss"]] + ([out["auc"]] if auc is not None else []))

    def predict(self, x, batch_size=None, verbose=0, **_):
        data = x if _is_loader(x) else _ArrayBatches(x, None, batch_size)
        outs = []
        for i in range(len(data)):
            (his, pred), _y = data[i]
            probs, _ = self._engine.forward(his, pred, mode="softmax")
            outs.append(probs.cpu().numpy())
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, 0), np.float32)

    def __call__(self, inputs, training=False):
        his, pred = inputs
        return self._engine.forward(his, pred, mode="softmax")[0]


class ScorerModel:
    """``NRMSModel.scorer``: inputs [his (N,H,T), pred_one (N,1,T)] -> sigmoid(u.n) (N,1)
    (nrms.py:204-208).  The eval loader repeats the history once per candidate
    (dataloader.py:99-103); identical history rows of a batch are encoded ONCE here and the
    scores come from a ragged pair-dot kernel -- same outputs, ~C_i x fewer user encodings."""

    cache_articles = True  # encode the loader's article matrix once per predict() (set False to force per-batch encoding)

    def __init__(self, owner):
        self._owner = owner

    def predict(self, x, batch_size=None, verbose=0, **_):
        eng = self._owner._engine
        data = x if _is_loader(x) else _ArrayBatches(x, None, batch_size)
        outs = []
        compact = getattr(data, "compact_eval_batch", None) if getattr(data, "eval_mode", False) else None
        indexed = getattr(data, "index_eval_batch", None) if compact is not None else None
        if indexed is not None and self.cache_articles and hasattr(self._owner, "_score_indexed"):
            # every article of the loader's matrix is encoded ONCE (the weights do not change during predict); a batch
            # then costs one user-encoder pass over gathered news vectors and a ragged pair-dot
            news_all = self._owner._encode_article_matrix(data.lookup_article_matrix)
            for i in range(len(data)):
                his_idx, cand_idx, rows, _y = indexed(i)
                outs.append(self._owner._score_indexed(news_all, his_idx, cand_idx, rows).cpu().numpy().reshape(-1, 1))
            return np.concatenate(outs, axis=0) if outs else np.zeros((0, 1), np.float32)
        for i in range(len(data)):
            if compact is not None:  # this repo's loaders: history rows are NOT materialised per candidate
                his, cand, rows, _y = compact(i)
                s = self._owner._score_compact(np.asarray(his), np.asarray(cand), np.asarray(rows))
            else:
                (his, pred_one), _y = data[i]
                s = self._owner._score_pairs(np.asarray(his), np.asarray(pred_one))
            outs.append(s.cpu().numpy().reshape(-1, 1))
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, 1), np.float32)

    def __call__(self, inputs, training=False):
        his, pred_one = inputs
        return self._owner._score_pairs(np.asarray(his), np.asarray(pred_one)).view(-1, 1)


class EncoderModel:
    """``.newsencoder`` / ``.userencoder`` sub-models (nrms.py:192-193)."""

    def __init__(self, fn, name):
        self._fn, self.name = fn, name

    def predict(self, x, batch_size=None, verbose=0, **_):
        return self._fn(x).cpu().numpy()

    def __call__(self, x, training=False):
        return self._fn(x)


def dedup_rows(a: np.ndarray):
    """unique rows + inverse index (host side; rows are short int/float vectors)."""
    flat = np.ascontiguousarray(a.reshape(a.shape[0], -1))
    view = flat.view(np.dtype((np.void, flat.dtype.itemsize * flat.shape[1]))).reshape(-1)
    _, first, inv = np.unique(view, return_index=True, return_inverse=True)
    return a[first], inv.reshape(-1).astype(np.int32)
