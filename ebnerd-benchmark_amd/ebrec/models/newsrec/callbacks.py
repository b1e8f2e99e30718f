"""Keras-compatible training callbacks and the streaming AUC metric for the MI355X host layer.

The reference driver wires tf.keras callbacks into ``model.fit`` (ebnerd_nrms.py:212-237):
EarlyStopping / ModelCheckpoint / ReduceLROnPlateau on ``val_auc`` and TensorBoard.  These are
host-side control logic (no device math) with the same constructor arguments and the same
monitor/mode/patience semantics [KERAS-SEMANTICS].
"""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np


class Callback:
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None): ...
    def on_train_end(self, logs=None): ...
    def on_epoch_begin(self, epoch, logs=None): ...
    def on_epoch_end(self, epoch, logs=None): ...
    def on_train_batch_end(self, batch, logs=None): ...


def _improved(mode, monitor):
    if mode == "auto":
        mode = "max" if ("auc" in monitor or "acc" in monitor) else "min"
    if mode == "max":
        return (lambda cur, best, delta=0.0: cur - delta > best), -np.inf
    return (lambda cur, best, delta=0.0: cur + delta < best), np.inf


class History(Callback):
    def on_train_begin(self, logs=None):
        self.history = {}
        self.epoch = []

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto", baseline=None,
                 restore_best_weights=False, start_from_epoch=0):
        self.monitor, self.min_delta, self.patience, self.verbose = monitor, abs(min_delta), patience, verbose
        self.restore_best_weights, self.start_from_epoch, self.baseline = restore_best_weights, start_from_epoch, baseline
        self._better, self._init = _improved(mode, monitor)

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch, self.best, self.best_weights, self.best_epoch = 0, 0, self._init, None, 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None or epoch < self.start_from_epoch:
            return
        if self.restore_best_weights and self.best_weights is None:
            self.best_weights = self.model.get_weights()
        self.wait += 1
        if self._better(cur, self.best, self.min_delta):
            self.best, self.best_epoch = cur, epoch
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
            if self.baseline is None or self._better(cur, self.baseline, self.min_delta):
                self.wait = 0
            return
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self.best_weights is not None:
                if self.verbose:
                    print(f"Restoring model weights from the end of the best epoch: {self.best_epoch + 1}.")
                self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose:
            print(f"Epoch {self.stopped_epoch + 1}: early stopping")


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", save_freq="epoch"):
        self.filepath, self.monitor, self.verbose = str(filepath), monitor, verbose
        self.save_best_only, self.save_weights_only = save_best_only, save_weights_only
        self._better, self.best = _improved(mode, monitor)

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                return
            if not self._better(cur, self.best):
                if self.verbose:
                    print(f"\nEpoch {epoch + 1}: {self.monitor} did not improve from {self.best:.5f}")
                return
            if self.verbose:
                print(f"\nEpoch {epoch + 1}: {self.monitor} improved from {self.best:.5f} to {cur:.5f}, saving model to {path}")
            self.best = cur
        self.model.save_weights(path)


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4,
                 cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0.")
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self._better, self._init = _improved(mode, monitor)

    def on_train_begin(self, logs=None):
        self.best, self.wait, self.cooldown_counter = self._init, 0, 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs["lr"] = self.model.optimizer.learning_rate
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._better(cur, self.best, self.min_delta):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(self.model.optimizer.learning_rate)
                if old > np.float32(self.min_lr):
                    new = max(old * self.factor, self.min_lr)
                    self.model.optimizer.learning_rate = new
                    if self.verbose:
                        print(f"\nEpoch {epoch + 1}: ReduceLROnPlateau reducing learning rate to {new}.")
                    self.cooldown_counter, self.wait = self.cooldown, 0


class TensorBoard(Callback):
    """Scalar logging: torch's SummaryWriter when tensorboard is importable, else JSON lines."""

    def __init__(self, log_dir="logs", histogram_freq=0, **_):
        self.log_dir, self.histogram_freq = str(log_dir), histogram_freq
        self._writer = None

    @staticmethod
    def _rank0():
        import torch.distributed as dist

        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def on_train_begin(self, logs=None):
        self._jsonl = None
        if not self._rank0():  # data parallel: the logs are global (all-reduced), one writer is enough
            return
        Path(self.log_dir).mkdir(parents=True, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter

            self._writer = SummaryWriter(self.log_dir)
        except Exception:
            self._writer = None
        self._jsonl = open(os.path.join(self.log_dir, "scalars.jsonl"), "a")

    def on_epoch_end(self, epoch, logs=None):
        if self._jsonl is None:
            return
        logs = {k: float(v) for k, v in (logs or {}).items()}
        self._jsonl.write(json.dumps({"epoch": epoch, "time": time.time(), **logs}) + "\n")
        self._jsonl.flush()
        if self._writer is not None:
            for k, v in logs.items():
                self._writer.add_scalar(f"epoch_{k}", v, epoch)

    def on_train_end(self, logs=None):
        if self._writer is not None:
            self._writer.close()
        if self._jsonl is not None:
            self._jsonl.close()


class StreamingAUC:
    """[KERAS-SEMANTICS] tf.keras.metrics.AUC defaults: 200 thresholds, ROC curve, 'interpolation'
    (trapezoid) summation over ALL flattened (label, prediction) pairs of an epoch -- the
    ``auc`` / ``val_auc`` that ``compile(metrics=["AUC"])`` reports (ebnerd_nrms.py:244-248), not the
    per-impression sklearn AUC of the evaluator."""

    def __init__(self, num_thresholds=200):
        self.num_thresholds = num_thresholds
        eps = 1e-7
        inner = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
        self.thresholds = np.array([0.0 - eps] + inner + [1.0 + eps], dtype=np.float64)
        self.reset()

    def reset(self):
        self.pos_hist = np.zeros(self.num_thresholds + 1, dtype=np.float64)
        self.neg_hist = np.zeros(self.num_thresholds + 1, dtype=np.float64)
        self._dev = None

    def update_numpy(self, y_true, y_pred):
        y = np.asarray(y_true, dtype=np.float64).reshape(-1)
        p = np.asarray(y_pred, dtype=np.float32).astype(np.float64).reshape(-1)
        k = np.searchsorted(self.thresholds, p, side="left")  # number of thresholds strictly below p
        self.pos_hist += np.bincount(k, weights=(y > 0).astype(np.float64), minlength=self.num_thresholds + 1)
        self.neg_hist += np.bincount(k, weights=(y <= 0).astype(np.float64), minlength=self.num_thresholds + 1)

    def update_device(self, y_true, y_pred):
        """Same accumulation on the device with no host sync: one launch of ebn_auc_hist_f32 for fp32 device tensors (the
        training / validation loops), torch ops otherwise."""
        import torch

        dev = y_pred.device
        if self._dev is None:
            self._dev = (torch.from_numpy(self.thresholds).to(dev), torch.zeros(self.num_thresholds + 1, dtype=torch.int64, device=dev),
                         torch.zeros(self.num_thresholds + 1, dtype=torch.int64, device=dev))
        thr, ph, nh = self._dev
        if y_pred.is_cuda and y_pred.dtype == torch.float32 and y_true.dtype == torch.float32 and y_true.is_cuda and \
                y_pred.is_contiguous() and y_true.is_contiguous():
            from ebrec import _hip

            _hip.call("ebn_auc_hist_f32", _hip.ptr(y_pred), _hip.ptr(y_true), y_pred.numel(), _hip.ptr(thr), self.num_thresholds,
                      _hip.ptr(ph), _hip.ptr(nh), _hip.stream_handle())
            return
        p = y_pred.reshape(-1).to(torch.float64)
        y = y_true.reshape(-1)
        k = torch.searchsorted(thr, p, right=False)
        pos = (y > 0).to(torch.int64)
        ph.index_add_(0, k, pos)
        nh.index_add_(0, k, 1 - pos)

    def result(self, engine=None) -> float:
        """AUC of everything accumulated; with a data-parallel `engine` (world > 1) the histograms of all ranks are summed
        first, so every rank reports the same global value (callbacks then act identically on every rank)."""
        pos, neg = self.pos_hist.copy(), self.neg_hist.copy()
        if self._dev is not None:
            pos += self._dev[1].cpu().numpy().astype(np.float64)
            neg += self._dev[2].cpu().numpy().astype(np.float64)
        if engine is not None and int(getattr(engine, "world", 1)) > 1:
            import torch

            t = torch.from_numpy(np.concatenate([pos, neg])).to(engine.device)
            torch.distributed.all_reduce(t, group=getattr(engine, "pg", None))
            both = t.cpu().numpy()
            pos, neg = both[: pos.size], both[pos.size:]
        # prediction > threshold[i]  <=>  bucket index k > i
        tp = pos.sum() - np.cumsum(pos)[:-1]
        fp = neg.sum() - np.cumsum(neg)[:-1]
        fn, tn = pos.sum() - tp, neg.sum() - fp
        tpr = np.divide(tp, tp + fn, out=np.zeros_like(tp), where=(tp + fn) > 0)
        fpr = np.divide(fp, fp + tn, out=np.zeros_like(fp), where=(fp + tn) > 0)
        return float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0))
