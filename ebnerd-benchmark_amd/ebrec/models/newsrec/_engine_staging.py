"""Batch staging of the NRMS device engine: the loader's article-row numbers and labels travel to the
device in ONE asynchronous copy out of a pinned, double-buffered area and are unpacked -- and the step state advanced -- by one
kernel; token ids are expanded on the device from the resident article matrix (dataloader.py:169-179).  A mixin of NRMSEngine
(_engine.py): it only touches the engine's buffers and the C ABI."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ebrec import _hip

BETA1, BETA2 = 0.9, 0.999  # tf.keras.optimizers.Adam defaults (nrms.py:77): the staging launch advances the step state


class StagingMixin:
    # ------------------------------------------------------------------ training
    # ------------------------------------------------------------------ device-side batch assembly (a13)
    def set_article_matrix(self, matrix) -> None:
        """Keep the loader's (n_articles+1, T) token matrix in HBM; batches can then be given as article-row
        numbers (``train_step(..., indexed=True)``) and expanded to token ids on the device."""
        m = np.asarray(matrix)
        if m.ndim != 2 or m.shape[1] != self.T or not np.issubdtype(m.dtype, np.integer):
            raise ValueError(f"article matrix must be integer (n_articles+1, {self.T}), got {m.dtype} {m.shape}")
        if m.size and (m.min() < 0 or m.max() >= self.V):
            raise IndexError(f"token id out of range [0, {self.V}) for the embedding table")
        self.article_matrix = torch.from_numpy(np.ascontiguousarray(m.astype(np.int32))).to(self.device)
        self._article_matrix_src = matrix

    def _stage_host(self, int_arrays, dst_int: torch.Tensor, y, dst_lab: torch.Tensor) -> None:
        """Host integers (token ids or article-row numbers) + labels -> device in ONE asynchronous copy out of a pinned,
        double-buffered staging area; the kernel that unpacks it into `dst_int` / `dst_lab` also advances the step state.
        The host never waits for the GPU and runs up to two steps ahead."""
        flat = [np.asarray(a).reshape(-1) for a in int_arrays]
        n_int = sum(a.size for a in flat)
        lab = np.asarray(y, dtype=np.float32).reshape(-1)
        n = n_int + lab.size
        st = getattr(self, "_host_stage", None)
        if st is None or st["pinned"][0].numel() < n:
            st = self._host_stage = {"pinned": [torch.empty(2 * n, dtype=torch.int32).pin_memory() for _ in range(2)],
                                     "dev": torch.empty(2 * n, dtype=torch.int32, device=self.device), "ev": [None, None], "k": 0}
        k = st["k"] = st["k"] ^ 1
        if st["ev"][k] is not None:
            st["ev"][k].synchronize()  # the copy that last read this pinned buffer (two steps ago) has long finished
        hs = st["pinned"][k].numpy()
        off = 0
        for a in flat:
            hs[off: off + a.size] = a
            off += a.size
        hs[n_int:n].view(np.float32)[:] = lab
        st["dev"][:n].copy_(st["pinned"][k][:n], non_blocking=True)
        st["ev"][k] = torch.cuda.Event()
        st["ev"][k].record()
        _hip.call("ebn_copy3_advance", _hip.ptr(st["dev"]), _hip.ptr(dst_int), n_int * 4, None, None, 0, _hip.ptr(st["dev"][n_int:]),
                  _hip.ptr(dst_lab), lab.size * 4, _hip.ptr(self.state), BETA1, BETA2, _hip.stream_handle())

    def _stage_indexed(self, nb, his_idx, pred_idx, y=None):
        """Article-row numbers (+ labels) of a batch -> device, then the token ids are expanded on the device.  Host batches
        (what the loaders hand over) travel as ONE asynchronous copy out of a pinned, double-buffered staging area, and the
        step-state advance rides in the kernel that unpacks it: the host never waits for the GPU and runs a step ahead.
        Returns (y still to be uploaded or None, whether the step state has been advanced)."""
        B, C = his_idx.shape[0], pred_idx.shape[1]
        n_titles, n_lab = B * (self.H + C), B * C
        if not hasattr(nb, "art_idx") or nb.art_idx.numel() < n_titles:
            nb.art_idx = torch.empty(nb.n_seq, dtype=torch.int32, device=self.device)
        advanced = False
        host = not isinstance(his_idx, torch.Tensor) and not isinstance(pred_idx, torch.Tensor) and y is not None and \
            not isinstance(y, torch.Tensor)
        if host:
            self._stage_host([his_idx, pred_idx], nb.art_idx, y, nb.labels)
            y, advanced = None, True
        else:
            off = 0
            for a in (his_idx, pred_idx):
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1).astype(np.int32, copy=False)))
                t = t.reshape(-1)
                nb.art_idx[off: off + t.numel()].copy_(t.to(device=self.device, dtype=torch.int32), non_blocking=True)
                off += t.numel()
        _hip.call("ebn_expand_titles_i32", _hip.ptr(nb.art_idx), _hip.ptr(self.article_matrix), _hip.ptr(nb.ids), n_titles,
                  self.T, self.article_matrix.shape[0], _hip.ptr(self.oob_flag), _hip.stream_handle())
        return y, advanced

    def _device_batch(self, his, pred, y) -> bool:
        same_dev = lambda t: t.is_cuda and (self.device.index is None or t.device.index == self.device.index)
        ok = lambda t, dt: isinstance(t, torch.Tensor) and same_dev(t) and t.dtype == dt and t.is_contiguous()
        return ok(his, torch.int32) and ok(pred, torch.int32) and ok(y, torch.float32)

    def _upload_ids(self, dst, *arrays):
        """Token ids -> int32 device buffer; ids outside [0,V) raise like TF-CPU's Embedding does."""
        off = 0
        for a in arrays:
            if isinstance(a, torch.Tensor):
                t = a.reshape(-1).to(device=self.device, dtype=torch.int32)
            else:
                a = np.asarray(a)
                if a.size and (a.min() < 0 or a.max() >= self.V):
                    raise IndexError(f"token id out of range [0, {self.V}) for the embedding table")
                t = torch.from_numpy(np.ascontiguousarray(a.reshape(-1).astype(np.int32, copy=False)))
            dst[off: off + t.numel()].copy_(t, non_blocking=True)
            off += t.numel()
        return off

