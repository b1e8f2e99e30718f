"""``[Dense(u, relu, l2) -> BatchNormalization -> Dropout] x len(units)`` over a block of rows that belongs to TWO
TimeDistributed call sites (history rows first, candidate rows after) -- the news encoder of NRMSDocVec
(nrms_docvec.py:116-124) and the optional per-token stack of the NRMS news encoder (nrms.py:143-152).

Dense layers run over the whole block in one GEMM; BatchNormalization runs per call site with its own batch
statistics and moving-average update [KERAS-SEMANTICS]; the dropout stream is indexed by (row, column) of the whole
block.  Parameters live in the engine's FlatParams under ``<prefix>d{l}_W / d{l}_b / bn{l}_g / bn{l}_b``; the moving
statistics are plain tensors owned by this object.
"""
from __future__ import annotations

import ctypes

import torch

from ebrec import _hip

SITE_MLP0 = 8
STRIP_ROWS = 1024  # rows of BOTH call sites together the single-launch two-site BatchNorm kernels take (csrc/ebn_dense.hip)


class MLPStack:
    def __init__(self, params, prefix: str, din: int, units, device, l2: float, on_realloc=None):
        self.on_realloc = on_realloc  # engines drop their captured hipGraphs when the activation buffers move
        self.params, self.prefix, self.din, self.units, self.device, self.l2 = params, prefix, int(din), [int(u) for u in units], device, float(l2)
        if len(self.units) > 4:
            raise ValueError("at most 4 hidden Dense layers (dropout sites 8..11 of ebn_step_state)")
        # BatchNormalization moving statistics: ONE flat buffer (mean_0 | var_0 | mean_1 | ...; 256-byte aligned segments) so that the
        # data-parallel synchronisation below is a single collective; bn_mean[l] / bn_var[l] are views the kernels update in place
        off, spans = 0, []
        for u in self.units:
            spans.append((off, off + (-(-u // 64) * 64)))
            off = spans[-1][1] + (-(-u // 64) * 64)
        self.bn_stats = torch.zeros(max(off, 1), device=device)
        self.bn_mean = [self.bn_stats[a: a + u] for (a, _b), u in zip(spans, self.units)]
        self.bn_var = [self.bn_stats[b: b + u] for (_a, b), u in zip(spans, self.units)]
        for v in self.bn_var:
            v.fill_(1.0)
        self._b = None

    @staticmethod
    def shapes(prefix: str, din: int, units) -> dict:
        out, prev = {}, din
        for l, u in enumerate(units):
            out.update({f"{prefix}d{l}_W": (prev, u), f"{prefix}d{l}_b": (u,), f"{prefix}bn{l}_g": (u,), f"{prefix}bn{l}_b": (u,)})
            prev = u
        return out

    @property
    def out_dim(self):
        return self.units[-1] if self.units else self.din

    def _pv(self, name):
        return self.params.view(self.prefix + name)

    def _g(self, name):
        return self.params.g(self.prefix + name)

    def bufs(self, N):
        b = self._b
        if b is None or b["N"] < N:
            f = lambda *s: torch.empty(*s, device=self.device)
            b = {"N": N, "R": [f(N, u) for u in self.units], "xhat": [f(N, u) for u in self.units],
                 "Xn": [f(N, u) for u in self.units], "mean": [[f(u), f(u)] for u in self.units],
                 "istd": [[f(u), f(u)] for u in self.units], "dA": [f(N, u) for u in self.units],
                 "dB": [f(N, u) for u in self.units], "dX0": f(N, self.din)}
            width = max(self.units + [1])
            b["partials"] = f(int(_hip.lib().ebn_colsum_partials_len(N, width)))
            dims = [self.din] + self.units
            wsf = _hip.lib().ebn_gemm_workspace_floats
            ws = max([int(wsf(dims[i], dims[i + 1], N)) for i in range(len(self.units))] +
                     [int(wsf(N, dims[i + 1], dims[i])) for i in range(len(self.units))] +
                     [int(wsf(N, dims[i], dims[i + 1])) for i in range(len(self.units))] + [1])
            b["ws"] = f(ws)
            self._b = b
            if self.on_realloc is not None:
                self.on_realloc()
        return b

    @staticmethod
    def gemm(tA, tB, M, N, K, A, lda, B, ldb, beta, C, ldc, ws=None):
        _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), lda, _hip.ptr(B), ldb,
                  ctypes.c_float(beta), _hip.ptr(C), ldc, _hip.ptr(ws), 0 if ws is None else ws.numel(), _hip.stream_handle())

    def forward(self, x, n0: int, n1: int, train: bool, state, p: float):
        """x (n0+n1, din): rows [0,n0) = first call site, [n0,n0+n1) = second.  Returns the (N, out_dim) output."""
        N, S, b = n0 + n1, _hip.stream_handle, self.bufs(n0 + n1)
        st = _hip.ptr(state) if train else None
        prev = self.din
        for l, u in enumerate(self.units):
            R = b["R"][l]
            _hip.call("ebn_dense_relu_fwd_f32", N, u, prev, _hip.ptr(x), prev, _hip.ptr(self._pv(f"d{l}_W")), u,
                      _hip.ptr(self._pv(f"d{l}_b")), _hip.ptr(R), u, _hip.ptr(b["ws"]), b["ws"].numel(), S())
            if train and n0 + n1 <= STRIP_ROWS:  # both call sites in one launch
                _hip.call("ebn_batchnorm2_fwd_f32", _hip.ptr(R), _hip.ptr(self._pv(f"bn{l}_g")), _hip.ptr(self._pv(f"bn{l}_b")),
                          _hip.ptr(self.bn_mean[l]), _hip.ptr(self.bn_var[l]), _hip.ptr(b["Xn"][l]), _hip.ptr(b["xhat"][l]),
                          _hip.ptr(b["mean"][l][0]), _hip.ptr(b["istd"][l][0]), _hip.ptr(b["mean"][l][1]), _hip.ptr(b["istd"][l][1]),
                          n0, n1, u, st, SITE_MLP0 + l, ctypes.c_float(p), S())
            else:
                for site, (r0, nr) in enumerate(((0, n0), (n0, n1))):
                    if nr == 0:
                        continue
                    _hip.call("ebn_batchnorm_fwd_f32", _hip.ptr(R[r0:]), _hip.ptr(self._pv(f"bn{l}_g")), _hip.ptr(self._pv(f"bn{l}_b")),
                              _hip.ptr(self.bn_mean[l]), _hip.ptr(self.bn_var[l]), _hip.ptr(b["Xn"][l][r0:]),
                              _hip.ptr(b["xhat"][l][r0:]), _hip.ptr(b["mean"][l][site]), _hip.ptr(b["istd"][l][site]),
                              _hip.ptr(b["partials"]), nr, u, 1 if train else 0, st, SITE_MLP0 + l,
                              ctypes.c_float(p if train else 0.0), ctypes.c_int64(r0 * u), S())
            x, prev = b["Xn"][l], u
        return x

    def backward(self, d_last, x0, n0: int, n1: int, state, p: float, need_dx0: bool, loss_dev=None):
        """d_last (N, out_dim) is consumed.  Parameter gradients are OVERWRITTEN in the FlatParams grad buffer; the L2
        penalty lambda*sum(W^2) of every regularised kernel is added to loss_dev[0]; returns d(x0) (N, din) when need_dx0."""
        N, S, b = n0 + n1, _hip.stream_handle, self._b
        st = _hip.ptr(state)
        dY = d_last
        for l in reversed(range(len(self.units))):
            u = self.units[l]
            x_in, din = (b["Xn"][l - 1], self.units[l - 1]) if l else (x0, self.din)
            dR = b["dB"][l]
            if n0 + n1 <= STRIP_ROWS:  # BN backward of both call sites + the ReLU backward / bias gradient of the Dense: one launch
                _hip.call("ebn_batchnorm2_relu_bwd_f32", _hip.ptr(dY), _hip.ptr(b["xhat"][l]), _hip.ptr(b["R"][l]), _hip.ptr(self._pv(f"bn{l}_g")),
                          _hip.ptr(b["istd"][l][0]), _hip.ptr(b["istd"][l][1]), _hip.ptr(dR), _hip.ptr(self._g(f"bn{l}_g")),
                          _hip.ptr(self._g(f"bn{l}_b")), _hip.ptr(self._g(f"d{l}_b")), n0, n1, u, st, SITE_MLP0 + l, ctypes.c_float(p), S())
            else:
                for site, (r0, nr) in enumerate(((0, n0), (n0, n1))):
                    if nr == 0:
                        continue
                    first = (site == 0) or (n0 == 0)
                    _hip.call("ebn_batchnorm_bwd_f32", _hip.ptr(dY[r0:]), _hip.ptr(b["xhat"][l][r0:]), _hip.ptr(self._pv(f"bn{l}_g")),
                              _hip.ptr(b["istd"][l][site]), _hip.ptr(dR[r0:]), _hip.ptr(self._g(f"bn{l}_g")), _hip.ptr(self._g(f"bn{l}_b")),
                              _hip.ptr(b["partials"]), nr, u, 1, 0 if first else 1, st, SITE_MLP0 + l, ctypes.c_float(p),
                              ctypes.c_int64(r0 * u), S())
                _hip.call("ebn_bias_relu_bwd_f32", _hip.ptr(b["R"][l]), _hip.ptr(dR), _hip.ptr(dR), _hip.ptr(self._g(f"d{l}_b")),
                          _hip.ptr(b["partials"]), N, u, 0, S())
            if l or need_dx0:  # dW = X^T.dR and dX = dR.W^T: independent of each other, one launch for the pair
                dx = b["dA"][l - 1] if l else b["dX0"]
                ws = b["ws"]
                _hip.call("ebn_dense_bwd_pair_f32", N, din, u, _hip.ptr(x_in), din, _hip.ptr(dR), u, _hip.ptr(self._pv(f"d{l}_W")), u,
                          ctypes.c_float(0.0), _hip.ptr(self._g(f"d{l}_W")), u, _hip.ptr(dx), din,
                          _hip.ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, S())
                if l:
                    dY = dx
            else:
                self.gemm(1, 0, din, u, N, x_in, din, dR, u, 0.0, self._g(f"d{l}_W"), u, b["ws"])
        if self.l2 > 0 and self.units:
            # kernel_regularizer=l2(lambda) of every Dense kernel of the stack, after all dW are written: gW += 2*lambda*W
            # and loss += lambda*sum(W^2) -- one pass over the weights, two launches for the whole stack
            seg, dims = [], [self.din] + self.units
            for l in range(len(self.units)):
                seg += [_hip.ptr(self._pv(f"d{l}_W")), _hip.ptr(self._g(f"d{l}_W")), dims[l] * dims[l + 1]]
            seg += [None, None, 0] * (4 - len(self.units))
            _hip.call("ebn_l2_reg4_f32", *seg, ctypes.c_float(self.l2), _hip.ptr(b["partials"]), _hip.ptr(loss_dev), S())
        return b["dX0"] if (need_dx0 and self.units) else (d_last if need_dx0 else None)

    def sync_moving_statistics(self, group=None) -> None:
        """Data parallel: replace every rank's BatchNormalization moving mean / variance by their MEAN over the ranks (a collective).

        The contract under data parallel [KERAS-SEMANTICS: tf.keras BatchNormalization is not synchronised across replicas either]:
        each rank normalises ITS half-batch with ITS batch statistics, per TimeDistributed call site (nrms_docvec.py:116-124,
        nrms.py:143-152), the gradients of gamma / beta / the Dense kernels are all-reduced with every other gradient, and the
        moving statistics -- which only inference reads -- drift apart by each rank's data.  Averaging them (each is an exponential
        average of per-batch statistics: the mean over ranks is the same average taken over all ranks' batches) once per epoch and
        in front of every evaluate / checkpoint makes the replicas score identically and rank 0's checkpoint carry everyone's data."""
        import torch.distributed as dist

        if not self.units or not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(self.bn_stats, group=group)
            self.bn_stats.mul_(1.0 / world)

    def l2_penalty(self) -> float:
        """lambda * sum(W^2) of every regularised kernel (host readback; evaluate() only)."""
        if self.l2 <= 0:
            return 0.0
        acc = torch.zeros(1, device=self.device)
        for l in range(len(self.units)):
            w = self._pv(f"d{l}_W")
            _hip.call("ebn_sumsq_f32", _hip.ptr(w), w.numel(), ctypes.c_float(self.l2), _hip.ptr(acc), 1, _hip.stream_handle())
        return float(acc.item())

    # ---- weights in Keras creation order per layer: kernel, bias, gamma, beta, moving_mean, moving_variance
    def weight_names(self, base: str):
        names = []
        for l in range(len(self.units)):
            names += [f"{base}.dense{l}.kernel", f"{base}.dense{l}.bias", f"{base}.bn{l}.gamma", f"{base}.bn{l}.beta",
                      f"{base}.bn{l}.moving_mean", f"{base}.bn{l}.moving_variance"]
        return names

    def get_weights(self):
        out = []
        for l in range(len(self.units)):
            out += [self._pv(f"d{l}_W").cpu().numpy(), self._pv(f"d{l}_b").cpu().numpy(), self._pv(f"bn{l}_g").cpu().numpy(),
                    self._pv(f"bn{l}_b").cpu().numpy(), self.bn_mean[l].cpu().numpy(), self.bn_var[l].cpu().numpy()]
        return out

    def set_weights(self, w):
        """w: list of 6*len(units) torch tensors (already float32)."""
        with torch.no_grad():
            for l in range(len(self.units)):
                k = 6 * l
                self._pv(f"d{l}_W").copy_(w[k]); self._pv(f"d{l}_b").copy_(w[k + 1])
                self._pv(f"bn{l}_g").copy_(w[k + 2]); self._pv(f"bn{l}_b").copy_(w[k + 3])
                self.bn_mean[l].copy_(w[k + 4]); self.bn_var[l].copy_(w[k + 5])

    def init_weights(self, seed_fn, glorot):
        with torch.no_grad():
            prev = self.din
            for l, u in enumerate(self.units):  # Dense: GlorotUniform kernel, zero bias; BN: gamma 1, beta 0
                self._pv(f"d{l}_W").copy_(torch.from_numpy(glorot((prev, u), seed_fn(l))))
                self._pv(f"d{l}_b").zero_()
                self._pv(f"bn{l}_g").fill_(1.0)
                self._pv(f"bn{l}_b").zero_()
                prev = u
