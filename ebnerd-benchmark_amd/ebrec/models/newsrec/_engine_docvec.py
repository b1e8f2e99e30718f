"""Device engine of NRMSDocVec (reference nrms_docvec.py:99-188): news encoder = per-article MLP
``[Dense(u, relu, l2) -> BatchNormalization -> Dropout] x len(units) -> Dense(E, relu)`` over pre-computed
document vectors, then the same user encoder, scorer, loss and Keras-form Adam as NRMS.

Layout: the B*H history vectors and the B*C candidate vectors of a step are ONE (N, Din) row block (history
first).  Dense layers run over all N rows in one GEMM; BatchNormalization runs per CALL SITE (rows [0,B*H) and
[B*H,N) separately: TimeDistributed(newsencoder) is applied twice, nrms_docvec.py:88-90 and 176-178), each
with its own batch statistics and its own moving-average update [KERAS-SEMANTICS].  The dropout stream is
indexed by (row, column) of the whole block, so both sites draw from one mask.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ebrec import _hip

from ._engine import ADAM_EPS, BETA1, BETA2, LOSS_KIND, loss_kind_of, EncoderBuffers, FlatParams, glorot_uniform_np, require_gpu
from ._mlp import MLPStack


class DocVecEngine:
    def __init__(self, doc_dim: int, units, history_size: int, head_num: int, head_dim: int, attention_hidden_dim: int,
                 dropout: float, learning_rate: float, loss: str, l2: float, seed=None, device=None, process_group=None,
                 bce_on: str = "logits"):
        self.device = require_gpu() if device is None else torch.device(device)
        if loss not in LOSS_KIND:
            raise ValueError(f"this loss not defined {loss}")
        loss_kind_of(loss, bce_on)
        self.bce_on = bce_on
        self.Din, self.units, self.H = int(doc_dim), [int(u) for u in (units or [])], int(history_size)
        self.h, self.d, self.A = int(head_num), int(head_dim), int(attention_hidden_dim)
        self.E = self.h * self.d
        self.p, self.loss, self.l2, self.seed, self.pg = float(dropout), loss, float(l2), seed, process_group
        E, A = self.E, self.A
        shapes = MLPStack.shapes("", self.Din, self.units)
        prev = self.units[-1] if self.units else self.Din
        shapes.update({"out_W": (prev, E), "out_b": (E,), "u_Wqkv": (E, 3 * E), "u_W": (E, A), "u_b": (A,), "u_q": (A,)})
        self.params = FlatParams(shapes, self.device)
        self.use_graph, self._graphs = False, {}
        self.fuse_user_head = True  # False: the per-impression head of a step as its six separate launches (validation)
        # True: BatchNormalization / Dropout / ReLU-backward of the news encoder ride in the Dense matmuls of a training step
        # (csrc/ebn_docvec.hip: one launch per layer and direction); False: the separate passes of csrc/ebn_dense.hip (validation form)
        self.fuse_news_mlp = True
        # True (one rank, fused news encoder + one-launch head): the step ENDS with ebn_dvn_finale_f32 -- the Dense weight gradients with
        # Adam in their tiles' epilogues, Adam over every other parameter, the user head's finishing sums and the batch loss in ONE
        # launch (was: tn_group | user_head_finish | Adam, three launches of the dependent chain).  False: the separate launches
        # (what a multi-rank step runs: the gradient all-reduce sits between the gradients and Adam; the validation form)
        self.fuse_finale = True
        self._dvn_dirty = False  # a fused forward whose backward has not been issued yet (see _dvn)
        self.mlp = MLPStack(self.params, "", self.Din, self.units, self.device, self.l2, on_realloc=lambda: self._graphs.clear())
        self.bn_mean, self.bn_var = self.mlp.bn_mean, self.mlp.bn_var
        self._init_weights(seed)
        st = _hip.StepState()
        st.step, st.seed, st.lr, st.adam_alpha = 0, (0 if seed is None else int(seed)) & 0xFFFFFFFF, learning_rate, 0.0
        self.state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(self.device)
        self._lr = float(learning_rate)
        self._bufs = {}
        self.loss_dev = torch.zeros(1, device=self.device)
        self.reg_dev = torch.zeros(1, device=self.device)
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=self.device)  # a fused launch's column sum left the fixed-point range / was not finite
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # world > 1: save_weights / evaluate / fit raise instead of hanging when only some ranks call them (nothing is built here: the
        # guard's store rendezvous starts at its first enter(); `guard.status()` says "disabled: ..." when no store is reachable)
        self.guard = None
        if self.world > 1:
            from ._dist import LockStepGuard

            self.guard = LockStepGuard(process_group)
        self.force_collectives = False  # True: the multi-rank launch form on a one-rank group (`bench.py --force-dist`; see NRMSEngine)

    @property
    def multi(self) -> bool:
        return self.world > 1 or self.force_collectives

    @property
    def loss_kind(self) -> int:
        return loss_kind_of(self.loss, self.bce_on)

    # ------------------------------------------------------------------ parameters
    def _init_weights(self, seed):
        rng_seed = (lambda k: None) if seed is None else (lambda k: int(seed) * 1000 + k)
        pv = self.params.view
        self.mlp.init_weights(rng_seed, glorot_uniform_np)
        with torch.no_grad():
            prev = self.mlp.out_dim
            pv("out_W").copy_(torch.from_numpy(glorot_uniform_np((prev, self.E), rng_seed(99))))
            s = (lambda: seed) if seed is not None else (lambda: None)
            pv("u_Wqkv").copy_(torch.from_numpy(np.concatenate([glorot_uniform_np((self.E, self.E), s()) for _ in range(3)], 1)))
            pv("u_W").copy_(torch.from_numpy(glorot_uniform_np((self.E, self.A), s())))
            pv("u_q").copy_(torch.from_numpy(glorot_uniform_np((self.A, 1), s())[:, 0]))

    def weight_names(self):
        return self.mlp.weight_names("news") + ["news.out.kernel", "news.out.bias", "user.attn.WQ", "user.attn.WK",
                                                 "user.attn.WV", "user.att.W", "user.att.b", "user.att.q"]

    def get_weights(self):
        pv, E = self.params.view, self.E
        w = pv("u_Wqkv").cpu().numpy()
        return self.mlp.get_weights() + [pv("out_W").cpu().numpy(), pv("out_b").cpu().numpy(), w[:, :E].copy(), w[:, E:2 * E].copy(),
                                         w[:, 2 * E:].copy(), pv("u_W").cpu().numpy(), pv("u_b").cpu().numpy(),
                                         pv("u_q").cpu().numpy().reshape(-1, 1)]

    def set_weights(self, weights):
        n = 6 * len(self.units) + 8
        if len(weights) != n:
            raise ValueError(f"expected {n} weight arrays, got {len(weights)}")
        w = [torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))) for a in weights]
        pv = self.params.view
        self.mlp.set_weights(w)
        i = 6 * len(self.units)
        with torch.no_grad():
            pv("out_W").copy_(w[i]); pv("out_b").copy_(w[i + 1])
            pv("u_Wqkv").copy_(torch.cat(w[i + 2:i + 5], dim=1))
            pv("u_W").copy_(w[i + 5]); pv("u_b").copy_(w[i + 6].reshape(-1)); pv("u_q").copy_(w[i + 7].reshape(-1))

    def count_params(self):
        return sum(int(np.prod(s)) for s in self.params.shapes.values()) + 2 * sum(self.units)

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
    def _mlp_bufs(self, N):
        b = self._bufs.get("mlp")
        if b is None or b["N"] < N:
            f = lambda *s: torch.empty(*s, device=self.device)
            b = {"N": N, "X0": f(N, self.Din), "NE": f(N, self.E), "dNE": f(N, self.E), "dXl": f(N, self.mlp.out_dim),
                 "scores": f(N), "probs": f(N), "labels": f(N)}
            b["partials"] = f(int(_hip.lib().ebn_colsum_partials_len(N, self.E)))
            wsf = _hip.lib().ebn_gemm_workspace_floats
            b["ws"] = f(max(int(wsf(self.mlp.out_dim, self.E, N)), int(wsf(N, self.E, self.mlp.out_dim)), int(wsf(N, self.mlp.out_dim, self.E)), 1))
            self.mlp.bufs(N)
            self._bufs["mlp"] = b
            self._graphs.clear()  # captured graphs hold raw pointers into the old buffers (an eval pass may grow them)
        return b

    def _user_bufs(self, B, H=None):
        """H: history length of an INFERENCE pass that differs from hparams.history_size (the history-length sweep of
        ebnerd_nrms_doc_hist.py:270-300); training always runs hparams.history_size."""
        H = self.H if H is None else int(H)
        key = "user" if H == self.H else ("user", H)
        b = self._bufs.get(key)
        if b is None or b.n_seq < B:
            b = EncoderBuffers(B, H, self.E, self.E, self.A, self.device, own_input=False, need_dx=False)
            b.duser = torch.empty(B, self.E, device=self.device)
            b.loss_rows = torch.empty(B, device=self.device)
            b.head_partials = torch.empty(max(int(_hip.lib().ebn_user_head_partials_len(B, self.A)), 1), device=self.device)
            self._bufs[key] = b
            if H == self.H:
                self._graphs.clear()
        return b

    # ------------------------------------------------------------------ kernels
    def _gemm(self, tA, tB, M, N, K, A, lda, B, ldb, beta, C, ldc, ws=None):
        _hip.call("ebn_gemm_f32_ws", tA, tB, M, N, K, ctypes.c_float(1.0), _hip.ptr(A), lda, _hip.ptr(B), ldb,
                  ctypes.c_float(beta), _hip.ptr(C), ldc, _hip.ptr(ws), 0 if ws is None else ws.numel(), _hip.stream_handle())

    def _dvn(self, mb, n_hist, n_cand):
        """ebn_dvn_args of the fused news-encoder step over this buffer set, or None when the shape is outside what the fused
        launches take (then the per-pass kernels run)."""
        if self._dvn_dirty:
            # a step that ran only half way (an exception between its forward and its backward) left the fixed-point accumulators of
            # the fused launches dirty: the contract of ebn_dvn_fwd_train_f32 is a zero scratch.  Every shape's scratch, whether or
            # not the CURRENT shape takes the fused launches
            for k, v in mb.items():
                if isinstance(k, tuple) and k[0] == "dvn_stat":
                    v.zero_()
            self._dvn_dirty = False
        if not (self.fuse_news_mlp and self.units):
            return None
        # the cached argument block freezes raw pointers AND the scalars p / l2: they are part of the key, so that changing
        # `eng.p` or `eng.l2` between steps reaches the fused launches as it reaches the per-pass kernels
        fin_args = self._finale_args() if self._finale_live(n_cand // max(n_hist // self.H, 1) if n_hist else 0) else None
        fin = fin_args is not None
        key = ("dvn", n_hist, n_cand, self.p, self.l2, fin)
        a = mb.get(key)
        if a is None:
            L, N, pv, g, b = len(self.units), mb["N"], self.params.view, self.params.g, self.mlp.bufs(mb["N"])
            a = _hip.DvnArgs()
            a.n_layers, a.din, a.e_out, a.n0, a.n1, a.drop_p, a.l2 = L, self.Din, self.E, n_hist, n_cand, self.p, self.l2
            for l, u in enumerate(self.units):
                a.units[l] = u
            if not _hip.lib().ebn_dvn_supported(ctypes.byref(a)):
                mb[key] = False
                return None
            f = lambda *s: torch.empty(*s, device=self.device)
            if "dvn_dY" not in mb:
                mb["dvn_dY"] = [f(N, u) for u in self.units]
                mb["dvn_dP"] = [f(N, u) for u in self.units] + [f(N, self.E)]
            # the scratch is sized by the row tiling of (n_hist, n_cand): one per shape
            mb[("dvn_stat", n_hist, n_cand, self.p, self.l2, fin)] = stat = torch.zeros(int(_hip.lib().ebn_dvn_stat_floats(ctypes.byref(a))), device=self.device)
            for l in range(L):
                a.W[l], a.b[l] = pv(f"d{l}_W").data_ptr(), pv(f"d{l}_b").data_ptr()
                a.gamma[l], a.beta[l] = pv(f"bn{l}_g").data_ptr(), pv(f"bn{l}_b").data_ptr()
                a.moving_mean[l], a.moving_var[l] = self.bn_mean[l].data_ptr(), self.bn_var[l].data_ptr()
                a.R[l], a.Xn[l] = b["R"][l].data_ptr(), b["Xn"][l].data_ptr()
                a.dY[l], a.dP[l] = mb["dvn_dY"][l].data_ptr(), mb["dvn_dP"][l].data_ptr()
                a.ggamma[l], a.gbeta[l] = g(f"bn{l}_g").data_ptr(), g(f"bn{l}_b").data_ptr()
            a.W[L], a.b[L], a.dP[L] = pv("out_W").data_ptr(), pv("out_b").data_ptr(), mb["dvn_dP"][L].data_ptr()
            a.X0, a.NE, a.dNE, a.stat = mb["X0"].data_ptr(), mb["NE"].data_ptr(), mb["dNE"].data_ptr(), stat.data_ptr()
            a.loss = None if fin else self.loss_dev.data_ptr()  # the finale forms the whole batch loss, L2 term included
            a.range_flag = self.range_flag.data_ptr()
            # weight gradients of all L + 1 Dense kernels as ONE launch: dW_l = Xn_{l-1}^T . dP_l, bias gradient = column sums
            # of dP_l, + 2 l2 W_l for the regularised (hidden) kernels
            probs = (_hip.TnProblem * (L + 1))()
            dims = [self.Din] + self.units + [self.E]
            for l in range(L + 1):
                q = probs[l]
                q.M, q.N, q.K = dims[l], dims[l + 1], n_hist + n_cand
                x = mb["X0"] if l == 0 else b["Xn"][l - 1]
                q.A, q.lda, q.B, q.ldb = x.data_ptr(), dims[l], mb["dvn_dP"][l].data_ptr(), dims[l + 1]
                wn, bn = (f"d{l}_W", f"d{l}_b") if l < L else ("out_W", "out_b")
                q.C, q.ldc, q.colsum = g(wn).data_ptr(), dims[l + 1], g(bn).data_ptr()
                if l < L and self.l2 > 0:
                    q.l2_W, q.two_lambda = pv(wn).data_ptr(), 2.0 * self.l2
            a._probs, a._stat, a._finale = probs, stat, fin_args
            mb[key] = a
        return a or None

    def _finale_live(self, C=None) -> bool:
        """whether a training step of this engine ends with ebn_dvn_finale_f32 (see `fuse_finale`)"""
        if not (self.fuse_finale and self.fuse_news_mlp and self.fuse_user_head and self.units and not self.multi):
            return False
        L = _hip.lib()
        if C is not None and int(L.ebn_user_head_supported(self.H, int(C), self.E, self.A)) == 0:
            return False
        return int(L.ebn_attn_bwd_pooled_supported(self.H, self.d)) != 0

    def _finale_args(self):
        """ebn_dvn_finale of this engine's flat parameter buffers: the `rest` ranges are every parameter whose gradient no tile of the
        weight-gradient group and not the head's finishing sums produce -- the BatchNormalization scales / offsets and the user encoder's
        projection and AttLayer2 kernels."""
        P = self.params
        owned = {f"d{l}_{s}" for l in range(len(self.units)) for s in ("W", "b")} | {"out_W", "out_b", "u_q", "u_b"}
        rest = sorted((P.offsets[k], int(np.prod(P.shapes[k]))) for k in P.shapes if k not in owned)
        merged = []
        for off, n in rest:
            if merged and merged[-1][0] + merged[-1][1] == off:
                merged[-1][1] += n
            else:
                merged.append([off, n])
        if len(merged) > _hip.DVN_FINALE_MAX_REST:
            return None
        f = _hip.DvnFinale()
        f.theta, f.grad, f.m, f.v, f.numel = P.data.data_ptr(), P.grad.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.numel
        f.beta1, f.beta2, f.eps, f.grad_scale = BETA1, BETA2, ADAM_EPS, 1.0
        f.n_rest = len(merged)
        for i, (off, n) in enumerate(merged):
            f.rest_off[i], f.rest_len[i] = off, n
        f.A, f.dq, f.db, f.loss_out = self.A, P.g("u_q").data_ptr(), P.g("u_b").data_ptr(), self.loss_dev.data_ptr()
        return f

    def _news_forward(self, mb, n_hist, n_cand, train):
        """MLP over the N = n_hist + n_cand rows already in mb['X0'] -> mb['NE'][:N]."""
        N = n_hist + n_cand
        pv = self.params.view
        a = self._dvn(mb, n_hist, n_cand) if train else None
        mb["dvn_live"] = a
        if a is not None:
            self._dvn_dirty = True  # until the backward call of this step has been issued
            _hip.call("ebn_dvn_fwd_train_f32", ctypes.byref(a), _hip.ptr(self.state), _hip.stream_handle())
            return
        x = self.mlp.forward(mb["X0"], n_hist, n_cand, train, self.state, self.p)
        mb["x_last"] = x
        ws = mb.get("ws")  # Dense(E, relu): bias and ReLU ride in the GEMM epilogue
        _hip.call("ebn_dense_relu_fwd_f32", N, self.E, self.mlp.out_dim, _hip.ptr(x), self.mlp.out_dim, _hip.ptr(pv("out_W")), self.E,
                  _hip.ptr(pv("out_b")), _hip.ptr(mb["NE"]), self.E, _hip.ptr(ws), 0 if ws is None else ws.numel(), _hip.stream_handle())

    def _news_backward(self, mb, n_hist, n_cand):
        N = n_hist + n_cand
        pv, g = self.params.view, self.params.g
        a = mb.get("dvn_live")
        if a is not None:
            _hip.call("ebn_dvn_bwd_f32", ctypes.byref(a), _hip.ptr(self.state), _hip.stream_handle())
            if a._finale is not None:  # gradients, the head's finishing sums, the loss and Adam: the step's last launch
                ub, f = self._user_bufs(n_hist // self.H), a._finale
                f.head_partials, f.loss_rows, f.B = ub.head_partials.data_ptr(), ub.loss_rows.data_ptr(), n_hist // self.H
                _hip.call("ebn_dvn_finale_f32", ctypes.byref(a), a._probs, len(a._probs), ctypes.byref(f), _hip.ptr(self.state), _hip.stream_handle())
            else:
                _hip.call("ebn_gemm_tn_group_f32", a._probs, len(a._probs), _hip.stream_handle())
            self._dvn_dirty = False
            return
        prev, x_last = self.mlp.out_dim, mb["x_last"]
        dpre = mb["dNE"]  # relu backward in place
        _hip.call("ebn_bias_relu_bwd_f32", _hip.ptr(mb["NE"]), _hip.ptr(mb["dNE"]), _hip.ptr(dpre), _hip.ptr(g("out_b")),
                  _hip.ptr(mb["partials"]), N, self.E, 0, _hip.stream_handle())
        if self.units:  # dW = X^T.dpre and dX = dpre.W^T of the output Dense: independent of each other, one launch for the pair
            ws = mb["ws"]
            _hip.call("ebn_dense_bwd_pair_f32", N, prev, self.E, _hip.ptr(x_last), prev, _hip.ptr(dpre), self.E, _hip.ptr(pv("out_W")), self.E,
                      ctypes.c_float(0.0), _hip.ptr(g("out_W")), self.E, _hip.ptr(mb["dXl"]), prev,
                      _hip.ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, _hip.stream_handle())
            self.mlp.backward(mb["dXl"], mb["X0"], n_hist, n_cand, self.state, self.p, need_dx0=False, loss_dev=self.loss_dev)
        else:
            self._gemm(1, 0, prev, self.E, N, x_last, prev, dpre, self.E, 0.0, g("out_W"), self.E, mb["ws"])

    @staticmethod
    def _fwd_scratch(b):
        return b.fwd_scratch

    def _enc(self, ub, B, X, dims_only=False):
        pv = self.params.view
        dims = _hip.EncoderDims(B, ub.L, self.E, self.h, self.d, self.A, -1, 0.0)
        params = _hip.EncoderParams(pv("u_Wqkv").data_ptr(), pv("u_W").data_ptr(), pv("u_b").data_ptr(), pv("u_q").data_ptr())
        acts = _hip.EncoderActs(X.data_ptr(), ub.QKV.data_ptr(), ub.Y.data_ptr(), ub.U.data_ptr(), ub.w.data_ptr(), ub.out.data_ptr())
        return dims, params, acts

    def _upload(self, mb, his, pred):
        B, C = his.shape[0], pred.shape[1]
        n_hist, n_cand = B * self.H, B * C
        for arr, r0, n in ((his, 0, n_hist), (pred, n_hist, n_cand)):
            t = arr if isinstance(arr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.float32)))
            mb["X0"][r0:r0 + n].copy_(t.reshape(n, self.Din).to(device=self.device, dtype=torch.float32), non_blocking=True)
        return n_hist, n_cand

    def _check_shapes(self, his, pred):
        if his.ndim != 3 or his.shape[1] != self.H or his.shape[2] != self.Din:
            raise ValueError(f"his_input_title must be (B, {self.H}, {self.Din}), got {tuple(his.shape)}")
        if pred.ndim != 3 or pred.shape[0] != his.shape[0] or pred.shape[2] != self.Din:
            raise ValueError(f"pred_input_title must be (B, C, {self.Din}), got {tuple(pred.shape)}")

    # ------------------------------------------------------------------ public compute
    def encode_news(self, vecs, chunk=65536) -> torch.Tensor:
        vecs = vecs if isinstance(vecs, torch.Tensor) else np.asarray(vecs, dtype=np.float32)
        N = vecs.shape[0]
        out = torch.empty(N, self.E, device=self.device)
        for s in range(0, N, chunk):
            n = min(chunk, N - s)
            mb = self._mlp_bufs(min(chunk, N))
            t = vecs[s:s + n] if isinstance(vecs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(vecs[s:s + n]))
            mb["X0"][:n].copy_(t.to(device=self.device, dtype=torch.float32))
            self._news_forward(mb, n, 0, False)
            out[s:s + n].copy_(mb["NE"][:n])
        return out

    def encode_users_from_news(self, NEh: torch.Tensor) -> torch.Tensor:
        B, H = NEh.shape[0], NEh.shape[1]  # H from the input: no weight of the user encoder depends on it
        ub = self._user_bufs(B, H)
        X = NEh.reshape(B * H, self.E).contiguous()
        dims, params, acts = self._enc(ub, B, X)
        _hip.call("ebn_encoder_fwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), ctypes.byref(self._fwd_scratch(ub)), None, _hip.stream_handle())
        return ub.out[:B].clone()

    def encode_users(self, his) -> torch.Tensor:
        his = np.asarray(his, dtype=np.float32) if not isinstance(his, torch.Tensor) else his
        B, H = his.shape[0], his.shape[1]
        return self.encode_users_from_news(self.encode_news(his.reshape(B * H, self.Din)).view(B, H, self.E))

    def forward(self, his, pred, mode="softmax"):
        his = his if isinstance(his, torch.Tensor) else np.asarray(his)
        pred = pred if isinstance(pred, torch.Tensor) else np.asarray(pred)
        self._check_shapes(his, pred)
        B, C = his.shape[0], pred.shape[1]
        mb, ub = self._mlp_bufs(B * (self.H + C)), self._user_bufs(B)
        n_hist, n_cand = self._upload(mb, his, pred)
        self._news_forward(mb, n_hist, n_cand, False)
        dims, params, acts = self._enc(ub, B, mb["NE"])
        _hip.call("ebn_encoder_fwd_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), ctypes.byref(self._fwd_scratch(ub)), None, _hip.stream_handle())
        scores, probs = torch.empty(B, C, device=self.device), torch.empty(B, C, device=self.device)
        _hip.call("ebn_score_fwd_f32", _hip.ptr(mb["NE"][n_hist:]), _hip.ptr(ub.out), _hip.ptr(scores), _hip.ptr(probs), B, C,
                  self.E, 0 if mode == "softmax" else 1, _hip.stream_handle())
        return probs, scores

    def eval_loss(self, his, pred, y):
        probs, scores = self.forward(his, pred)
        B, C = scores.shape
        labels = torch.as_tensor(np.asarray(y, dtype=np.float32)).to(self.device).reshape(B, C).contiguous()
        mb, ub = self._mlp_bufs(B * (self.H + C)), self._user_bufs(B)
        rows, jc, ju = torch.empty(B, device=self.device), torch.empty(B * C, self.E, device=self.device), torch.empty(B, self.E, device=self.device)
        loss = torch.empty(1, device=self.device)
        _hip.call("ebn_score_loss_bwd_f32", _hip.ptr(mb["NE"][B * self.H:]), _hip.ptr(ub.out), _hip.ptr(scores), _hip.ptr(labels),
                  _hip.ptr(rows), _hip.ptr(jc), _hip.ptr(ju), B, C, self.E, self.loss_kind, ctypes.c_float(1.0 / B), _hip.stream_handle())
        _hip.call("ebn_sum_f32", _hip.ptr(rows), B, ctypes.c_float(1.0), _hip.ptr(loss), 0, _hip.stream_handle())
        return loss, probs

    def pair_scores(self, user, news, u_idx, n_idx, sigmoid=True):
        n = u_idx.numel()
        out = torch.empty(n, device=self.device)
        _hip.call("ebn_pair_score_f32", _hip.ptr(user), _hip.ptr(news), _hip.ptr(u_idx), _hip.ptr(n_idx), _hip.ptr(out), n,
                  self.E, 1 if sigmoid else 0, _hip.stream_handle())
        return out

    def enable_graphs(self, flag=True):
        """Capture the per-(B, C) kernel sequence of a train step into a hipGraph (the DocVec step is ~70 small
        launches: launch-bound without it)."""
        self.use_graph = bool(flag)
        if not flag:
            self._graphs = {}
        return self

    def set_article_matrix(self, matrix) -> None:
        """Keep the loader's (n_articles+1, Din) document-vector matrix in HBM (386 MB for the 125 542 EB-NeRD
        articles x 768); batches can then be article-row numbers, gathered on the device by the embedding-gather
        kernel (``train_step(..., indexed=True)``)."""
        m = np.asarray(matrix)
        if m.ndim != 2 or m.shape[1] != self.Din or not np.issubdtype(m.dtype, np.floating):
            raise ValueError(f"article matrix must be float (n_articles+1, {self.Din}), got {m.dtype} {m.shape}")
        self.article_matrix = torch.from_numpy(np.ascontiguousarray(m.astype(np.float32))).to(self.device)
        self._article_matrix_src = matrix
        self._oob = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _stage_indexed(self, mb, his_idx, pred_idx, y=None):
        """Article-row numbers -> mb["art_idx"], then the document vectors are gathered on the device.  Returns y, or None
        when the labels were copied along (device-resident batch in the step's dtypes: one copy launch for all three)."""
        n = his_idx.shape[0] * (self.H + pred_idx.shape[1])
        if "art_idx" not in mb:
            mb["art_idx"] = torch.empty(mb["N"], dtype=torch.int32, device=self.device)
        same_dev = lambda t: t.is_cuda and (self.device.index is None or t.device.index == self.device.index)
        ok = lambda t, dt: isinstance(t, torch.Tensor) and same_dev(t) and t.dtype == dt and t.is_contiguous()
        if ok(his_idx, torch.int32) and ok(pred_idx, torch.int32) and ok(y, torch.float32):
            # ONE launch: step-state advance + label copy + the gather, reading the row numbers from the two tensors as they are
            _hip.call("ebn_docvec_stage_gather_f32", _hip.ptr(his_idx), his_idx.numel(), _hip.ptr(pred_idx), pred_idx.numel(), _hip.ptr(y),
                      _hip.ptr(mb["labels"]), y.numel(), _hip.ptr(self.article_matrix), self.article_matrix.shape[0], self.Din,
                      _hip.ptr(mb["X0"]), _hip.ptr(self._oob), _hip.ptr(self.state), BETA1, BETA2, _hip.stream_handle())
            self._advanced = True
            return None
        elif not isinstance(his_idx, torch.Tensor) and not isinstance(pred_idx, torch.Tensor) and y is not None and not isinstance(y, torch.Tensor):
            # host batch (what the loaders hand over): ONE asynchronous copy out of a pinned, double-buffered staging area; the
            # kernel that unpacks it also advances the step state -- the host never waits for the GPU
            B, C = his_idx.shape[0], pred_idx.shape[1]
            n_lab, tot = B * C, n + B * C
            st = getattr(self, "_host_stage", None)
            if st is None or st["pinned"][0].numel() < tot:
                st = self._host_stage = {"pinned": [torch.empty(2 * tot, dtype=torch.int32).pin_memory() for _ in range(2)],
                                         "dev": torch.empty(2 * tot, dtype=torch.int32, device=self.device), "ev": [None, None], "k": 0}
            k = st["k"] = st["k"] ^ 1
            if st["ev"][k] is not None:
                st["ev"][k].synchronize()
            hs = st["pinned"][k].numpy()
            nh = B * self.H
            hs[:nh] = np.asarray(his_idx).reshape(-1)
            hs[nh:n] = np.asarray(pred_idx).reshape(-1)
            hs[n:tot].view(np.float32)[:] = np.asarray(y, dtype=np.float32).reshape(-1)
            st["dev"][:tot].copy_(st["pinned"][k][:tot], non_blocking=True)
            st["ev"][k] = torch.cuda.Event()
            st["ev"][k].record()
            _hip.call("ebn_docvec_stage_gather_f32", _hip.ptr(st["dev"]), n, None, 0, _hip.ptr(st["dev"][n:]), _hip.ptr(mb["labels"]), n_lab,
                      _hip.ptr(self.article_matrix), self.article_matrix.shape[0], self.Din, _hip.ptr(mb["X0"]), _hip.ptr(self._oob),
                      _hip.ptr(self.state), BETA1, BETA2, _hip.stream_handle())
            self._advanced = True
            return None
        else:
            off = 0
            for a in (his_idx, pred_idx):
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1).astype(np.int32, copy=False)))
                t = t.reshape(-1)
                mb["art_idx"][off: off + t.numel()].copy_(t.to(device=self.device, dtype=torch.int32), non_blocking=True)
                off += t.numel()
        _hip.call("ebn_gather_rows_f32", _hip.ptr(mb["art_idx"]), _hip.ptr(self.article_matrix), _hip.ptr(mb["X0"]), n, self.Din,
                  self.article_matrix.shape[0], None, -1, ctypes.c_float(0.0), _hip.ptr(self._oob), _hip.stream_handle())
        return y

    def train_step(self, his, pred, y, return_probs=False, indexed=False):
        his = his if isinstance(his, torch.Tensor) else np.asarray(his)
        pred = pred if isinstance(pred, torch.Tensor) else np.asarray(pred)
        if not indexed:
            self._check_shapes(his, pred)
        elif his.ndim != 2 or his.shape[1] != self.H or pred.ndim != 2 or pred.shape[0] != his.shape[0]:
            raise ValueError(f"indexed batches must be (B, {self.H}) and (B, C), got {tuple(his.shape)} {tuple(pred.shape)}")
        B, C = his.shape[0], pred.shape[1]
        mb, ub = self._mlp_bufs(B * (self.H + C)), self._user_bufs(B)  # (re)allocation clears the captured graphs
        self._dvn(mb, B * self.H, B * C)  # buffers of the fused news-encoder launches: allocated outside any capture
        self._advanced = False
        if indexed:
            y = self._stage_indexed(mb, his, pred, y)
        else:
            self._upload(mb, his, pred)
        if y is not None:
            labels = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32))))
            mb["labels"][: B * C].copy_(labels.reshape(-1).to(device=self.device, dtype=torch.float32))
        if self.use_graph:
            # graph(forward + backward) -> gradient all-reduce over RCCL (eager, data-parallel only) -> graph(Adam)
            adv = self._advanced
            gkey = (B, C, adv, bool(self.fuse_news_mlp), self.p, self.l2, bool(self.fuse_finale), bool(self.multi))  # a captured graph freezes the launch form and its scalars
            g = self._graphs.get(gkey)
            if g is None:
                torch.cuda.synchronize()
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with _hip.capture(g1):
                    self._fwd_bwd_kernels(B, C, adv)
                if self._step_applied_adam(mb):  # one rank, the finale: the optimizer ran inside the step's last launch
                    g2 = None
                else:
                    with _hip.capture(g2, pool=g1.pool()):
                        self._optimizer_kernels()
                g = self._graphs[gkey] = (g1, g2)
            g[0].replay()
            if g[1] is not None:
                self._allreduce_grads()
                g[1].replay()
        else:
            self._fwd_bwd_kernels(B, C, self._advanced)
            if not self._step_applied_adam(mb):
                self._allreduce_grads()
                self._optimizer_kernels()
        if return_probs:
            return self.loss_dev, mb["probs"][: B * C].view(B, C), mb["labels"][: B * C].view(B, C)
        return self.loss_dev

    @staticmethod
    def _step_applied_adam(mb) -> bool:
        a = mb.get("dvn_live")
        return a is not None and a._finale is not None

    def _fwd_bwd_kernels(self, B, C, advanced=False):
        E = self.E
        S = _hip.stream_handle
        mb, ub = self._mlp_bufs(B * (self.H + C)), self._user_bufs(B)
        n_hist, n_cand = B * self.H, B * C
        st = _hip.ptr(self.state)
        if not advanced:
            _hip.call("ebn_step_advance", st, BETA1, BETA2, S())
        self._news_forward(mb, n_hist, n_cand, True)
        a = mb.get("dvn_live")
        if a is not None and a._finale is not None:
            self._user_stage_deferred(B, C, mb, ub)
            self._news_backward(mb, n_hist, n_cand)
            return
        dims, params, acts = self._enc(ub, B, mb["NE"])
        cand = mb["NE"][n_hist:]
        g = self.params.g
        grads = _hip.EncoderGrads(g("u_Wqkv").data_ptr(), g("u_W").data_ptr(), g("u_b").data_ptr(), g("u_q").data_ptr())
        scratch = _hip.EncoderScratch(ub.dY.data_ptr(), ub.dQKV.data_ptr(), ub.de.data_ptr(), ub.partials.data_ptr(),
                                      ub.ws.data_ptr(), ub.ws.numel())
        # user encoder + scorer + compiled loss, forward and backward, as one stage call (the per-impression middle of it is one
        # launch where it fits): batch loss, d(cand) = dNE[n_hist:], d(history news vectors) = dNE[:n_hist]
        _hip.call("ebn_user_stage_train_f32", ctypes.byref(dims), ctypes.byref(params), ctypes.byref(acts), _hip.ptr(cand),
                  _hip.ptr(mb["labels"]), _hip.ptr(mb["scores"]), _hip.ptr(mb["probs"]), _hip.ptr(ub.loss_rows), _hip.ptr(self.loss_dev),
                  _hip.ptr(mb["dNE"][n_hist:]), _hip.ptr(ub.duser), ctypes.byref(grads), ctypes.byref(scratch),
                  _hip.ptr(ub.head_partials) if self.fuse_user_head else None, _hip.ptr(mb["dNE"]), C, self.loss_kind,
                  ctypes.c_float(1.0 / B), st, S())
        self._news_backward(mb, n_hist, n_cand)

    def _user_stage_deferred(self, B, C, mb, ub):
        """ebn_user_stage_train_f32's fused branch (csrc/ebn_encoder.hip) kernel by kernel, with the head's finishing launch left to
        the step's finale (dq == db == NULL): user encoder forward up to the AttLayer2 matmul, the one-launch head, the two
        gradient-GEMM pairs around the attention backward.  Writes d(cand) = dNE[n_hist:], d(history news vectors) = dNE[:n_hist],
        the user encoder's kernel gradients, and the per-impression partials / loss rows the finale sums."""
        S, H, E, A = _hip.stream_handle, self.H, self.E, self.A
        R = B * H
        pv, g = self.params.view, self.params.g
        ws, wsn = _hip.ptr(ub.ws), ub.ws.numel()
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        X, cand, dcand = mb["NE"], mb["NE"][R:], mb["dNE"][R:]
        _hip.call("ebn_gemm_f32_site", 0, 0, R, 3 * E, E, one, _hip.ptr(X), E, _hip.ptr(pv("u_Wqkv")), 3 * E, zero, _hip.ptr(ub.QKV), 3 * E, ws, wsn, 1, S())
        _hip.call("ebn_attn_fwd_f32", _hip.ptr(ub.QKV), 3 * E, _hip.ptr(ub.Y), E, B, H, self.h, self.d, None, -1, zero, S())
        _hip.call("ebn_gemm_f32_ws", 0, 0, R, A, E, one, _hip.ptr(ub.Y), E, _hip.ptr(pv("u_W")), A, zero, _hip.ptr(ub.U), A, ws, wsn, S())
        _hip.call("ebn_user_head_train_f32", _hip.ptr(ub.U), _hip.ptr(pv("u_b")), _hip.ptr(pv("u_q")), _hip.ptr(ub.Y), _hip.ptr(cand), _hip.ptr(mb["labels"]),
                  _hip.ptr(ub.w), _hip.ptr(ub.out), _hip.ptr(mb["scores"]), _hip.ptr(mb["probs"]), _hip.ptr(ub.loss_rows), _hip.ptr(self.loss_dev), _hip.ptr(dcand),
                  _hip.ptr(ub.duser), _hip.ptr(ub.de), None, None, _hip.ptr(ub.head_partials), B, H, C, E, A, self.loss_kind, ctypes.c_float(1.0 / B), S())
        _hip.call("ebn_dense_bwd_pair_f32", R, E, A, _hip.ptr(ub.Y), E, _hip.ptr(ub.U), A, _hip.ptr(pv("u_W")), A, zero, _hip.ptr(g("u_W")), A,
                  _hip.ptr(ub.dY), E, ws, wsn, S())
        _hip.call("ebn_attn_bwd_pooled_f32", _hip.ptr(ub.QKV), 3 * E, _hip.ptr(ub.dY), E, _hip.ptr(ub.w), _hip.ptr(ub.duser), E, _hip.ptr(ub.dQKV), 3 * E,
                  B, H, self.h, self.d, None, -1, zero, S())
        _hip.call("ebn_dense_bwd_pair_f32", R, E, 3 * E, _hip.ptr(X), E, _hip.ptr(ub.dQKV), 3 * E, _hip.ptr(pv("u_Wqkv")), 3 * E, zero, _hip.ptr(g("u_Wqkv")),
                  3 * E, _hip.ptr(mb["dNE"]), E, ws, wsn, S())

    def _allreduce_grads(self):
        if self.multi:
            torch.distributed.all_reduce(self.params.grad, group=self.pg)

    def _optimizer_kernels(self):
        S = _hip.stream_handle
        st = _hip.ptr(self.state)
        P = self.params
        _hip.call("ebn_adam_keras_step_f32", _hip.ptr(P.data), _hip.ptr(P.grad), _hip.ptr(P.m), _hip.ptr(P.v), P.numel, st,
                  BETA1, BETA2, ADAM_EPS, ctypes.c_float(1.0 / self.world), S())

    def roofline_kernels(self, B, C):
        """Launchers of single kernels of the training step at batch shape (B, C) on the step's own buffers (bench.py):
        "gather" = factory rows -> launcher of the document-vector gather, "dense0" = the widest Dense(relu) GEMM of the MLP
        (rows x Din -> units[0])."""
        mb = self._mlp_bufs(B * (self.H + C))
        n = B * (self.H + C)
        S = _hip.stream_handle

        def make_gather(rows):  # rows: (n,) int32 article-row numbers on the device (bench.py cycles through several sets)
            def gather():
                _hip.call("ebn_gather_rows_f32", _hip.ptr(rows), _hip.ptr(self.article_matrix), _hip.ptr(mb["X0"]), n, self.Din,
                          self.article_matrix.shape[0], None, -1, ctypes.c_float(0.0), _hip.ptr(self._oob), S())

            return gather

        out = {"gather": make_gather}
        a = self._dvn(mb, B * self.H, B * C)
        if a is not None:
            # the time-dominant launch of the fused step: the weight gradients of all Dense kernels as one grouped TN product -- on one
            # rank the step's closing launch (ebn_dvn_finale_f32: the same tiles + Adam + the head's finishing sums).  The probe runs
            # it with Adam pointed at SHADOW copies of the parameter / moment buffers: same traffic, the model is left alone
            out["dw_group_flops"] = float(sum(2.0 * q.M * q.N * q.K for q in a._probs))
            out["dw_group_bytes"] = float(sum(4.0 * (q.K * (q.M + q.N) + q.M * q.N) for q in a._probs))
            if a._finale is not None:
                P, f = self.params, _hip.DvnFinale()
                ctypes.memmove(ctypes.byref(f), ctypes.byref(a._finale), ctypes.sizeof(f))
                shadow = mb.setdefault("finale_shadow", [P.data.clone(), P.m.clone(), P.v.clone()])
                f.theta, f.m, f.v = (t.data_ptr() for t in shadow)
                ub = self._user_bufs(B)
                f.head_partials, f.loss_rows, f.B = ub.head_partials.data_ptr(), ub.loss_rows.data_ptr(), B
                out["dw_group"] = lambda: _hip.call("ebn_dvn_finale_f32", ctypes.byref(a), a._probs, len(a._probs), ctypes.byref(f), _hip.ptr(self.state), S())
                out["dw_group_bytes"] += 4.0 * 7 * P.numel  # Adam: theta, m, v read and written, the gradient re-read for the ranges no tile owns (upper bound)
                out["dw_group_is_finale"] = True
            else:
                out["dw_group"] = lambda: _hip.call("ebn_gemm_tn_group_f32", a._probs, len(a._probs), S())
        return out

    def check_oob(self):
        """Article-row numbers outside the document-vector matrix raise (device-resident batches are checked here,
        once per epoch, not per step)."""
        oob = getattr(self, "_oob", None)
        flags = torch.cat([oob if oob is not None else torch.zeros(1, dtype=torch.int32, device=self.device), self.range_flag])
        if self.world > 1:  # every rank raises together: a rank raising alone would leave its peers in the next collective
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        bad_rows, bad_range = (int(v) != 0 for v in flags.cpu().tolist())
        if bad_rows and oob is not None:
            oob.zero_()
        if bad_range:  # (both words cleared before anything is raised)
            self.range_flag.zero_()
            self._dvn_dirty = True  # the accumulators hold a partial step: zero them before the next one
        if bad_rows:
            raise IndexError(f"article row out of range [0, {self.article_matrix.shape[0]}) for the document-vector matrix")
        if bad_range:
            raise FloatingPointError("a column sum of the news encoder's BatchNormalization left the range of its fixed-point accumulator or was not "
                                     "finite: the run has diverged (fuse_news_mlp=False runs the separate passes, which propagate NaN instead)")

    def sync_moving_statistics(self) -> None:
        """world > 1 with BatchNormalization layers: average the moving mean / variance over the ranks (a COLLECTIVE; fit() calls it
        at the end of every epoch, evaluate() and save_weights() on entry -- see MLPStack.sync_moving_statistics for the contract)."""
        if self.world > 1:
            self.mlp.sync_moving_statistics(self.pg)

    def l2_penalty(self) -> float:
        return self.mlp.l2_penalty()

    def extra_state(self):
        return {}
