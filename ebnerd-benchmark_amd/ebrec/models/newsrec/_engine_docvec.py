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
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)

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
            b["ws"] = f(max(int(wsf(self.mlp.out_dim, self.E, N)), int(wsf(N, self.E, self.mlp.out_dim)), 1))
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

    def _news_forward(self, mb, n_hist, n_cand, train):
        """MLP over the N = n_hist + n_cand rows already in mb['X0'] -> mb['NE'][:N]."""
        N = n_hist + n_cand
        pv = self.params.view
        x = self.mlp.forward(mb["X0"], n_hist, n_cand, train, self.state, self.p)
        mb["x_last"] = x
        ws = mb.get("ws")  # Dense(E, relu): bias and ReLU ride in the GEMM epilogue
        _hip.call("ebn_dense_relu_fwd_f32", N, self.E, self.mlp.out_dim, _hip.ptr(x), self.mlp.out_dim, _hip.ptr(pv("out_W")), self.E,
                  _hip.ptr(pv("out_b")), _hip.ptr(mb["NE"]), self.E, _hip.ptr(ws), 0 if ws is None else ws.numel(), _hip.stream_handle())

    def _news_backward(self, mb, n_hist, n_cand):
        N = n_hist + n_cand
        pv, g = self.params.view, self.params.g
        prev, x_last = self.mlp.out_dim, mb["x_last"]
        dpre = mb["dNE"]  # relu backward in place
        _hip.call("ebn_bias_relu_bwd_f32", _hip.ptr(mb["NE"]), _hip.ptr(mb["dNE"]), _hip.ptr(dpre), _hip.ptr(g("out_b")),
                  _hip.ptr(mb["partials"]), N, self.E, 0, _hip.stream_handle())
        self._gemm(1, 0, prev, self.E, N, x_last, prev, dpre, self.E, 0.0, g("out_W"), self.E, mb["ws"])
        if self.units:
            self._gemm(0, 1, N, prev, self.E, dpre, self.E, pv("out_W"), self.E, 0.0, mb["dXl"], prev)
            self.mlp.backward(mb["dXl"], mb["X0"], n_hist, n_cand, self.state, self.p, need_dx0=False, loss_dev=self.loss_dev)

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

    def _allreduce_grads(self):
        if self.world > 1:
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
        if self.units:
            u, b = self.units[0], self.mlp.bufs(n)
            pv = self.params.view

            def dense0():
                _hip.call("ebn_dense_relu_fwd_f32", n, u, self.Din, _hip.ptr(mb["X0"]), self.Din, _hip.ptr(pv("d0_W")), u, _hip.ptr(pv("d0_b")),
                          _hip.ptr(b["R"][0]), u, _hip.ptr(b["ws"]), b["ws"].numel(), S())

            out["dense0"] = dense0
        return out

    def check_oob(self):
        """Article-row numbers outside the document-vector matrix raise (device-resident batches are checked here,
        once per epoch, not per step)."""
        oob = getattr(self, "_oob", None)
        if oob is not None and int(oob.item()) != 0:
            oob.zero_()
            raise IndexError(f"article row out of range [0, {self.article_matrix.shape[0]}) for the document-vector matrix")

    def l2_penalty(self) -> float:
        return self.mlp.l2_penalty()

    def extra_state(self):
        return {}
