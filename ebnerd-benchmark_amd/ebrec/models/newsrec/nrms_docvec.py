"""NRMSDocVec on MI355X: the reference's surface (nrms_docvec.py:21-39, 182-188) over the HIP kernels.

    model = NRMSDocVec(hparams=hparams_nrms_docvec, seed=123)
    model.model.fit(train_loader, validation_data=val_loader, ...)     # loaders built from {article_id: doc vector}
    scores = model.scorer.predict(test_loader)

The news encoder consumes pre-computed document vectors of width ``hparams.title_size`` (768): per
article ``[Dense(u, relu, l2) -> BatchNormalization -> Dropout] x units -> Dense(head_num*head_dim, relu)``
(nrms_docvec.py:113-135); user encoder, scorer, loss and optimizer are those of NRMS.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ebrec import _hip

from ._engine_docvec import DocVecEngine
from ._keras_like import EncoderModel, ScorerModel, TrainModel, dedup_rows


class NRMSDocVec:
    def __init__(self, hparams, seed: int = None, *, device=None, process_group=None, bce_on: str = "logits"):
        self.hparams = hparams
        self.seed = seed
        if seed is not None:
            np.random.seed(seed)  # nrms_docvec.py:31-32
            torch.manual_seed(seed)
        self._get_loss(hparams.loss)
        self._get_opt(hparams.optimizer, hparams.learning_rate)
        self._engine = DocVecEngine(
            hparams.title_size, hparams.newsencoder_units_per_layer, hparams.history_size, hparams.head_num,
            hparams.head_dim, hparams.attention_hidden_dim, hparams.dropout, hparams.learning_rate, hparams.loss,
            hparams.newsencoder_l2_regularization, seed=seed, device=device, process_group=process_group, bce_on=bce_on)
        self.model, self.scorer = self._build_graph()

    def _get_loss(self, loss: str):
        if loss == "cross_entropy_loss":
            return "categorical_crossentropy"
        if loss == "log_loss":
            return "binary_crossentropy"
        raise ValueError(f"this loss not defined {loss}")

    def _get_opt(self, optimizer: str, lr: float):
        if optimizer == "adam":
            return "adam"
        raise ValueError(f"this optimizer not defined {optimizer}")

    def _set_loss(self, loss: str):
        self._get_loss(loss)
        self._engine.loss = loss

    def _build_graph(self):
        self.newsencoder = EncoderModel(lambda x: self._engine.encode_news(np.asarray(x, dtype=np.float32).reshape(-1, self._engine.Din)),
                                        "news_encoder")
        self.userencoder = EncoderModel(lambda x: self._engine.encode_users(np.asarray(x, dtype=np.float32)), "user_encoder")
        return TrainModel(self, self._engine.weight_names()), ScorerModel(self)

    def _score_pairs(self, his: np.ndarray, pred_one: np.ndarray) -> torch.Tensor:
        eng = self._engine
        cands = np.asarray(pred_one, dtype=np.float32).reshape(-1, eng.Din)
        his = np.asarray(his, dtype=np.float32)
        if his.shape[0] != cands.shape[0]:
            raise ValueError(f"scorer expects one candidate per history row, got {his.shape} vs {pred_one.shape}")
        his_u, u_inv = dedup_rows(his)
        cand_u, c_inv = dedup_rows(cands)
        return eng.pair_scores(eng.encode_users(his_u), eng.encode_news(cand_u), torch.from_numpy(u_inv).to(eng.device),
                               torch.from_numpy(c_inv).to(eng.device), sigmoid=True)

    def _score_compact(self, his: np.ndarray, cands: np.ndarray, rows: np.ndarray) -> torch.Tensor:
        eng = self._engine
        cand_u, c_inv = dedup_rows(np.asarray(cands, dtype=np.float32))
        user = eng.encode_users(np.asarray(his, dtype=np.float32))
        news = eng.encode_news(cand_u)
        return eng.pair_scores(user, news, torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(eng.device),
                               torch.from_numpy(c_inv).to(eng.device), sigmoid=True)

    def _encode_article_matrix(self, matrix) -> torch.Tensor:
        """news vectors (n_articles+1, E) of every document vector of a loader's matrix, on the device."""
        return self._engine.encode_news(np.asarray(matrix, dtype=np.float32))

    def _score_indexed(self, news_all: torch.Tensor, his_idx, cand_idx, rows) -> torch.Tensor:
        eng = self._engine
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(eng.device)
        hi = dev(np.asarray(his_idx).reshape(-1))
        b, H = len(his_idx), np.asarray(his_idx).shape[1]  # H from the batch: the history-length sweep scores truncated histories
        NEh = torch.empty(b * H, eng.E, device=eng.device)
        _hip.call("ebn_gather_rows_f32", _hip.ptr(hi), _hip.ptr(news_all), _hip.ptr(NEh), b * H, eng.E, news_all.shape[0],
                  None, -1, ctypes.c_float(0.0), None, _hip.stream_handle())
        user = eng.encode_users_from_news(NEh.view(b, H, eng.E))
        return eng.pair_scores(user, news_all, dev(rows), dev(cand_idx), sigmoid=True)

    def train_step(self, his, pred, y):
        return self._engine.train_step(his, pred, y)
