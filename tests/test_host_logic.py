"""Host-side control logic that needs no GPU: callbacks (Keras semantics), streaming AUC, hparams, and the
loud failure of the model path without a GPU / without the HIP library."""
import numpy as np
import pytest

from ebrec.evaluation.metrics import roc_auc_score
from ebrec.models.newsrec.callbacks import (EarlyStopping, History, ModelCheckpoint, ReduceLROnPlateau, StreamingAUC,
                                            TensorBoard)
from ebrec.models.newsrec.model_config import hparams_nrms, hparams_nrms_docvec, hparams_to_dict, print_hparams


class FakeOpt:
    learning_rate = 1e-3


class FakeModel:
    def __init__(self):
        self.optimizer = FakeOpt()
        self.stop_training = False
        self.w = [np.zeros(2)]
        self.saved = []

    def get_weights(self):
        return [w.copy() for w in self.w]

    def set_weights(self, w):
        self.w = [x.copy() for x in w]

    def save_weights(self, path):
        self.saved.append(path)


def run(cb, values, key="val_auc"):
    m = FakeModel()
    cb.set_model(m)
    cb.on_train_begin()
    for epoch, v in enumerate(values):
        m.w = [np.full(2, float(epoch))]
        cb.on_epoch_end(epoch, {key: v})
        if m.stop_training:
            break
    cb.on_train_end()
    return m, epoch


def test_early_stopping_restores_best_weights():
    m, last = run(EarlyStopping(monitor="val_auc", mode="max", patience=2, restore_best_weights=True), [0.5, 0.6, 0.55, 0.58, 0.7])
    assert last == 3 and m.stop_training and m.w[0].tolist() == [1.0, 1.0]  # best epoch was 1
    m, last = run(EarlyStopping(monitor="val_loss", patience=1), [1.0, 0.9, 0.8], key="val_loss")
    assert not m.stop_training and last == 2


def test_model_checkpoint_saves_only_improvements():
    cb = ModelCheckpoint(filepath="w_{epoch}.h5", monitor="val_auc", mode="max", save_best_only=True, save_weights_only=True)
    m, _ = run(cb, [0.5, 0.4, 0.6])
    assert m.saved == ["w_1.h5", "w_3.h5"]
    m, _ = run(ModelCheckpoint(filepath="w.h5"), [0.5, 0.4], key="val_loss")
    assert m.saved == ["w.h5", "w.h5"]


def test_reduce_lr_on_plateau_like_the_driver():  # ebnerd_nrms.py:230-236
    cb = ReduceLROnPlateau(monitor="val_auc", mode="max", factor=0.2, patience=2, min_lr=1e-6)
    m, _ = run(cb, [0.5, 0.5, 0.5, 0.5, 0.5])
    assert m.optimizer.learning_rate == pytest.approx(1e-3 * 0.2 * 0.2)
    with pytest.raises(ValueError):
        ReduceLROnPlateau(factor=1.0)


def test_history_and_tensorboard(tmp_path):
    h = History()
    run(h, [0.1, 0.2])
    assert h.history["val_auc"] == [0.1, 0.2] and h.epoch == [0, 1]
    tb = TensorBoard(log_dir=tmp_path / "tb", histogram_freq=1)
    run(tb, [0.3])
    assert "val_auc" in (tmp_path / "tb" / "scalars.jsonl").read_text()


def test_streaming_auc_tracks_exact_auc_and_device_path_matches_numpy():
    import torch

    rng = np.random.default_rng(0)
    y = (rng.random(5000) < 0.2).astype(np.int8)
    p = np.clip(rng.normal(0.4 + 0.15 * y, 0.15), 0, 1).astype(np.float32)
    a = StreamingAUC()
    for s in range(0, 5000, 160):  # one update per "step"
        a.update_numpy(y[s:s + 160], p[s:s + 160])
    exact = roc_auc_score(y, p)
    assert abs(a.result() - exact) < 2e-3  # 200-threshold trapezoid approximation
    b = StreamingAUC()
    b.update_device(torch.from_numpy(y), torch.from_numpy(p))
    assert b.result() == pytest.approx(a.result(), abs=1e-12)
    c = StreamingAUC()
    c.update_numpy([0, 1], [0.0, 1.0])  # the +-1e-7 end thresholds keep exact 0/1 predictions inside
    assert c.result() == pytest.approx(1.0)


def test_hparams_surface():
    d = hparams_to_dict(hparams_nrms)
    assert d == {"title_size": 30, "history_size": 20, "head_num": 20, "head_dim": 20, "attention_hidden_dim": 200,
                 "optimizer": "adam", "loss": "cross_entropy_loss", "dropout": 0.2, "learning_rate": 1e-4,
                 "newsencoder_units_per_layer": None, "newsencoder_l2_regularization": 1e-4}
    dv = hparams_to_dict(hparams_nrms_docvec)
    assert dv["title_size"] == 768 and dv["head_num"] == 16 and dv["newsencoder_units_per_layer"] == [512, 512, 512]
    hparams_nrms.history_size = 50  # drivers mutate the class attributes in place (ebnerd_nrms.py:85-96)
    assert hparams_to_dict(hparams_nrms)["history_size"] == 50
    hparams_nrms.history_size = 20
    print_hparams(hparams_nrms)


def test_model_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from ebrec.models.newsrec import NRMSModel

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NRMSModel(hparams_nrms, seed=1)

    class Bad(hparams_nrms):
        loss = "hinge"

    with pytest.raises(ValueError, match="this loss not defined hinge"):
        NRMSModel(Bad)

    class BadOpt(hparams_nrms):
        optimizer = "sgd"

    with pytest.raises(ValueError, match="this optimizer not defined sgd"):
        NRMSModel(BadOpt)


def test_missing_library_raises(monkeypatch, tmp_path):
    from ebrec._hip import binding

    monkeypatch.setattr(binding, "_lib", None)
    monkeypatch.setenv("EBNERD_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(binding.HipError, match="not built"):
        binding.lib()
    monkeypatch.delenv("EBNERD_HIP_LIB")
    monkeypatch.setattr(binding, "_lib", None)
    assert binding.lib() is not None


def test_bce_default_is_the_one_the_tf_golden_tests_expect():
    """tests/test_tf_golden.py flips DEFAULT_BCE_ON and the engines' default together."""
    import inspect

    from ebrec.models.newsrec._engine import NRMSEngine, loss_kind_of
    from ebrec.models.newsrec._engine_docvec import DocVecEngine
    from tests.test_tf_golden import DEFAULT_BCE_ON

    for cls in (NRMSEngine, DocVecEngine):
        assert inspect.signature(cls.__init__).parameters["bce_on"].default == DEFAULT_BCE_ON
    assert [loss_kind_of("cross_entropy_loss", "probs"), loss_kind_of("log_loss", "logits"), loss_kind_of("log_loss", "probs")] == [0, 1, 2]
    with pytest.raises(ValueError):
        loss_kind_of("log_loss", "sigmoid")
