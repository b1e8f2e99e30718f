"""Pin model parity against the REAL reference (needs TensorFlow 2.12-2.15 and the reference checkout; neither
exists in the build container, which is why oracle/ is "parity unpinned" for the model math).

Run this somewhere with TF:

    PYTHONPATH=<ebnerd-benchmark>/src python tools/dump_tf_golden.py tests/golden/nrms_tf_golden.npz

ONE run pins everything the oracle tags [KERAS-SEMANTICS]:
  * forward of ``model.model`` / ``model.scorer`` / ``newsencoder`` / ``userencoder`` on seeded inputs (dropout off),
    with the 13 weight arrays in ``get_weights()`` order and their Keras variable names       -> layers.py:55-81,200-254
  * ``evaluate`` under both compiled losses (cross_entropy_loss, log_loss)                      -> loss semantics (BCE on
    the softmax's cached logits or on clipped probabilities: SURVEY A.5 vs oracle/nrms_numpy.py:386-398)
  * the weights after 3 ``train_on_batch`` steps with dropout = 0, for both losses             -> Adam's eps placement,
    dense moment decay of untouched embedding rows, gradient scaling of the mean
  * NRMSDocVec: ``get_weights()`` order + names, inference forward, one ``train_on_batch`` step (dropout 0): loss incl.
    the L2 term, weights and BatchNorm moving statistics afterwards                            -> per-call-site batch
    statistics and two moving-average updates per step
  * NRMS with ``newsencoder_units_per_layer``: ``get_weights()`` order + names and a forward pass
  * NRMSDocVec under ``tf.distribute.MirroredStrategy`` on TWO logical CPU devices (round 5): one ``fit`` step of a global batch
    of 2 x B rows -- the loss and the BatchNorm moving statistics afterwards                    -> the data-parallel BatchNorm
    contract of this build (per-replica batch statistics, moving statistics read as the MEAN over the replicas:
    ``MLPStack.sync_moving_statistics``); skipped (fields absent) where the logical-device split is refused
``tests/test_tf_golden.py`` checks the float64 oracle (CPU) and the HIP path (GPU) against every field at 1e-4 -- the
forward-parity experiment of BASELINE.json's north_star, widened to training.  Only data is written; no reference
source travels.
"""
import sys

import numpy as np


def _names(model):
    return np.array([w.name for w in model.weights])


def main(out_path):
    import tensorflow as tf  # noqa: F401
    from ebrec.models.newsrec import NRMSDocVec, NRMSModel
    from ebrec.models.newsrec.model_config import hparams_nrms, hparams_nrms_docvec

    rng = np.random.default_rng(2024)
    V, D, B, C = 1000, 300, 8, 5
    hp = hparams_nrms
    hp.dropout = 0.0  # training-mode fields below must not depend on TF's RNG stream
    hp.newsencoder_units_per_layer = None
    emb = rng.standard_normal((V, D)).astype(np.float32) * 0.1
    his = rng.integers(0, V, (B, hp.history_size, hp.title_size)).astype(np.int32)
    his[0, :4] = 0  # padded history slots
    pred = rng.integers(0, V, (B, C, hp.title_size)).astype(np.int32)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    out = {"his": his, "pred": pred, "y": y, "dims": np.array([V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim]),
           "learning_rate": np.array(hp.learning_rate)}
    w0 = None
    for loss in ("cross_entropy_loss", "log_loss"):
        hp.loss = loss
        model = NRMSModel(hparams=hp, word2vec_embedding=emb, seed=42)
        if w0 is None:
            # break the WQ=WK=WV symmetry of the seeded initialisers so that a Q/K/V mix-up cannot hide
            w0 = [a + 0.05 * rng.standard_normal(a.shape).astype(np.float32) for a in model.model.get_weights()]
            w0[0] = emb
            out["weight_names"] = _names(model.model)
            for i, a in enumerate(w0):
                out[f"w{i:02d}"] = a
        model.model.set_weights(w0)
        if loss == "cross_entropy_loss":
            out["probs"] = model.model.predict((his, pred), verbose=0)
            out["scorer"] = model.scorer.predict((his, pred[:, :1]), verbose=0)
            out["newsencoder"] = model.newsencoder.predict(pred[0], verbose=0)
            out["userencoder"] = model.userencoder.predict(his, verbose=0)
        out[f"loss_{loss}"] = np.array(model.model.evaluate((his, pred), y, verbose=0))
        losses = [float(np.ravel(model.model.train_on_batch((his, pred), y))[0]) for _ in range(3)]
        out[f"train3_losses_{loss}"] = np.array(losses)
        for i, a in enumerate(model.model.get_weights()):
            out[f"train3_{loss}_w{i:02d}"] = a

    # ---- NRMS with the optional per-token Dense/BN stack (nrms.py:142-152)
    hp.loss, hp.newsencoder_units_per_layer = "cross_entropy_loss", [hp.head_num * hp.head_dim]
    m_units = NRMSModel(hparams=hp, word2vec_embedding=emb, seed=42)
    out["units_weight_names"] = _names(m_units.model)
    for i, a in enumerate(m_units.model.get_weights()):
        out[f"units_w{i:02d}"] = a
    out["units_probs"] = m_units.model.predict((his, pred), verbose=0)
    hp.newsencoder_units_per_layer = None

    # ---- NRMSDocVec (nrms_docvec.py)
    hd = hparams_nrms_docvec
    hd.dropout = 0.0
    hd.newsencoder_units_per_layer = [64, 48]
    hd.title_size, hd.history_size = 40, 6
    dv = NRMSDocVec(hparams=hd, seed=42)
    dhis = rng.standard_normal((B, hd.history_size, hd.title_size)).astype(np.float32)
    dhis[0, :2] = 0
    dpred = rng.standard_normal((B, C, hd.title_size)).astype(np.float32)
    wd = [a + 0.05 * rng.standard_normal(a.shape).astype(np.float32) for a in dv.model.get_weights()]
    wd = [np.abs(a) + 0.5 if "moving_variance" in n else a for a, n in zip(wd, _names(dv.model))]
    dv.model.set_weights(wd)
    out.update({"docvec_his": dhis, "docvec_pred": dpred, "docvec_weight_names": _names(dv.model),
                "docvec_dims": np.array([hd.title_size, hd.head_num, hd.head_dim, hd.attention_hidden_dim, hd.history_size]),
                "docvec_units": np.array(hd.newsencoder_units_per_layer), "docvec_l2": np.array(hd.newsencoder_l2_regularization),
                "docvec_learning_rate": np.array(hd.learning_rate)})
    for i, a in enumerate(wd):
        out[f"docvec_w{i:02d}"] = a
    out["docvec_probs"] = dv.model.predict((dhis, dpred), verbose=0)
    out["docvec_train1_loss"] = np.array(float(np.ravel(dv.model.train_on_batch((dhis, dpred), y))[0]))
    for i, a in enumerate(dv.model.get_weights()):
        out[f"docvec_train1_w{i:02d}"] = a
    # ---- NRMSDocVec, two replicas (data parallel): what do the BatchNorm moving statistics hold after one step?
    try:
        cpus = tf.config.list_physical_devices("CPU")
        tf.config.set_logical_device_configuration(cpus[0], [tf.config.LogicalDeviceConfiguration(), tf.config.LogicalDeviceConfiguration()])
        strategy = tf.distribute.MirroredStrategy(["CPU:0", "CPU:1"])
        with strategy.scope():
            dv2 = NRMSDocVec(hparams=hd, seed=42)
        dv2.model.set_weights(wd)
        his2 = np.concatenate([dhis, rng.standard_normal(dhis.shape).astype(np.float32)])
        pred2 = np.concatenate([dpred, rng.standard_normal(dpred.shape).astype(np.float32)])
        y2 = np.concatenate([y, np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]])
        ds = tf.data.Dataset.from_tensor_slices(((his2, pred2), y2)).batch(2 * B)  # one global batch: B rows per replica, in order
        h2 = dv2.model.fit(ds, epochs=1, verbose=0)
        out.update({"docvec_dp2_his": his2, "docvec_dp2_pred": pred2, "docvec_dp2_y": y2, "docvec_dp2_loss": np.array(h2.history["loss"][0])})
        for i, a in enumerate(dv2.model.get_weights()):
            out[f"docvec_dp2_w{i:02d}"] = a
    except Exception as e:  # logical devices must be configured before TF initialises its runtime: run this script in a fresh process
        print("two-replica NRMSDocVec step skipped:", type(e).__name__, e)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/nrms_tf_golden.npz")
