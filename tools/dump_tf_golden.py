"""Pin model parity against the REAL reference (needs TensorFlow 2.12-2.15 and the reference checkout; neither
exists in the build container, which is why oracle/ is "parity unpinned" for the model math).

Run this somewhere with TF:

    PYTHONPATH=<ebnerd-benchmark>/src python tools/dump_tf_golden.py tests/golden/nrms_tf_golden.npz

It builds the reference's NRMSModel, runs ``model.model`` / ``model.scorer`` / ``newsencoder`` / ``userencoder`` on
seeded inputs with dropout off, and stores inputs, the 13 weight arrays (``model.model.get_weights()`` order, SURVEY.md
A.6) and outputs.  ``tests/test_tf_golden.py`` then checks the float64 oracle (CPU) and the HIP path (GPU) against
it at 1e-4 -- the forward-parity experiment of BASELINE.json's north_star.  Only data is written; no reference
source travels.
"""
import sys

import numpy as np


def main(out_path):
    import tensorflow as tf  # noqa: F401
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.model_config import hparams_nrms

    rng = np.random.default_rng(2024)
    V, D, B, C = 1000, 300, 8, 5
    hp = hparams_nrms
    emb = rng.standard_normal((V, D)).astype(np.float32) * 0.1
    model = NRMSModel(hparams=hp, word2vec_embedding=emb, seed=42)
    # break the WQ=WK=WV symmetry of the seeded initialisers so that a Q/K/V mix-up cannot hide
    w = [a + 0.05 * rng.standard_normal(a.shape).astype(np.float32) for a in model.model.get_weights()]
    w[0] = emb
    model.model.set_weights(w)
    his = rng.integers(0, V, (B, hp.history_size, hp.title_size)).astype(np.int32)
    his[0, :4] = 0  # padded history slots
    pred = rng.integers(0, V, (B, C, hp.title_size)).astype(np.int32)
    out = {"his": his, "pred": pred, "dims": np.array([V, D, hp.head_num, hp.head_dim, hp.attention_hidden_dim])}
    for i, a in enumerate(model.model.get_weights()):
        out[f"w{i:02d}"] = a
    out["probs"] = model.model.predict((his, pred), verbose=0)
    out["scorer"] = model.scorer.predict((his, pred[:, :1]), verbose=0)
    out["newsencoder"] = model.newsencoder.predict(pred[0], verbose=0)
    out["userencoder"] = model.userencoder.predict(his, verbose=0)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    out["y"] = y
    out["loss_cross_entropy"] = np.array(model.model.evaluate((his, pred), y, verbose=0))
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/nrms_tf_golden.npz")
