"""Forward parity against a golden file produced by the REAL TensorFlow reference (tools/dump_tf_golden.py).
The file cannot be generated in the build container (no TensorFlow), so these tests skip until someone drops
tests/golden/nrms_tf_golden.npz in; they are the hook that turns "parity unpinned" into pinned."""
from pathlib import Path

import numpy as np
import pytest

from oracle import nrms_numpy as on

GOLD = Path(__file__).parent / "golden" / "nrms_tf_golden.npz"
needs_gold = pytest.mark.skipif(not GOLD.exists(), reason="no TF golden file (run tools/dump_tf_golden.py where TensorFlow exists)")


def _load():
    z = np.load(GOLD)
    V, D, h, d, A = (int(x) for x in z["dims"])
    P = {k: z[f"w{i:02d}"].astype(np.float64) for i, k in enumerate(on.PARAM_ORDER)}
    return z, P, h, d


@needs_gold
def test_oracle_matches_tensorflow_forward():
    z, P, h, d = _load()
    probs, _, _ = on.nrms_forward(z["his"], z["pred"], P, h, d)
    np.testing.assert_allclose(probs, z["probs"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(on.scorer_forward(z["his"], z["pred"][:, :1], P, h, d), z["scorer"], atol=1e-4, rtol=0)
    ne, _ = on.news_encoder_fwd(z["pred"][0], P, h, d)
    np.testing.assert_allclose(ne, z["newsencoder"], atol=1e-4, rtol=0)
    L, _ = on.loss_fwd_bwd(on.nrms_forward(z["his"], z["pred"], P, h, d)[1], z["y"], "cross_entropy_loss")
    assert abs(L - float(z["loss_cross_entropy"])) < 1e-4


@needs_gold
@pytest.mark.gpu
def test_hip_path_matches_tensorflow_forward(hip):
    from ebrec.models.newsrec import NRMSModel
    from ebrec.models.newsrec.model_config import hparams_nrms

    z, P, h, d = _load()
    m = NRMSModel(hparams_nrms, word2vec_embedding=P["emb"]).from_keras_weight_list([z[f"w{i:02d}"] for i in range(13)])
    np.testing.assert_allclose(m.model.predict((z["his"], z["pred"])), z["probs"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(m.scorer.predict((z["his"], z["pred"][:, :1])), z["scorer"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(m.userencoder.predict(z["his"]), z["userencoder"], atol=1e-4, rtol=0)
