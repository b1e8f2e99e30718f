"""NRMSDocVec on EB-NeRD with the MI355X-native model (reference: examples/reproducibility_scripts/
ebnerd_nrms_docvec.py): the article representation is a pre-computed document vector per article
(``--document_embeddings``: parquet with ``article_id`` + one vector column) instead of title tokens; everything
after that -- sampling, loaders, callbacks, chunked test prediction, submission -- is ebnerd_nrms.run()."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent))
import ebnerd_nrms  # noqa: E402  (also puts the package on sys.path)
from args_nrms import get_args  # noqa: E402
from ebrec.models.newsrec import NRMSDocVec  # noqa: E402
from ebrec.models.newsrec.model_config import hparams_nrms_docvec  # noqa: E402
from ebrec.utils._articles import create_article_id_to_value_mapping  # noqa: E402


def prepare(args):
    """Document-vector mapping and hparams from the command line (shared with ebnerd_nrms_doc_hist.py)."""
    PATH = Path(args.data_path).expanduser()
    df_vec = pd.read_parquet(PATH / args.document_embeddings)
    vec_col = [c for c in df_vec.columns if c != "article_id"][0]
    df_vec[vec_col] = [np.asarray(v, dtype=np.float32) for v in df_vec[vec_col]]
    article_mapping = create_article_id_to_value_mapping(df_vec, value_col=vec_col)
    hparams = hparams_nrms_docvec
    hparams.title_size = len(next(iter(article_mapping.values())))
    hparams.history_size = args.history_size
    hparams.head_num, hparams.head_dim, hparams.attention_hidden_dim = args.head_num, args.head_dim, args.attention_hidden_dim
    hparams.optimizer, hparams.loss, hparams.dropout, hparams.learning_rate = args.optimizer, args.loss, args.dropout, args.learning_rate
    hparams.newsencoder_units_per_layer = args.newsencoder_units_per_layer
    hparams.newsencoder_l2_regularization = args.newsencoder_l2_regularization
    return hparams, article_mapping


def main(argv=None):
    args = get_args(argv, docvec=True)
    hparams, article_mapping = prepare(args)
    return ebnerd_nrms.run(args, hparams, lambda: NRMSDocVec(hparams=hparams, seed=42), article_mapping, NRMSDocVec.__name__)


if __name__ == "__main__":
    main()
