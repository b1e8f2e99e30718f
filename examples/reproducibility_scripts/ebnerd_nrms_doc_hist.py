"""AUC as a function of the history length, NRMSDocVec on EB-NeRD with the MI355X-native model (reference:
examples/reproducibility_scripts/ebnerd_nrms_doc_hist.py).

Train ONE NRMSDocVec exactly like ebnerd_nrms_docvec.py does (reference lines 1-248 are that script again), then
(lines 250-300) score the validation users with long histories on histories TRUNCATED to 1, 2, ... 50 entries and write
``auc_history_length.json``.  The user encoder has no weight that depends on the history length (SelfAttention +
AttLayer2, layers.py:200-254, 55-81), so one trained model scores every length: the scorer takes H from the batch.

    python ebnerd_nrms_doc_hist.py --data_path ~/ebnerd_data --datasplit ebnerd_small --document_embeddings <parquet>
"""
from __future__ import annotations

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import ebnerd_nrms  # noqa: E402  (also puts the package on sys.path)
import ebnerd_nrms_docvec  # noqa: E402
from args_nrms import get_args  # noqa: E402
from ebrec.evaluation import AucScore, MetricEvaluator  # noqa: E402
from ebrec.models.newsrec import NRMSDocVec  # noqa: E402
from ebrec.models.newsrec.dataloader import NRMSDataLoader  # noqa: E402
from ebrec.utils._behaviors import add_prediction_scores, create_binary_labels_column, ebnerd_from_path, truncate_history  # noqa: E402
from ebrec.utils._constants import DEFAULT_HISTORY_ARTICLE_ID_COL, DEFAULT_LABELS_COL  # noqa: E402
from ebrec.utils._python import write_json_file  # noqa: E402

# (history size, eval batch size) pairs of the reference sweep, ebnerd_nrms_doc_hist.py:270-286
PAIRS = [(1, 256), (2, 256), (3, 256), (4, 256), (5, 256), (6, 256), (7, 256), (8, 256), (9, 256), (10, 256), (15, 128), (20, 128),
         (30, 64), (40, 64), (50, 64)]
LOAD_HISTORY = 120  # the validation histories are loaded 120 deep and un-padded (reference line 260)


def history_length_sweep(model, ctx, pairs=PAIRS):
    """reference lines 250-300: {history size: validation AUC of the trained model on histories cut to that size}."""
    args, PATH = ctx["args"], ctx["PATH"]
    df = ebnerd_from_path(PATH / ctx["DATASPLIT"] / "validation", history_size=LOAD_HISTORY, padding=None)
    df = df[df[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()].sample(frac=args.fraction_test, random_state=ctx["SEED"])
    df = df[[len(h) >= args.filter_min_history for h in df[DEFAULT_HISTORY_ARTICLE_ID_COL]]][ctx["COLUMNS"]]
    df = create_binary_labels_column(df).reset_index(drop=True)
    if not len(df):
        raise ValueError(f"no validation user has {args.filter_min_history} or more history entries (--filter_min_history)")
    results = {}
    for hist_size, batch_size in pairs:
        print(f"History size: {hist_size}, Batch size: {batch_size}")
        df_ = truncate_history(df, column=DEFAULT_HISTORY_ARTICLE_ID_COL, history_size=hist_size, padding_value=0, enable_warning=False)
        loader = NRMSDataLoader(behaviors=df_, article_dict=ctx["article_mapping"], unknown_representation="zeros",
                                history_column=DEFAULT_HISTORY_ARTICLE_ID_COL, eval_mode=True, batch_size=batch_size)
        scores = model.scorer.predict(loader)
        df_pred = add_prediction_scores(df_, scores.tolist())
        ev = MetricEvaluator(labels=df_pred[DEFAULT_LABELS_COL].tolist(), predictions=df_pred["scores"].tolist(),
                             metric_functions=[AucScore()]).evaluate()
        auc = ev.evaluations["auc"]
        results[hist_size] = round(float(auc), 4)
        print(f"{auc} (History size: {hist_size}, Batch size: {batch_size})")
    for h, a in results.items():
        print(f"({a}, {h}),")
    if ctx["rank"] == 0:
        write_json_file(results, ctx["ARTIFACT_DIR"] / "auc_history_length.json")
    return results


def main(argv=None):
    args = get_args(argv, docvec=True)
    hparams, article_mapping = ebnerd_nrms_docvec.prepare(args)
    return ebnerd_nrms.run(args, hparams, lambda: NRMSDocVec(hparams=hparams, seed=42), article_mapping, NRMSDocVec.__name__,
                           after_validation=history_length_sweep)


if __name__ == "__main__":
    main()
