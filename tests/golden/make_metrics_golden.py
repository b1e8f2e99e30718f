"""Generates tests/golden/metrics_golden.json by running the REFERENCE's own evaluator
(/root/reference/src/ebrec/evaluation, importable without TF/polars) on seeded ragged impressions.
Run in the build container only; the reference never travels, the JSON (inputs + expected outputs) does."""
import json
import sys
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from ebrec.evaluation import (AccuracyScore, AucScore, F1Score, LogLossScore, MetricEvaluator, MrrScore, NdcgScore,  # noqa: E402
                              RootMeanSquaredError)
from ebrec.evaluation.metrics import (auc_score_custom, dcg_score, mrr_score, ndcg_score, reciprocal_rank_score,  # noqa: E402
                                      roc_auc_score)

rng = np.random.default_rng(20240928)
labels, preds = [], []
for i in range(300):
    n = int(rng.integers(2, 40)) if i % 17 else 250  # inview lengths 2..39, a few beyond-accuracy sized (250)
    y = np.zeros(n, dtype=int)
    y[rng.choice(n, size=int(rng.integers(1, min(3, n - 1) + 1)), replace=False)] = 1
    p = rng.random(n)
    if i % 3 == 0:  # ties, incl. ties between a positive and a negative
        p = np.round(p, 1)
    if i % 50 == 0:
        p[:] = 0.5
    labels.append(y.tolist())
    preds.append(p.tolist())

metric_functions = [AucScore(), MrrScore(), NdcgScore(k=5), NdcgScore(k=10), LogLossScore(), RootMeanSquaredError(),
                    AccuracyScore(threshold=0.5), F1Score(threshold=0.5)]
ev = MetricEvaluator(labels=labels, predictions=preds, metric_functions=metric_functions).evaluate()
per_row = {"roc_auc": [], "auc_custom": [], "mrr": [], "rr": [], "ndcg5": [], "ndcg10": [], "dcg10": []}
for y, p in zip(labels, preds):
    y, p = np.array(y), np.array(p)
    per_row["roc_auc"].append(float(roc_auc_score(y, p)))
    per_row["auc_custom"].append(float(auc_score_custom(y, p)))
    per_row["mrr"].append(float(mrr_score(y, p)))
    per_row["rr"].append(float(reciprocal_rank_score(y, p)))
    per_row["ndcg5"].append(float(ndcg_score(y, p, 5)))
    per_row["ndcg10"].append(float(ndcg_score(y, p, 10)))
    per_row["dcg10"].append(float(dcg_score(y, p, 10)))
doc = MetricEvaluator(labels=[[1, 0, 0], [1, 1, 0], [1, 0, 0, 0]],
                      predictions=[[0.2, 0.3, 0.5], [0.18, 0.7, 0.1], [0.18, 0.2, 0.1, 0.1]],
                      metric_functions=[AucScore(), MrrScore(), NdcgScore(k=5), NdcgScore(k=10), LogLossScore(),
                                        RootMeanSquaredError(), AccuracyScore(threshold=0.5), F1Score(threshold=0.5)]).evaluate()
out = {"source": "ebanalyse/ebnerd-benchmark src/ebrec/evaluation run in the build container (numpy %s)" % np.__version__,
       "labels": labels, "predictions": preds, "evaluations": ev.evaluations, "per_row": per_row,
       "docstring_example": doc.evaluations}
Path(__file__).with_name("metrics_golden.json").write_text(json.dumps(out))
print(ev.evaluations)
print(doc.evaluations)
