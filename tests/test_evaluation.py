"""ebrec.evaluation (MI355X repo) against golden vectors produced by the REFERENCE's own evaluator
(tests/golden/make_metrics_golden.py ran /root/reference/src/ebrec/evaluation in the build container).
Pure host logic -- runs without a GPU."""
import json
from pathlib import Path

import numpy as np
import pytest

from ebrec.evaluation import (AccuracyScore, AucScore, F1Score, LogLossScore, MetricEvaluator, MrrScore, NdcgScore,
                              RootMeanSquaredError)
from ebrec.evaluation.metrics import (auc_score_custom, dcg_score, mrr_score, ndcg_score, reciprocal_rank_score,
                                      roc_auc_score)

G = json.loads((Path(__file__).parent / "golden" / "metrics_golden.json").read_text())


def all_metrics():
    return [AucScore(), MrrScore(), NdcgScore(k=5), NdcgScore(k=10), LogLossScore(), RootMeanSquaredError(),
            AccuracyScore(threshold=0.5), F1Score(threshold=0.5)]


def test_docstring_known_answers():
    """metrics_protocols.py:143-170 -- the only known-answer vector in the reference."""
    ev = MetricEvaluator(labels=[[1, 0, 0], [1, 1, 0], [1, 0, 0, 0]],
                         predictions=[[0.2, 0.3, 0.5], [0.18, 0.7, 0.1], [0.18, 0.2, 0.1, 0.1]],
                         metric_functions=all_metrics()).evaluate()
    want = {"auc": 0.5555555555555556, "mrr": 0.5277777777777778, "ndcg@5": 0.7103099178571526,
            "ndcg@10": 0.7103099178571526, "logloss": 0.716399020295845, "rmse": 0.5022870658128165,
            "accuracy": 0.5833333333333334, "f1": 0.2222222222222222}
    assert ev.evaluations.keys() == want.keys()
    for k, v in want.items():
        assert ev.evaluations[k] == pytest.approx(v, rel=1e-12), k
    assert G["docstring_example"] == pytest.approx(want, rel=1e-12)


def test_evaluator_matches_reference_on_300_ragged_impressions():
    ev = MetricEvaluator(labels=G["labels"], predictions=G["predictions"], metric_functions=all_metrics()).evaluate()
    for k, v in G["evaluations"].items():
        assert ev.evaluations[k] == pytest.approx(v, rel=1e-12, abs=1e-15), k


def test_per_impression_functions_match_reference_including_ties():
    fns = {"roc_auc": roc_auc_score, "auc_custom": auc_score_custom, "mrr": mrr_score, "rr": reciprocal_rank_score,
           "ndcg5": lambda y, p: ndcg_score(y, p, 5), "ndcg10": lambda y, p: ndcg_score(y, p, 10),
           "dcg10": lambda y, p: dcg_score(y, p, 10)}
    for name, fn in fns.items():
        got = [float(fn(np.array(y), np.array(p))) for y, p in zip(G["labels"], G["predictions"])]
        assert got == pytest.approx(G["per_row"][name], rel=1e-12, abs=1e-15), name


def test_threshold_metrics_binarise_in_place_like_the_reference():
    """utils.py:6-10 mutates an ndarray argument; a list argument is left alone."""
    p = [np.array([0.2, 0.7]), np.array([0.6, 0.1])]
    AccuracyScore(0.5)([[0, 1], [1, 0]], p)
    assert p[0].tolist() == [0.0, 1.0] and p[1].tolist() == [1.0, 0.0]
    q = [[0.2, 0.7]]
    AccuracyScore(0.5)([[0, 1]], q)
    assert q == [[0.2, 0.7]]


def test_auc_needs_both_classes_and_evaluator_rejects_non_callables():
    with pytest.raises(ValueError):
        AucScore()([[0, 0, 0]], [[0.1, 0.2, 0.3]])
    with pytest.raises(TypeError):
        MetricEvaluator([[1, 0]], [[0.3, 0.2]], metric_functions=[AucScore(), "mrr"])
    ev = MetricEvaluator([[1, 0]], [[0.3, 0.2]], metric_functions=[AucScore()])
    assert "{}" in str(ev) and ev.evaluate() is ev and "auc" in str(ev)


def test_metric_is_a_structural_type_and_prints_like_the_reference():
    """protocols.py:5-17: `name` + `calculate` make a metric (MetricLike, structural); `Metric` carries the reference's default
    call forwarding and printed form, so a metric written the reference's way -- `class Foo(Metric)` with only `calculate` --
    is callable and the evaluator takes it (round-5 ADVICE: it was rejected as "not callable")."""
    from ebrec.evaluation.protocols import Metric, MetricBase, MetricLike

    class Foo(Metric):  # the reference's own idiom (metrics_protocols.py:74-86 derive from Metric and define calculate only)
        def __init__(self):
            self.name = "foo"

        def calculate(self, y_true, y_score):
            return 0.25

    assert Foo()([[1, 0]], [[0.3, 0.2]]) == 0.25 and str(Foo()) == "<Callable Metric: foo>: params: {'name': 'foo'}" == repr(Foo())
    assert MetricEvaluator([[1, 0]], [[0.3, 0.2]], [Foo()]).evaluate().evaluations == {"foo": 0.25}

    class Mine:  # a third-party metric object: no base class
        name = "mine"

        def calculate(self, y_true, y_score):
            return 1.0

    assert isinstance(Mine(), MetricLike) and isinstance(NdcgScore(5), MetricLike) and isinstance(Foo(), MetricLike) and not isinstance(object(), MetricLike)
    assert isinstance(AucScore(), MetricBase) and isinstance(AucScore(), Metric)
    assert str(NdcgScore(k=5)) == "<Callable Metric: ndcg@5>: params: {'k': 5, 'name': 'ndcg@5'}" == repr(NdcgScore(k=5))
    with pytest.raises(TypeError, match=r"not callable: \[\]"):
        MetricEvaluator([[1, 0]], [[0.3, 0.2]], metric_functions=[])  # the reference refuses an empty list too
    with pytest.raises(TypeError, match="not callable: .*str"):
        MetricEvaluator([[1, 0]], [[0.3, 0.2]], metric_functions=[AucScore()]).metric_functions = [AucScore(), "mrr"]
    ev = MetricEvaluator([[1, 0]], [[0.3, 0.2]], metric_functions=[AucScore()])
    assert str(ev) == "<MetricEvaluator class>: {}"
    assert str(ev.evaluate()) == '<MetricEvaluator class>: \n {\n    "auc": 1.0\n}'
