from ._classification import auc_score_custom  # noqa: F401
from ._ranking import dcg_score, mrr_score, ndcg_score, reciprocal_rank_score  # noqa: F401
from ._sklearn import accuracy_score, f1_score, log_loss, mean_squared_error, roc_auc_score  # noqa: F401
