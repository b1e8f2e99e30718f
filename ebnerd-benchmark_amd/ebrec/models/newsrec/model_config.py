"""Hyper-parameter holders for the MI355X NRMS / NRMSDocVec path.

Same contract as the reference's ``model_config.py:82-114``: plain classes whose CLASS
attributes are read by the model (``hparams.title_size`` ...) and may be overwritten in place
by driver scripts (``ebnerd_nrms.py:85-96``); any attribute-bearing object works.  Only the
two model families on the hot path are provided (NPA / LSTUR / NAML are out of scope,
SURVEY.md section 2 rows 6).
"""
from __future__ import annotations

DEFAULT_TITLE_SIZE = 30
DEFAULT_DOCUMENT_SIZE = 768
UNKNOWN_TITLE_VALUE = [0] * DEFAULT_TITLE_SIZE


def hparams_to_dict(hparams_class) -> dict:
    """{annotated attribute: current value} (reference model_config.py:16-20)."""
    return {name: getattr(hparams_class, name) for name in getattr(hparams_class, "__annotations__", {})}


def print_hparams(hparams_class) -> None:
    for name, value in hparams_to_dict(hparams_class).items():
        print(f"{name}: {value}")


class _OptimizerDefaults:
    optimizer: str = "adam"
    loss: str = "cross_entropy_loss"
    dropout: float = 0.2
    learning_rate: float = 1e-4


class hparams_nrms(_OptimizerDefaults):
    __annotations__ = {
        "title_size": int, "history_size": int, "head_num": int, "head_dim": int,
        "attention_hidden_dim": int, "optimizer": str, "loss": str, "dropout": float,
        "learning_rate": float, "newsencoder_units_per_layer": list, "newsencoder_l2_regularization": float,
    }
    # input dimensions
    title_size = DEFAULT_TITLE_SIZE
    history_size = 20
    # architecture
    head_num = 20
    head_dim = 20
    attention_hidden_dim = 200
    # optional per-position MLP between self-attention and additive attention (nrms.py:142-152)
    newsencoder_units_per_layer = None
    newsencoder_l2_regularization = 1e-4


class hparams_nrms_docvec(_OptimizerDefaults):
    __annotations__ = dict(hparams_nrms.__annotations__)
    title_size = DEFAULT_DOCUMENT_SIZE  # width of the pre-computed document vector
    history_size = 20
    head_num = 16
    head_dim = 16
    attention_hidden_dim = 200
    newsencoder_units_per_layer = [512, 512, 512]
    newsencoder_l2_regularization = 1e-4
