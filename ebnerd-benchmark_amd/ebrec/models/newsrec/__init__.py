"""MI355X-native NRMS / NRMSDocVec (NPA, LSTUR and NAML of the reference are out of scope)."""
from .model_config import hparams_nrms, hparams_nrms_docvec, hparams_to_dict, print_hparams  # noqa: F401


def __getattr__(name):  # lazy: importing model_config / dataloader must not need torch or a GPU
    if name == "NRMSModel":
        from .nrms import NRMSModel
        return NRMSModel
    if name == "NRMSDocVec":
        from .nrms_docvec import NRMSDocVec
        return NRMSDocVec
    raise AttributeError(name)
