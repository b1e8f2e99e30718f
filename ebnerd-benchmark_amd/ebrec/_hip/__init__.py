"""ctypes binding of the gfx950 C-ABI library (include/ebnerd_hip.h).

The prototypes are parsed from the header itself, so the header is the single
source of truth for the boundary.  There is NO CPU fallback: if the library is
missing (``python __graft_entry__.py`` / ``make -C ebnerd-benchmark_amd/csrc``
builds it) or no GPU is visible, callers get a RuntimeError.
"""
from .binding import (  # noqa: F401
    DVN_MAX_LAYERS, TN_GROUP_MAX, DVN_FINALE_MAX_REST, ADAM_FLAT_MAX_REST, AdamFlat, DvnArgs, DvnFinale, TnProblem, EncoderActs, EncoderDims, EncoderGrads, EncoderParams, EncoderScratch, FinishJob, HipError,
    FINISH_COLRED, FINISH_HEAD, FINISH_MAX_JOBS, FINISH_SPLITK,
    StepState, call, capture, declared_functions, header_path, lib, library_path, ptr, stream_handle,
)
