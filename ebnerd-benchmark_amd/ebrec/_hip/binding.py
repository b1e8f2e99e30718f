from __future__ import annotations

import ctypes
import os
import re
import threading
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]  # .../ebnerd-benchmark_amd
_REPO_ROOT = _PKG_ROOT.parent


def library_path() -> Path:
    return Path(os.environ.get("EBNERD_HIP_LIB", _PKG_ROOT / "csrc" / "libebnerd_hip.so"))


def header_path() -> Path:
    return _REPO_ROOT / "include" / "ebnerd_hip.h"


class HipError(RuntimeError):
    pass


# ---- struct mirrors of include/ebnerd_hip.h --------------------------------
EBN_N_SITES = 12


class StepState(ctypes.Structure):
    _fields_ = [("step", ctypes.c_uint32), ("seed", ctypes.c_uint32), ("adam_alpha", ctypes.c_float),
                ("lr", ctypes.c_float), ("drop_key", ctypes.c_uint32 * EBN_N_SITES)]


class EncoderDims(ctypes.Structure):
    _fields_ = [("n_seq", ctypes.c_int64), ("L", ctypes.c_int32), ("Din", ctypes.c_int32),
                ("h", ctypes.c_int32), ("d", ctypes.c_int32), ("A", ctypes.c_int32),
                ("drop_site", ctypes.c_int32), ("drop_p", ctypes.c_float)]


class EncoderParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("Wqkv", "W", "b", "q")]


class EncoderActs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("X", "QKV", "Y", "U", "w", "out")]


class EncoderGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("dWqkv", "dW", "db", "dq")]


class EncoderScratch(ctypes.Structure):
    _fields_ = [("dY", ctypes.c_void_p), ("dQKV", ctypes.c_void_p), ("de", ctypes.c_void_p),
                ("partials", ctypes.c_void_p), ("gemm_ws", ctypes.c_void_p),
                ("gemm_ws_floats", ctypes.c_int64)]


class FinishJob(ctypes.Structure):
    """ebn_finish_job: one finishing pass of ebn_grad_finish_f32 (kind 0 split-K sum, 1 column-partials sum, 2 head finish)."""
    _fields_ = [("kind", ctypes.c_int32), ("n_parts", ctypes.c_int32), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64),
                ("partials", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p), ("ld", ctypes.c_int64),
                ("beta", ctypes.c_float), ("scale", ctypes.c_float), ("loss_rows", ctypes.c_void_p), ("loss_out", ctypes.c_void_p)]


FINISH_SPLITK, FINISH_COLRED, FINISH_HEAD, FINISH_MAX_JOBS = 0, 1, 2, 6

DVN_MAX_LAYERS, TN_GROUP_MAX = 4, 8


class DvnArgs(ctypes.Structure):
    """ebn_dvn_args: the fused training step of the NRMSDocVec news encoder (csrc/ebn_docvec.hip)."""
    _fields_ = ([("n_layers", ctypes.c_int32), ("din", ctypes.c_int32), ("e_out", ctypes.c_int32),
                 ("units", ctypes.c_int32 * DVN_MAX_LAYERS), ("n0", ctypes.c_int32), ("n1", ctypes.c_int32),
                 ("drop_p", ctypes.c_float), ("l2", ctypes.c_float),
                 ("W", ctypes.c_void_p * (DVN_MAX_LAYERS + 1)), ("b", ctypes.c_void_p * (DVN_MAX_LAYERS + 1))] +
                [(n, ctypes.c_void_p * DVN_MAX_LAYERS) for n in ("gamma", "beta", "moving_mean", "moving_var")] +
                [("X0", ctypes.c_void_p), ("R", ctypes.c_void_p * DVN_MAX_LAYERS), ("Xn", ctypes.c_void_p * DVN_MAX_LAYERS),
                 ("NE", ctypes.c_void_p), ("stat", ctypes.c_void_p), ("dNE", ctypes.c_void_p),
                 ("dY", ctypes.c_void_p * DVN_MAX_LAYERS), ("dP", ctypes.c_void_p * (DVN_MAX_LAYERS + 1)),
                 ("ggamma", ctypes.c_void_p * DVN_MAX_LAYERS), ("gbeta", ctypes.c_void_p * DVN_MAX_LAYERS),
                 ("loss", ctypes.c_void_p), ("range_flag", ctypes.c_void_p)])


class TnProblem(ctypes.Structure):
    """ebn_tn_problem: one weight-gradient product C = A^T . B of ebn_gemm_tn_group_f32."""
    _fields_ = [("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64), ("A", ctypes.c_void_p),
                ("lda", ctypes.c_int64), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64), ("C", ctypes.c_void_p),
                ("ldc", ctypes.c_int64), ("colsum", ctypes.c_void_p), ("l2_W", ctypes.c_void_p), ("two_lambda", ctypes.c_float)]


DVN_FINALE_MAX_REST = 12
ADAM_FLAT_MAX_REST = 12


class AdamFlat(ctypes.Structure):
    """ebn_adam_flat: the optimizer inside a one-rank step's finishing launch (ebn_grad_finish_adam_f32)."""
    _fields_ = [("theta", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("numel", ctypes.c_int64),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("grad_scale", ctypes.c_float),
                ("n_rest", ctypes.c_int32), ("rest_off", ctypes.c_int64 * ADAM_FLAT_MAX_REST), ("rest_len", ctypes.c_int64 * ADAM_FLAT_MAX_REST)]



class DvnFinale(ctypes.Structure):
    """ebn_dvn_finale: the closing launch of a one-rank NRMSDocVec training step (ebn_dvn_finale_f32)."""
    _fields_ = [("theta", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("numel", ctypes.c_int64),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("grad_scale", ctypes.c_float),
                ("n_rest", ctypes.c_int32), ("rest_off", ctypes.c_int64 * DVN_FINALE_MAX_REST), ("rest_len", ctypes.c_int64 * DVN_FINALE_MAX_REST),
                ("head_partials", ctypes.c_void_p), ("B", ctypes.c_int64), ("A", ctypes.c_int32), ("dq", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("loss_rows", ctypes.c_void_p), ("loss_out", ctypes.c_void_p)]


# ---- header parser -------------------------------------------------------
_PROTO = re.compile(r"^(int64_t|int|const char\*)\s+(ebn_\w+)\s*\(([^;{}]*?)\)\s*;", re.M | re.S)


def _ctype(decl: str):
    decl = decl.strip()
    if decl in ("void", ""):
        return None
    if "*" in decl or decl.startswith("ebn_stream_t"):
        return ctypes.c_void_p
    base = decl.rsplit(" ", 1)[0].replace("const", "").strip()
    return {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "int": ctypes.c_int32,
            "uint32_t": ctypes.c_uint32, "float": ctypes.c_float, "double": ctypes.c_double}[base]


def declared_functions() -> dict:
    """{name: (restype, [argtypes])} for every prototype in ebnerd_hip.h."""
    text = re.sub(r"/\*.*?\*/", "", header_path().read_text(), flags=re.S)
    out = {}
    for ret, name, args in _PROTO.findall(text):
        argt = [t for t in (_ctype(a) for a in args.replace("\n", " ").split(",")) if t is not None]
        res = {"int": ctypes.c_int32, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}[ret]
        out[name] = (res, argt)
    return out


_lib = None
_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    """Load (once) and type the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            path = library_path()
            if not path.exists():
                raise HipError(
                    f"{path} not found: the MI355X HIP extension is not built. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                    "ebnerd-benchmark_amd/csrc`). There is no CPU fallback for the model path.")
            # torch FIRST: it ships its own libamdhip64 and this library must bind to the SAME runtime (its device pointers
            # and streams come from torch).  Loaded before torch, the library pulls in /opt/rocm's copy, torch then brings
            # its own, and every launch fails with hipErrorNoDevice ("no ROCm-capable device is detected").
            import torch  # noqa: F401

            handle = ctypes.CDLL(str(path))
            for name, (res, argt) in declared_functions().items():
                fn = getattr(handle, name)  # AttributeError -> header/library mismatch
                fn.restype = res
                fn.argtypes = argt
            if handle.ebn_abi_version() != 1:
                raise HipError("libebnerd_hip.so ABI version mismatch")
            _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_handle():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class capture:
    """``with capture(graph, pool=None):`` -- torch.cuda.graph(...) in thread_local error mode (the RCCL watchdog thread of
    torch.distributed polls events while this thread captures; in the default "global" mode any such call from another thread
    invalidates the capture) with Python's cyclic garbage collector held off for the duration: a collection that happens to run
    inside the capture can destroy an old CUDAGraph / event / cached block of an earlier model, and the runtime aborts the
    process on such a call from a capturing thread (seen as "Fatal Python error: Aborted ... Garbage-collecting")."""

    def __init__(self, graph, pool=None):
        import torch

        self._cm = torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local")
        self._gc = False

    def __enter__(self):
        import gc

        gc.collect()
        self._gc = gc.isenabled()
        gc.disable()
        try:
            return self._cm.__enter__()
        except BaseException:
            if self._gc:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc

        try:
            return self._cm.__exit__(*exc)
        finally:
            if self._gc:
                gc.enable()


def call(name: str, *args):
    """Invoke an int-returning entry point; raise HipError on a non-zero code."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        msg = handle.ebn_error_string(rc)
        raise HipError(f"{name} failed with code {rc}: {msg.decode() if msg else '?'}")
