"""ctypes binding of libmrag_hip.so (include/mrag.h).  There is no CPU fallback: a missing
library or a missing GPU raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

MRAG_F32, MRAG_F16, MRAG_BF16, MRAG_F64 = 0, 1, 2, 3
METRIC_COSINE, METRIC_IP = 0, 1
POOL_MEAN, POOL_CLS = 0, 1

ERR_NAMES = {-1: "MRAG_ERR_INVALID", -2: "MRAG_ERR_NO_DEVICE", -3: "MRAG_ERR_HIP", -4: "MRAG_ERR_OOM",
             -5: "MRAG_ERR_UNSUPPORTED"}


class MragError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class MragLibraryMissing(ImportError):
    pass


class EncoderConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("intermediate", C.c_int32), ("max_position", C.c_int32), ("type_vocab_size", C.c_int32),
                ("layer_norm_eps", C.c_float), ("compute_dtype", C.c_int32)]


_LIB = None
LIB_NAME = "libmrag_hip.so"


def lib_path() -> Path:
    env = os.environ.get("MRAG_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / LIB_NAME


_i, _i64, _vp, _fp, _dp = C.c_int, C.c_int64, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)
_h = C.c_uint64

# name -> argtypes (every symbol include/mrag.h declares; restype is int unless noted)
SIGNATURES = {
    "mrag_abi_version": [],
    "mrag_last_error": [],
    "mrag_device_count": [C.POINTER(_i)],
    "mrag_cosine_f64": [_i, _vp, _vp, _i64, _i, _vp, _i, _vp],
    "mrag_cosine_matrix_f64": [_i, _vp, _i64, _i, _vp, _i, _vp],
    "mrag_cosine_adjacent_f64": [_i, _vp, _i64, _i, C.c_double, _vp, _vp],
    "mrag_index_create": [_i, _i, _i, _i, C.POINTER(_h)],
    "mrag_index_destroy": [_h],
    "mrag_index_reserve": [_h, _i64],
    "mrag_index_add": [_h, _vp, _i64, _i, _i, _i, _vp],
    "mrag_index_size": [_h, C.POINTER(_i64)],
    "mrag_index_dim": [_h, C.POINTER(_i)],
    "mrag_index_set_id_base": [_h, _i64],
    "mrag_index_max_k": [_i64, C.POINTER(_i)],
    "mrag_index_last_wide_redone": [C.POINTER(_i64)],
    "mrag_index_get_rows": [_h, _i64, _i64, _vp, _i, _vp],
    "mrag_index_search": [_h, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mrag_index_score_rows": [_h, _vp, _i, _i, _vp, _i64, _vp, _vp],
    "mrag_index_last_timing": [_h, _fp, _fp],
    "mrag_index_measure_clock": [_h, _i],
    "mrag_index_last_clock": [_h, _fp],
    "mrag_topk_merge": [_vp, _vp, _i, _i64, _i, _vp, _vp, _i],
    "mrag_topk_merge_device": [_i, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp],
    "mrag_pack_partial_device": [_i, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "mrag_topk_merge_packed_device": [_i, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp],
    "mrag_ivf_create": [_i, _i, _i, _i, _i, C.POINTER(_h)],
    "mrag_ivf_destroy": [_h],
    "mrag_ivf_train": [_h, _vp, _i64, _i, _i, _i, _i, C.c_uint64, _vp],
    "mrag_ivf_set_centroids": [_h, _vp, _i, _i, _i, _vp],
    "mrag_ivf_get_centroids": [_h, _vp, _i, _vp],
    "mrag_ivf_add": [_h, _vp, _i64, _i, _i, _i, _vp],
    "mrag_ivf_size": [_h, C.POINTER(_i64)],
    "mrag_ivf_set_id_base": [_h, _i64],
    "mrag_ivf_get_assignments": [_h, _vp, _i, _vp],
    "mrag_ivf_search": [_h, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mrag_ivf_last_timing": [_h, _fp, _fp, C.POINTER(_i64), C.POINTER(_i)],
    "mrag_bm25_create": [_i, _i64, _i64, _vp, _vp, _vp, _vp, C.c_double, C.POINTER(_h)],
    "mrag_bm25_destroy": [_h],
    "mrag_bm25_search": [_h, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(_i), _vp],
    "mrag_fuse_topk": [_i, _vp, _vp, _i, _i, _i, C.c_double, C.c_double, C.c_double, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _vp],
    "mrag_encoder_create": [C.POINTER(EncoderConfig), _i, C.POINTER(_h)],
    "mrag_encoder_destroy": [_h],
    "mrag_encoder_set_param": [_h, C.c_char_p, _vp, _i64, _i, _vp],
    "mrag_encoder_missing_params": [_h, C.POINTER(_i)],
    "mrag_encoder_forward": [_h, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "mrag_encoder_last_timing": [_h, _fp],
}


def load():
    """dlopen the library once and declare every entry point."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not p.exists():
        raise MragLibraryMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C a-modular-rag-framework_amd/csrc`).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.  Two HIP runtimes
    # in one process cannot both open the GPU, so torch's copy must be the one already mapped when
    # libmrag_hip.so resolves its libamdhip64.so.7 dependency (same SONAME -> shared, one runtime).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(p))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI drift, surface it
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "mrag_last_error" else C.c_int
    _LIB = lib
    return lib


def check(status: int):
    if status != 0:
        msg = load().mrag_last_error()
        raise MragError(status, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    check(load().mrag_device_count(C.byref(n)))
    return n.value
