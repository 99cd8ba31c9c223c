"""ctypes binding of libmtv_hip.so (include/mtv_hip.h).  No torch types cross this boundary:
tensors go down as raw device pointers (`Tensor.data_ptr()`), sizes as ints.

The library is built in-tree by `moditalker_amd/csrc/build.sh` (or `__graft_entry__.build()`).
There is NO CPU fallback: if the library is missing or no HIP device is present the product path
raises `MtvError`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

MTV_OK = 0
MTV_IGNORED = 1
MTV_MAX_LEVELS = 8
# (MTV_LIB: load another build of the same library, e.g. the phase-timestamp diagnostic build libmtv_hip_stamp.so)
LIB_PATH = os.environ.get("MTV_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmtv_hip.so")


class MtvError(RuntimeError):
    pass


class MtvConfig(C.Structure):
    _fields_ = [
        ("model_channels", C.c_int32),
        ("num_res_blocks", C.c_int32),
        ("num_heads", C.c_int32),
        ("n_levels", C.c_int32),
        ("channel_mult", C.c_int32 * MTV_MAX_LEVELS),
        ("n_attention_resolutions", C.c_int32),
        ("attention_resolutions", C.c_int32 * MTV_MAX_LEVELS),
        ("use_scale_shift_norm", C.c_int32),
        ("out_channels", C.c_int32),
        ("res", C.c_int32),
        ("frames", C.c_int32),
        ("max_batch", C.c_int32),
    ]


class MtvAeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("channels", "resolution", "frames", "patch", "embed_dim", "depth", "heads",
                                         "dim_head", "max_batch")]


class MtvXattnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("query_dim", "context_dim", "heads", "dim_head", "max_batch", "max_queries", "max_keys")]


class MtvDdimStep(C.Structure):
    _fields_ = [
        ("t", C.c_int32),
        ("last", C.c_int32),
        ("sqrt_recip_ac", C.c_float),
        ("sqrt_recipm1_ac", C.c_float),
        ("sqrt_ac_next", C.c_float),
        ("c", C.c_float),
        ("sigma", C.c_float),
        ("noise_index", C.c_int32),
    ]


class MtvWork(C.Structure):
    _fields_ = [
        ("flops_conv3x3", C.c_double),
        ("flops_1x1", C.c_double),
        ("flops_attn_core", C.c_double),
        ("flops_linear", C.c_double),
        ("bytes_weights_conv", C.c_double),
        ("bytes_weights_other", C.c_double),
        ("bytes_act_conv_path", C.c_double),
        ("n_launches", C.c_int32),
        ("n_launches_step", C.c_int32),
    ]


class MtvOpTime(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/mtv_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("mtv_last_error", C.c_char_p, []),
    ("mtv_version", C.c_int, []),
    ("mtv_create", C.c_int, [C.POINTER(MtvConfig), C.POINTER(_P)]),
    ("mtv_destroy", C.c_int, [_P]),
    ("mtv_num_weights", C.c_int, [_P]),
    ("mtv_weight_info", C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    ("mtv_load_weight", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64)]),
    ("mtv_weights_missing", C.c_int, [_P]),
    ("mtv_forward", C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    ("mtv_ddim_sample", C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.POINTER(MtvDdimStep), C.c_int, C.c_int, _P]),
    ("mtv_debug_tap", C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("mtv_get_work", C.c_int, [_P, C.POINTER(MtvWork)]),
    ("mtv_set_eager", C.c_int, [_P, C.c_int]),
    ("mtv_profile_forward", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(MtvOpTime), C.c_int, C.POINTER(C.c_int), _P]),
    ("mtv_profile_step", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(MtvOpTime), C.c_int, C.POINTER(C.c_int), _P]),
    ("mtv_debug_stamps", C.c_int, [_P, C.c_int, C.c_char_p, _P]),
    ("mtv_debug_force_lds", C.c_int, [C.c_int, C.c_int]),
    ("mtv_debug_force_lin", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_debug_force_win", C.c_int, [C.c_int, C.c_int]),
    ("mtv_debug_force_win_ks", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_debug_force_pw", C.c_int, [C.c_int, C.c_int]),
    ("mtv_debug_force_pw_waves", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_debug_force_b3", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_debug_attention_qb", C.c_int, [C.c_int]),
    ("mtv_debug_deep", C.c_int, [C.c_int]),
    ("mtv_debug_deep_options", C.c_int, [C.c_int]),
    ("mtv_ae_create", C.c_int, [C.POINTER(MtvAeConfig), C.POINTER(_P)]),
    ("mtv_ae_destroy", C.c_int, [_P]),
    ("mtv_ae_set_rotary", C.c_int, [_P, _P, _P]),
    ("mtv_ae_decode", C.c_int, [_P, _P, _P, C.c_int, _P]),
    ("mtv_ae_extract", C.c_int, [_P, _P, _P, C.c_int, _P]),
    ("mtv_ae_profile", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(MtvOpTime), C.c_int, C.POINTER(C.c_int), _P]),
    ("mtv_xattn_create", C.c_int, [C.POINTER(MtvXattnConfig), C.POINTER(_P)]),
    ("mtv_xattn_destroy", C.c_int, [_P]),
    ("mtv_xattn_forward", C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("mtv_selftest_geometry", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_selftest_deep", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_selftest_block", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("mtv_selftest_win", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("mtv_debug_gather_index", C.c_int, [C.c_int] * 6),
    ("mtv_check_fault", C.c_int, [_P]),
    ("mtv_resident_cus", C.c_int, [_P]),
    ("mtv_debug_resident_cus", C.c_int, [C.c_int]),
    ("mtv_debug_arm_fault", C.c_int, [_P]),
]

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libmtv_hip.so and type every entry point.  Raises MtvError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtvError(f"{LIB_PATH} not found: build it with moditalker_amd/csrc/build.sh "
                       f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise MtvError(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = load().mtv_last_error()
        raise MtvError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
    return rc
