"""ctypes loader for libkamihip.so — the C ABI declared in include/kami_hip.h."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkamihip.so")

KH_OK, KH_ERR_INVALID, KH_ERR_HIP, KH_ERR_NO_WEIGHTS = 0, 1, 2, 3
KH_ERR_NAN_POLICY, KH_ERR_NAN_VALUE, KH_ERR_NO_DEVICE = 4, 5, 6
KH_F32, KH_BF16, KH_F16 = 0, 1, 2
KH_VALUE_REFERENCE_FLAT, KH_VALUE_PER_SAMPLE0 = 0, 1
KH_MAX_OUTSTANDING = 64
DTYPES = {"f32": KH_F32, "fp32": KH_F32, "bf16": KH_BF16, "f16": KH_F16, "fp16": KH_F16}


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("features", C.c_int32),
                ("psize", C.c_int32), ("filters", C.c_int32), ("residuals", C.c_int32),
                ("dtype", C.c_int32), ("value_mode", C.c_int32), ("device", C.c_int32),
                ("reserved", C.c_int32 * 7)]


# numpy view of kh_board (80 bytes)
BOARD_DTYPE = np.dtype([("piece_occ", "<u8", (6,)), ("color_occ", "<u8", (2,)),
                        ("ply", "<i4"), ("halfmove_clock", "<i4"),
                        ("ctm", "u1"), ("castle_rights", "u1"), ("pad", "u1", (6,))])
assert BOARD_DTYPE.itemsize == 80

# every symbol include/kami_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class TrainConfig(C.Structure):
    _fields_ = [("lr", C.c_float), ("epochs", C.c_int32), ("batch", C.c_int32), ("detect_anomaly", C.c_int32), ("reserved", C.c_int32 * 4)]


SYMBOLS = {
    "kh_weight_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "kh_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "kh_destroy": (None, [_P]),
    "kh_load_weights": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "kh_train": (C.c_int, [_P, _P, _P, _P, C.c_int, C.POINTER(TrainConfig), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "kh_train_order": (C.c_int, [C.c_int, C.c_int, _P]),
    "kh_checkpoint_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kh_load_checkpoint": (C.c_int, [_P, C.c_char_p]),
    "kh_get_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "kh_generation": (C.c_int, [_P]),
    "kh_clone": (C.c_int, [_P, C.POINTER(_P)]),
    "kh_infer": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "kh_pin_buffer": (C.c_int, [_P, _P, C.c_size_t]),
    "kh_unpin_buffer": (C.c_int, [_P, _P]),
    "kh_infer_full": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "kh_encode": (C.c_int, [_P, _P, C.c_int, _P]),
    "kh_encode_infer": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "kh_infer_legal": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "kh_encode_infer_legal": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "kh_submit_infer": (C.c_int, [_P, _P, C.c_int, _P, _P, C.POINTER(C.c_int64)]),
    "kh_submit_encode_infer_legal": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, C.POINTER(C.c_int64)]),
    "kh_wait": (C.c_int, [_P, C.c_int64]),
    "kh_try_wait": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int)]),
    "kh_set_coalesce": (C.c_int, [_P, C.c_int, C.c_int]),
    "kh_set_coalesce_callers": (C.c_int, [_P, C.c_int]),
    "kh_coalesce_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kh_infer_device": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "kh_encode_device": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "kh_encode_infer_device": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "kh_time_infer_device": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_float)]),
    "kh_time_encode_device": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.POINTER(C.c_float)]),
    "kh_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "kh_dev_free": (C.c_int, [_P, _P]),
    "kh_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kh_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kh_sync": (C.c_int, [_P]),
    "kh_device_count": (C.c_int, []),
    "kh_last_error": (C.c_char_p, []),
    "kh_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libkamihip.so; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is not built "
                "(run `python -m kami_amd.build`); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (load().kh_last_error() or b"").decode("utf-8", "replace")      # (a message may quote bytes of a malformed file)
