"""ctypes binding of oracle/libkami_oracle.so — TEST INFRASTRUCTURE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (kami_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PSIZE = 4672
VALUE_WIDTH = 256
OBSIZE = 64 * 30


class Board(C.Structure):
    _fields_ = [("piece_occ", C.c_uint64 * 6), ("color_occ", C.c_uint64 * 2),
                ("ply", C.c_int32), ("halfmove_clock", C.c_int32),
                ("ctm", C.c_uint8), ("castle_rights", C.c_uint8), ("pad", C.c_uint8 * 6)]


BOARD_DTYPE = np.dtype([("piece_occ", "<u8", (6,)), ("color_occ", "<u8", (2,)),
                        ("ply", "<i4"), ("halfmove_clock", "<i4"),
                        ("ctm", "u1"), ("castle_rights", "u1"), ("pad", "u1", (6,))])
assert BOARD_DTYPE.itemsize == 80 and C.sizeof(Board) == 80


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libkami_oracle.so")
    src = os.path.join(_HERE, "kami_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libkami_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.ko_weight_count.restype = C.c_size_t
        L.ko_weight_count.argtypes = [C.c_int] * 3
        L.ko_observe_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ko_board_from_fen.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.ko_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ko_infer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def weight_count(F, Cc, R):
    return lib().ko_weight_count(F, Cc, R)


def observe(boards: np.ndarray) -> np.ndarray:
    boards = np.ascontiguousarray(boards, dtype=BOARD_DTYPE)
    out = np.empty((boards.shape[0], 8, 8, 30), dtype=np.float32)
    lib().ko_observe_batch(_p(boards), boards.shape[0], _p(out))
    return out


def boards_from_fens(fens, plies) -> np.ndarray:
    out = np.zeros(len(fens), dtype=BOARD_DTYPE)
    for i, (f, p) in enumerate(zip(fens, plies)):
        b = Board()
        rc = lib().ko_board_from_fen(f.encode() if isinstance(f, str) else f, int(p), C.byref(b))
        if rc:
            raise ValueError(f"bad FEN ({rc}): {f}")
        out[i] = np.frombuffer(bytes(b), dtype=BOARD_DTYPE)[0]
    return out


def forward(blob, F, Cc, R, x, nthreads=0, want_logits=True):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    B = x.shape[0]
    assert x.size == B * 64 * F and blob.size == weight_count(F, Cc, R)
    policy = np.empty((B, PSIZE), np.float32)
    vfull = np.empty((B, VALUE_WIDTH), np.float32)
    logits = np.empty((B, PSIZE), np.float32) if want_logits else None
    rc = lib().ko_forward(_p(blob), F, Cc, R, _p(x), B, _p(policy), _p(vfull),
                          _p(logits) if want_logits else None, nthreads)
    if rc:
        raise RuntimeError(f"ko_forward failed ({rc})")
    return policy, vfull, logits


def infer(blob, F, Cc, R, x, nthreads=0):
    """-> (status, policy, value) with the reference's Q10 value copy-out."""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    B = x.shape[0]
    policy = np.empty((B, PSIZE), np.float32)
    value = np.empty((B,), np.float32)
    rc = lib().ko_infer(_p(blob), F, Cc, R, _p(x), B, _p(policy), _p(value), nthreads)
    return rc, policy, value
