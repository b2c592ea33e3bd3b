"""ctypes binding of libkamisearch.so (include/kami_search.h): rules / MCTS mirrors and the self-play pool."""
from __future__ import annotations

import ctypes as C
import os

from . import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkamisearch.so")
MAX_RECORD_ACTIONS = 96


class Board(C.Structure):
    """kh_board (include/kami_hip.h); same bytes as kami_amd._lib.BOARD_DTYPE."""
    _fields_ = [("piece_occ", C.c_uint64 * 6), ("color_occ", C.c_uint64 * 2), ("ply", C.c_int32), ("halfmove_clock", C.c_int32),
                ("ctm", C.c_uint8), ("castle_rights", C.c_uint8), ("pad", C.c_uint8 * 6)]


assert C.sizeof(Board) == 80


class PoolConfig(C.Structure):
    _fields_ = [("games", C.c_int32), ("threads", C.c_int32), ("nodes", C.c_int32), ("leaves_per_tree", C.c_int32),
                ("cpuct", C.c_float), ("noise_weight", C.c_float),
                ("alpha_initial", C.c_float), ("alpha_decay", C.c_float), ("alpha_final", C.c_float),
                ("alpha_cutoff", C.c_int32), ("draw_value_pct", C.c_int32), ("seed", C.c_uint32),
                ("pipeline", C.c_int32), ("coalesce_target", C.c_int32), ("coalesce_wait_us", C.c_int32), ("reserved", C.c_int32 * 1)]


class PoolStats(C.Structure):
    _fields_ = [("evals", C.c_int64), ("batches", C.c_int64), ("moves", C.c_int64), ("games_finished", C.c_int64),
                ("white_wins", C.c_int64), ("black_wins", C.c_int64), ("draws", C.c_int64), ("records", C.c_int64),
                ("seconds", C.c_double), ("evals_per_s", C.c_double), ("mean_batch", C.c_double),
                ("engine_seconds", C.c_double)]


class Record(C.Structure):
    _fields_ = [("board", Board), ("value", C.c_float), ("nact", C.c_int32),
                ("actions", C.c_int16 * MAX_RECORD_ACTIONS), ("visits", C.c_float * MAX_RECORD_ACTIONS)]


SYMBOLS = {
    "ks_perft": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]),
    "ks_fen_actions": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.c_int]),
    "ks_env_new": (C.c_void_p, []),
    "ks_env_free": (None, [C.c_void_p]),
    "ks_env_ply": (C.c_int, [C.c_void_p]),
    "ks_env_actions": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "ks_env_push": (C.c_int, [C.c_void_p, C.c_int]),
    "ks_env_pop": (C.c_int, [C.c_void_p]),
    "ks_env_terminal": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ks_env_turn": (C.c_float, [C.c_void_p]),
    "ks_env_fen": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ks_env_record": (None, [C.c_void_p, C.POINTER(Board)]),
    "ks_mcts_synthetic": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_char_p, C.c_int]),
    "ks_pool_create": (C.c_int, [C.c_void_p, C.POINTER(PoolConfig), C.POINTER(C.c_void_p)]),
    "ks_pool_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(PoolConfig), C.POINTER(C.c_void_p)]),
    "ks_pool_publish_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ks_pool_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.POINTER(PoolStats)]),
    "ks_pool_drain_records": (C.c_int64, [C.c_void_p, C.POINTER(Record), C.c_int64]),
    "ks_pool_destroy": (None, [C.c_void_p]),
    "ks_last_error": (C.c_char_p, []),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing (run `python -m kami_amd.build`)")
        L.load()                                   # libkamihip.so first: libkamisearch.so links it
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Env:
    """kami::Env (kami/env.h:41-485) through the C ABI."""

    def __init__(self):
        self.lib = load()
        self.h = C.c_void_p(self.lib.ks_env_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ks_env_free(self.h)
            self.h = None

    def ply(self):
        return self.lib.ks_env_ply(self.h)

    def actions(self):
        buf = (C.c_int32 * 256)()
        n = self.lib.ks_env_actions(self.h, buf, 256)
        return list(buf[:n])

    def push(self, action):
        if self.lib.ks_env_push(self.h, int(action)):
            raise ValueError(self.lib.ks_last_error().decode())

    def pop(self):
        if self.lib.ks_env_pop(self.h):
            raise ValueError(self.lib.ks_last_error().decode())

    def terminal(self):
        v = C.c_float()
        t = self.lib.ks_env_terminal(self.h, C.byref(v))
        return bool(t), v.value

    def turn(self):
        return self.lib.ks_env_turn(self.h)

    def print(self):
        buf = C.create_string_buffer(128)
        self.lib.ks_env_fen(self.h, buf, 128)
        return buf.value.decode()

    def record(self):
        """the compact kh_board record as a numpy scalar array of kami_amd._lib.BOARD_DTYPE"""
        import numpy as np
        b = Board()
        self.lib.ks_env_record(self.h, C.byref(b))
        return np.frombuffer(bytes(b), dtype=L.BOARD_DTYPE).copy()


def perft(fen: str, depth: int) -> int:
    n = C.c_uint64()
    if load().ks_perft(fen.encode(), depth, C.byref(n)):
        raise ValueError(load().ks_last_error().decode())
    return n.value


def fen_actions(fen: str):
    buf = (C.c_int32 * 256)()
    n = load().ks_fen_actions(fen.encode(), buf, 256)
    if n < 0:
        raise ValueError(load().ks_last_error().decode())
    return list(buf[:n])


def mcts_synthetic(nodes: int, nmoves: int, leaves: int = 1, picks=None) -> str:
    lib = load()
    picks = list(picks or [])
    arr = (C.c_int32 * max(1, len(picks)))(*picks)
    buf = C.create_string_buffer(1 << 20)
    if lib.ks_mcts_synthetic(nodes, nmoves, leaves, arr, len(picks), buf, len(buf)):
        raise RuntimeError(lib.ks_last_error().decode())
    return buf.value.decode()


import weakref

_POOLS = weakref.WeakSet()      # live pools: closed before their engines (kami_amd.nn._close_all, NN.close)


def close_pools_of(nn=None):
    """Close every live pool (of engine `nn`, or of any engine): a pool must not outlive the engine it feeds."""
    for pool in list(_POOLS):
        if nn is None or any(e is nn for e in pool.engines):
            pool.close()


class Pool:
    """Self-play pool (kami/selfplay.cpp:58-213) feeding an engine (kami_amd.NN) with compact records."""

    def __init__(self, nn, games=512, threads=4, nodes=64, leaves_per_tree=1, cpuct=1.0, noise_weight=0.05,
                 alpha=(1.0, 1.0, 1.0), alpha_cutoff=1, draw_value_pct=50, seed=1, pipeline=False, coalesce_target=0,
                 coalesce_wait_us=0):
        self.lib = load()
        self.engines = list(nn) if isinstance(nn, (list, tuple)) else [nn]      # one evaluator per GPU; kept alive
        self.nn = self.engines[0]
        cfg = PoolConfig(games, threads, nodes, leaves_per_tree, cpuct, noise_weight, alpha[0], alpha[1], alpha[2],
                         alpha_cutoff, draw_value_pct, seed, int(pipeline), coalesce_target, coalesce_wait_us)   # pipeline: False / True (two sets) / 2..4 sets
        self.h = C.c_void_p()
        handles = (C.c_void_p * len(self.engines))(*[e.handle for e in self.engines])
        if self.lib.ks_pool_create_multi(handles, len(self.engines), C.byref(cfg), C.byref(self.h)):
            self.h = None
            raise RuntimeError(self.lib.ks_last_error().decode())
        _POOLS.add(self)

    def run(self, min_evals=0, max_seconds=1.0):
        if not getattr(self, "h", None):
            raise RuntimeError("the pool is closed (its engine was closed)")
        st = PoolStats()
        if self.lib.ks_pool_run(self.h, int(min_evals), float(max_seconds), C.byref(st)):
            raise RuntimeError(self.lib.ks_last_error().decode())
        return st

    def publish_weights(self, blob, generation: int):
        """A new generation on every engine of the pool (selfplay.cpp:282-283)."""
        import numpy as np
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        if self.lib.ks_pool_publish_weights(self.h, blob.ctypes.data_as(C.c_void_p), blob.size, int(generation)):
            raise RuntimeError(self.lib.ks_last_error().decode())
        for e in self.engines:
            e._blob = blob

    def drain(self, cap=1 << 16):
        buf = (Record * cap)()
        n = self.lib.ks_pool_drain_records(self.h, buf, cap)
        return buf[:n]

    def close(self):
        if getattr(self, "h", None):
            self.lib.ks_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
