"""Host-side mirror of the reference's evaluator class, over the C ABI.

`NN` keeps the method set and argument meaning of `kami::NN` (kami/nn/nn.h:40-73):
`NN(width, height, features, psize)`, `infer`, `read`, `write`, `get_generation`, `isCUDA`,
`obsize`, `polsize`, `clone`; the two option keys the reference's module reads
(`filters`, `residuals`, nn.cpp:42-43) are explicit constructor arguments here.
`infer` raises RuntimeError with the reference's messages on NaN outputs (nn.cpp:176-180).
The C++ twin used to drop into kami.cpp is kami_amd/host/nn.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L
from . import weights as W

PSIZE = 4672
NFEATURES = 30
OBSIZE = 8 * 8 * NFEATURES
VALUE_WIDTH = 256


class KamiError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(msg)
        self.status = status


def _chk(rc: int) -> None:
    if rc != L.KH_OK:
        raise KamiError(rc, L.last_error())


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def read_checkpoint(path: str):
    """kh_checkpoint_read: -> (blob, features, filters, residuals, generation) of a checkpoint file — a libtorch
    archive written by the reference's NN::write (nn.cpp:189-202) or an engine KAMW blob.  Needs no GPU."""
    lib = L.load()
    F, Cc, R, gen, n = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    p = path.encode()
    _chk(lib.kh_checkpoint_read(p, C.byref(F), C.byref(Cc), C.byref(R), C.byref(gen), None, 0, C.byref(n)))
    blob = np.empty(n.value, np.float32)
    _chk(lib.kh_checkpoint_read(p, None, None, None, None, _ptr(blob), blob.size, None))
    return blob, F.value, Cc.value, R.value, gen.value


class Ticket:
    """An outstanding kh_submit_*: keeps the caller-side buffers alive until wait() (the engine reads and writes them)."""

    def __init__(self, nn, ticket, outputs, keep):
        self._nn, self._t, self._out, self._keep = nn, ticket, outputs, keep

    def wait(self):
        _chk(self._nn._lib.kh_wait(self._nn._h, self._t))
        return self._out

    def try_wait(self):
        """kh_try_wait: the outputs if the submission has finished (the ticket is consumed, as by wait()), else None."""
        done = C.c_int(0)
        _chk(self._nn._lib.kh_try_wait(self._nn._h, self._t, C.byref(done)))
        return self._out if done.value else None


# engines still open when the interpreter exits are closed while the HIP runtime is still up (module teardown order
# is arbitrary otherwise, and an engine owns streams, the queue's dispatcher thread and device memory)
import atexit
import weakref

_LIVE = weakref.WeakSet()


def _close_pools(nn=None):
    import sys
    search = sys.modules.get(__package__ + ".search")
    if search is not None:
        search.close_pools_of(nn)


@atexit.register
def _close_all():
    _close_pools()                      # pools first: they hold engine handles
    for nn in list(_LIVE):
        try:
            nn.close()
        except Exception:
            pass


class NN:
    def __init__(self, width: int = 8, height: int = 8, features: int = NFEATURES,
                 psize: int = PSIZE, *, filters: int = 256, residuals: int = 2,
                 dtype: str = "f32", value_mode: int = L.KH_VALUE_REFERENCE_FLAT,
                 device: int = 0):
        self._lib = L.load()
        self.cfg = L.Config(width=width, height=height, features=features, psize=psize,
                            filters=filters, residuals=residuals, dtype=L.DTYPES[dtype],
                            value_mode=value_mode, device=device)
        self.dtype = dtype
        self._h = C.c_void_p()
        _chk(self._lib.kh_create(C.byref(self.cfg), C.byref(self._h)))
        _LIVE.add(self)

    # -- kami::NN surface ---------------------------------------------------------------
    def get_generation(self) -> int:
        return self._lib.kh_generation(self._h)

    def train(self, inputs: np.ndarray, obs_p: np.ndarray, obs_v: np.ndarray, *, mlr: int = 5, epochs: int = 8,
              batchsize: int = 8, detect_anomaly: bool = False):
        """NN::train (nn.cpp:224-377) on the device; option names and defaults of nn.cpp:236-238.
        Returns (average loss of the first epoch, of the last epoch); the generation goes up by one.
        detect_anomaly (nn.cpp:231-232,329-344): every batch's input and the forward's two outputs are checked for NaN,
        and the call fails with the reference's messages."""
        x = np.ascontiguousarray(inputs, dtype=np.float32)
        p = np.ascontiguousarray(obs_p, dtype=np.float32)
        v = np.ascontiguousarray(obs_v, dtype=np.float32)
        n = x.shape[0]
        assert p.shape == (n, PSIZE) and v.shape == (n,)
        cfg = L.TrainConfig(mlr / 1000.0, epochs, batchsize, 1 if detect_anomaly else 0)
        first, last = C.c_float(), C.c_float()
        _chk(self._lib.kh_train(self._h, x.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                                v.ctypes.data_as(C.c_void_p), n, C.byref(cfg), C.byref(first), C.byref(last)))
        self._blob = self.get_weights()
        return first.value, last.value

    def get_weights(self) -> np.ndarray:
        """The engine's current fp32 parameters in blob order (kh_get_weights)."""
        n = self._lib.kh_weight_count(self.cfg.features, self.cfg.filters, self.cfg.residuals)
        out = np.empty(n, np.float32)
        _chk(self._lib.kh_get_weights(self._h, _ptr(out), n))
        return out

    def isCUDA(self) -> bool:            # nn.h:62 — true: the engine only exists on the GPU
        return True

    def obsize(self) -> int:
        return self.cfg.width * self.cfg.height * self.cfg.features

    def polsize(self) -> int:
        return self.cfg.psize

    def infer(self, input: np.ndarray, batch: Optional[int] = None,
              policy: Optional[np.ndarray] = None, value: Optional[np.ndarray] = None):
        """nn.cpp:155-187.  input [batch,8,8,F] fp32 -> (policy [batch,4672], value [batch])."""
        x = np.ascontiguousarray(input, dtype=np.float32)
        if batch is None:
            batch = x.size // self.obsize()
        if x.size != batch * self.obsize():
            raise ValueError("input size does not match batch * obsize()")
        if policy is None:
            policy = np.empty((batch, PSIZE), np.float32)
        if value is None:
            value = np.empty((batch,), np.float32)
        _chk(self._lib.kh_infer(self._h, _ptr(x), batch, _ptr(policy), _ptr(value)))
        return policy, value

    def read(self, path: str) -> None:
        """NN::read (nn.cpp:204-222): the reference's own libtorch archives and the engine's KAMW blobs."""
        blob, F, Cc, R, gen = read_checkpoint(path)
        if (F, Cc, R) != (self.cfg.features, self.cfg.filters, self.cfg.residuals):
            raise KamiError(L.KH_ERR_INVALID, "checkpoint shape does not match this NN")
        self.load_weights(blob, gen)

    def write(self, path: str) -> None:
        """NN::write (nn.cpp:189-202) into the engine's KAMW container (read() takes it back, and the reference's own
        archives as well)."""
        W.save(path, self.get_weights(), self.cfg.features, self.cfg.filters, self.cfg.residuals, self.get_generation())

    def clone(self) -> "NN":
        other = object.__new__(NN)
        other._lib, other.cfg, other.dtype = self._lib, self.cfg, self.dtype
        other._blob = self._blob
        other._h = C.c_void_p()
        _chk(self._lib.kh_clone(self._h, C.byref(other._h)))
        _LIVE.add(other)
        return other

    # -- engine extras ------------------------------------------------------------------
    _blob: Optional[np.ndarray] = None

    def load_weights(self, blob: np.ndarray, generation: int = 0) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _chk(self._lib.kh_load_weights(self._h, _ptr(blob), blob.size, generation))
        self._blob = blob

    def infer_full(self, input: np.ndarray, want_logits: bool = True):
        """-> (policy [B,4672], value_full [B,256], logits [B,4672] or None)"""
        x = np.ascontiguousarray(input, dtype=np.float32)
        batch = x.size // self.obsize()
        policy = np.empty((batch, PSIZE), np.float32)
        vfull = np.empty((batch, VALUE_WIDTH), np.float32)
        logits = np.empty((batch, PSIZE), np.float32) if want_logits else None
        _chk(self._lib.kh_infer_full(self._h, _ptr(x), batch, _ptr(policy), _ptr(vfull),
                                     _ptr(logits) if want_logits else None))
        return policy, vfull, logits

    def encode(self, boards: np.ndarray) -> np.ndarray:
        """Env::observe (env.h:202-262) for kh_board records -> [n,8,8,30] fp32."""
        b = np.ascontiguousarray(boards, dtype=L.BOARD_DTYPE)
        out = np.empty((b.shape[0], 8, 8, NFEATURES), np.float32)
        _chk(self._lib.kh_encode(self._h, _ptr(b), b.shape[0], _ptr(out)))
        return out

    def encode_infer(self, boards: np.ndarray):
        b = np.ascontiguousarray(boards, dtype=L.BOARD_DTYPE)
        n = b.shape[0]
        policy = np.empty((n, PSIZE), np.float32)
        value = np.empty((n,), np.float32)
        _chk(self._lib.kh_encode_infer(self._h, _ptr(b), n, _ptr(policy), _ptr(value)))
        return policy, value

    def infer_legal(self, input_or_boards: np.ndarray, action_offsets: np.ndarray, actions: np.ndarray):
        """Legal-move policy gather (MCTS::expand, mcts.h:273-276): -> (priors [sum of counts], value [B]).
        `input_or_boards` is either fp32 planes [B,8,8,F] or kh_board records."""
        offs = np.ascontiguousarray(action_offsets, dtype=np.int32)
        acts = np.ascontiguousarray(actions, dtype=np.int32)
        n = offs.size - 1
        priors = np.empty((int(offs[-1]),), np.float32)
        value = np.empty((n,), np.float32)
        if input_or_boards.dtype == L.BOARD_DTYPE:
            b = np.ascontiguousarray(input_or_boards)
            _chk(self._lib.kh_encode_infer_legal(self._h, _ptr(b), n, _ptr(offs), _ptr(acts), _ptr(priors), _ptr(value)))
        else:
            x = np.ascontiguousarray(input_or_boards, dtype=np.float32)
            _chk(self._lib.kh_infer_legal(self._h, _ptr(x), n, _ptr(offs), _ptr(acts), _ptr(priors), _ptr(value)))
        return priors, value

    # -- submit / wait (the engine's coalescing queue) -------------------------------------
    def submit_infer(self, input: np.ndarray) -> "Ticket":
        """kh_submit_infer: queue a small batch of planes; Ticket.wait() -> (policy, value).  Submissions queued while a
        launch is in flight are evaluated as one launch."""
        x = np.ascontiguousarray(input, dtype=np.float32)
        batch = x.size // self.obsize()
        policy = np.empty((batch, PSIZE), np.float32)
        value = np.empty((batch,), np.float32)
        t = C.c_int64()
        _chk(self._lib.kh_submit_infer(self._h, _ptr(x), batch, _ptr(policy), _ptr(value), C.byref(t)))
        return Ticket(self, t.value, (policy, value), (x,))

    def submit_infer_legal(self, boards: np.ndarray, action_offsets: np.ndarray, actions: np.ndarray) -> "Ticket":
        """kh_submit_encode_infer_legal; Ticket.wait() -> (priors, value)."""
        b = np.ascontiguousarray(boards, dtype=L.BOARD_DTYPE)
        offs = np.ascontiguousarray(action_offsets, dtype=np.int32)
        acts = np.ascontiguousarray(actions, dtype=np.int32)
        n = offs.size - 1
        priors = np.empty((max(1, int(offs[-1])),), np.float32)
        value = np.empty((n,), np.float32)
        t = C.c_int64()
        _chk(self._lib.kh_submit_encode_infer_legal(self._h, _ptr(b), n, _ptr(offs), _ptr(acts), _ptr(priors), _ptr(value), C.byref(t)))
        return Ticket(self, t.value, (priors[:int(offs[-1])], value), (b, offs, acts, priors))

    def pin(self, array: np.ndarray) -> None:
        """kh_pin_buffer: register a long-lived numpy buffer used as infer()'s input / policy (unpin before freeing it)."""
        _chk(self._lib.kh_pin_buffer(self._h, _ptr(array), array.nbytes))

    def unpin(self, array: np.ndarray) -> None:
        _chk(self._lib.kh_unpin_buffer(self._h, _ptr(array)))

    def set_coalesce(self, target_batch: int = 0, max_wait_us: int = 0) -> None:
        _chk(self._lib.kh_set_coalesce(self._h, target_batch, max_wait_us))


    def set_coalesce_callers(self, callers: int):
        """With a target set: a batch also goes once it holds `callers` submissions (0: off)."""
        _chk(self._lib.kh_set_coalesce_callers(self._h, callers))

    def coalesce_stats(self):
        """(launches made by the queue, positions they held)"""
        a, b = C.c_int64(), C.c_int64()
        _chk(self._lib.kh_coalesce_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def handle(self):
        return self._h

    def close(self) -> None:
        if getattr(self, "_h", None):
            _close_pools(self)          # a self-play pool on this engine is closed with it
            self._lib.kh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
