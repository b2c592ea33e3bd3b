"""The device-pointer entry points — what bench.py times (kh_infer_device, kh_encode_infer_device; kh_api.hip) — against
the host-buffer calls of the same engine: same bits, on the engine's own stream, on a caller's stream, and on two
caller streams in alternation (the per-layer paths keep scratch in the engine's slot: a call on another stream than the
previous one must be ordered behind it).  nn.cpp:155-187 is the contract both implement.  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest

from kami_amd import NN, _lib as L, weights as W
from oracle import pyoracle as ko

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class DevBuf:
    def __init__(self, nn, nbytes):
        self.nn, self.p, self.nbytes = nn, C.c_void_p(), nbytes
        assert nn._lib.kh_dev_alloc(nn.handle, nbytes, C.byref(self.p)) == 0, L.last_error()

    def put(self, a):
        a = np.ascontiguousarray(a)
        assert self.nn._lib.kh_memcpy_h2d(self.nn.handle, self.p, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
        return self

    def get(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert self.nn._lib.kh_memcpy_d2h(self.nn.handle, out.ctypes.data_as(C.c_void_p), self.p, out.nbytes) == 0
        return out

    def free(self):
        self.nn._lib.kh_dev_free(self.nn.handle, self.p)


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


CASES = [  # dtype, F, filters, residuals, batch: the headline shape, the exact-f32 per-layer path, a wide net, the VALU anchor's shape
    ("bf16", 119, 64, 6, 512), ("f16", 119, 64, 6, 77), ("f32", 30, 16, 2, 33), ("bf16", 119, 128, 2, 130), ("bf16", 30, 64, 6, 257),
]


@pytest.mark.parametrize("dtype,F,Cc,R,B", CASES, ids=[f"{c[0]}-F{c[1]}-{c[2]}x{c[3]}-B{c[4]}" for c in CASES])
def test_infer_device_equals_infer_full_on_every_stream(dtype, F, Cc, R, B):
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dtype)
    nn.load_weights(W.random_weights(F, Cc, R, seed=11, peaky=3.0), 1)
    rng = np.random.default_rng(5)
    xs = [rng.random((B, 8, 8, F), dtype=np.float32) for _ in range(2)]
    want = [nn.infer_full(x, want_logits=False) for x in xs]
    d_in = [DevBuf(nn, x.nbytes).put(x) for x in xs]
    d_p = [DevBuf(nn, B * 4672 * 4) for _ in range(2)]
    d_v = [DevBuf(nn, B * 256 * 4) for _ in range(2)]
    streams = [None, torch.cuda.Stream(), torch.cuda.Stream()]

    def call(i, st):
        sp = C.c_void_p(st.cuda_stream) if st is not None else None
        assert nn._lib.kh_infer_device(nn.handle, d_in[i].p, B, d_p[i].p, d_v[i].p, sp) == 0, L.last_error()

    def check(i, what):
        torch.cuda.synchronize(); assert nn._lib.kh_sync(nn.handle) == 0
        assert same_bits(d_p[i].get((B, 4672)), want[i][0]), f"policy differs: {what}"
        assert same_bits(d_v[i].get((B, 256)), want[i][1]), f"value tensor differs: {what}"

    call(0, streams[0]); check(0, "engine's own stream")
    call(1, streams[1]); check(1, "caller's stream")
    # two caller streams in alternation, no host synchronisation in between: input 0 on one, input 1 on the other
    for rep in range(6):
        call(rep & 1, streams[1 + (rep & 1)])
    check(0, "alternating streams"); check(1, "alternating streams")
    # and the same OUTPUT buffers reused across streams: the last call wins, whole rows
    for rep in range(4):
        sp = C.c_void_p(streams[1 + (rep & 1)].cuda_stream)
        assert nn._lib.kh_infer_device(nn.handle, d_in[rep & 1].p, B, d_p[0].p, d_v[0].p, sp) == 0
        torch.cuda.synchronize(); assert nn._lib.kh_sync(nn.handle) == 0
        assert same_bits(d_p[0].get((B, 4672)), want[rep & 1][0]) and same_bits(d_v[0].get((B, 256)), want[rep & 1][1])
    for b in d_in + d_p + d_v:
        b.free()
    nn.close()


@pytest.mark.parametrize("dtype,Cc", [("bf16", 64), ("f16", 64), ("f32", 24), ("bf16", 128)])
def test_encode_infer_device_equals_encode_infer(observe_fixture, dtype, Cc):
    f = observe_fixture
    boards = ko.boards_from_fens([s.decode() for s in f["fen"][:515]], f["ply"][:515])
    B = len(boards)
    nn = NN(8, 8, 30, 4672, filters=Cc, residuals=2, dtype=dtype)
    nn.load_weights(W.random_weights(30, Cc, 2, seed=4, peaky=3.0), 1)
    planes = nn.encode(boards)
    want = nn.infer_full(planes, want_logits=False)
    d_b = DevBuf(nn, boards.nbytes).put(boards)
    d_p, d_v = DevBuf(nn, B * 4672 * 4), DevBuf(nn, B * 256 * 4)
    for st in (None, torch.cuda.Stream(), torch.cuda.Stream(), None):
        sp = C.c_void_p(st.cuda_stream) if st is not None else None
        assert nn._lib.kh_encode_infer_device(nn.handle, d_b.p, B, d_p.p, d_v.p, sp) == 0, L.last_error()
        torch.cuda.synchronize(); assert nn._lib.kh_sync(nn.handle) == 0
        assert same_bits(d_p.get((B, 4672)), want[0]) and same_bits(d_v.get((B, 256)), want[1])
    for b in (d_b, d_p, d_v):
        b.free()
    nn.close()


def test_encode_device_equals_encode(observe_fixture):
    f = observe_fixture
    boards = ko.boards_from_fens([s.decode() for s in f["fen"]], f["ply"])
    nn = NN(filters=8, residuals=0)
    d_b = DevBuf(nn, boards.nbytes).put(boards)
    d_x = DevBuf(nn, len(boards) * 7680)
    st = torch.cuda.Stream()
    for sp in (None, C.c_void_p(st.cuda_stream)):
        assert nn._lib.kh_encode_device(nn.handle, d_b.p, len(boards), d_x.p, sp) == 0
        torch.cuda.synchronize(); assert nn._lib.kh_sync(nn.handle) == 0
        assert same_bits(d_x.get((len(boards), 1920)), f["obs"].astype(np.float32))
    d_b.free(); d_x.free(); nn.close()


_VARIANT_SCRIPT = r"""
import sys, numpy as np
from kami_amd import NN, weights as W
out = sys.argv[1]
res = {}
for F, B, dtype in ((119, 37, "bf16"), (30, 64, "bf16"), (119, 16, "f16")):
    nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype=dtype)
    nn.load_weights(W.random_weights(F, 64, 6, seed=11, peaky=8.0), 1)
    x = np.random.default_rng(F + B).random((B, 8, 8, F), dtype=np.float32)
    p, v, _ = nn.infer_full(x, want_logits=False)
    res[f"p_{F}_{B}_{dtype}"], res[f"v_{F}_{B}_{dtype}"] = p, v
    nn.close()
np.savez(out, **res)
"""


def test_both_whole_network_kernels_agree(tmp_path):
    """`tower8_kernel` (round 3, the default) against `tower_kernel` (rounds 1-2, KAMI_TOWER_V=4): the variant is read
    once per process, so each runs in a process of its own on the same weights and planes.  Same tiling and arithmetic;
    the 119-plane stem sums its reduction in another order (four unpadded passes instead of two padded halves) and the
    softmax another tree, and a different fp32 sum can round an activation to the neighbouring bf16: observed 2.0e-5
    on probabilities up to 0.3 (relative 1e-4) / 9e-5 on values with these peaky weights, bit-identical at 30 planes
    — three orders of magnitude inside the bf16 tolerances against the fp32 oracle (TOL_BY_DEPTH in test_gpu_parity.py:
    12 % relative on probabilities, 2e-3 on values), which every parity test checks on the default kernel."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for v in ("4", "8"):
        path = str(tmp_path / f"v{v}.npz")
        env = dict(os.environ, KAMI_TOWER_V=v, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _VARIANT_SCRIPT, path], env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[v] = np.load(path)
    for k in outs["4"].files:
        a, b = outs["4"][k], outs["8"][k]
        tol = 1e-4 if k.startswith("p_") else 1e-3
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert np.abs(a - b).max() <= tol, (k, float(np.abs(a - b).max()))
        if "_30_" in k:
            assert same_bits(a, b), k
