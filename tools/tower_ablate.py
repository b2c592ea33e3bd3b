"""Same-device, same-process A/B of several builds of the engine on the headline kernel (cdna_hip_programming.md §5.4
rule 24: N variants x M interleaved rounds in ONE process, median and min reported).
    python tools/tower_ablate.py [--dtype bf16] [--F 119] [--B 512] [--rounds 7] [--iters 2000] name=lib.so ...
The first library is the reference: every other one's outputs are compared with its bits (timing-only ablations differ
by construction; real variants must say 'bit-identical' or stay inside the parity tolerance)."""
import sys, os, ctypes as C, argparse, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import _lib as L, weights as W

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--F", type=int, default=119); ap.add_argument("--B", type=int, default=512)
ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--filters", type=int, default=64); ap.add_argument("--residuals", type=int, default=6)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    return lib


blob = W.random_weights(a.F, a.filters, a.residuals, seed=3, peaky=3.0)
x = np.random.default_rng(1).random((a.B, 8, 8, a.F), dtype=np.float32)
eng = []
for spec in a.libs:
    name, path = spec.split("=", 1)
    lib = bind(path)
    cfg = L.Config(8, 8, a.F, 4672, a.filters, a.residuals, L.DTYPES[a.dtype], 0, 0)
    h = C.c_void_p()
    assert lib.kh_create(C.byref(cfg), C.byref(h)) == 0, lib.kh_last_error()
    assert lib.kh_load_weights(h, blob.ctypes.data_as(C.c_void_p), blob.size, 1) == 0, lib.kh_last_error()
    d_in, d_p, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.kh_dev_alloc(h, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(h, a.B * 4672 * 4, C.byref(d_p)); lib.kh_dev_alloc(h, a.B * 256 * 4, C.byref(d_v))
    lib.kh_memcpy_h2d(h, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
    eng.append((name, lib, h, d_in, d_p, d_v, []))

ms = C.c_float()
for name, lib, h, d_in, d_p, d_v, t in eng:       # clocks up, scratch sized
    assert lib.kh_time_infer_device(h, d_in, a.B, d_p, d_v, a.iters, C.byref(ms)) == 0, lib.kh_last_error()
for rnd in range(a.rounds):
    for name, lib, h, d_in, d_p, d_v, t in eng:
        assert lib.kh_time_infer_device(h, d_in, a.B, d_p, d_v, a.iters, C.byref(ms)) == 0, lib.kh_last_error()
        t.append(ms.value * 1e3)
ref = None
print(f"# {a.dtype} F={a.F} B={a.B} {a.residuals}x{a.filters}  rounds={a.rounds} iters={a.iters}  (us per launch)")
for name, lib, h, d_in, d_p, d_v, t in eng:
    p = np.empty((a.B, 4672), np.float32); v = np.empty((a.B, 256), np.float32)
    lib.kh_memcpy_d2h(h, p.ctypes.data_as(C.c_void_p), d_p, p.nbytes); lib.kh_memcpy_d2h(h, v.ctypes.data_as(C.c_void_p), d_v, v.nbytes)
    if ref is None:
        ref = (p, v); cmp = "reference"
    elif np.array_equal(p.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(v.view(np.uint32), ref[1].view(np.uint32)):
        cmp = "bit-identical"
    else:
        with np.errstate(all="ignore"):
            cmp = f"differs: max|dp| {np.nanmax(np.abs(p - ref[0])):.3e} max|dv| {np.nanmax(np.abs(v - ref[1])):.3e} nan {int(np.isnan(p).sum())}"
    print(f"{name:24s} median {statistics.median(t):7.2f}  min {min(t):7.2f}  max {max(t):7.2f}   {cmp}", flush=True)
