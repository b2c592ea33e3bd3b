"""A/B of two builds of the engine on the headline kernel, on the same box: bits and time.
    python tools/tower_ab.py <libA.so> <libB.so>     (each library runs in its own process, A B A B)
    python tools/tower_ab.py --one <lib.so> <out.npz> (what those processes run)"""
import sys, os, subprocess, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np


def one(lib_path, out):
    from kami_amd import NN, weights as W, _lib as L
    L.LIB_PATH = os.path.abspath(lib_path)
    lib = L.load()
    res, times = {}, {}
    for dt, F, B in (("bf16", 119, 512), ("f16", 119, 512), ("bf16", 30, 512), ("bf16", 119, 2048), ("bf16", 119, 37)):
        nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype=dt)
        nn.load_weights(W.random_weights(F, 64, 6, seed=3, peaky=3.0), 1)
        x = np.random.default_rng(1).random((B, 8, 8, F), dtype=np.float32)
        d_in, d_p, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B * 4672 * 4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B * 256 * 4, C.byref(d_v))
        lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
        ms = C.c_float()
        for _ in range(3):
            assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 3000, C.byref(ms)) == 0, L.last_error()
        p = np.empty((B, 4672), np.float32); v = np.empty((B, 256), np.float32)
        lib.kh_memcpy_d2h(nn.handle, p.ctypes.data_as(C.c_void_p), d_p, p.nbytes); lib.kh_memcpy_d2h(nn.handle, v.ctypes.data_as(C.c_void_p), d_v, v.nbytes)
        key = f"{dt}_F{F}_B{B}"
        res["p_" + key] = p; res["v_" + key] = v; times[key] = ms.value * 1e3
        nn.close()
    np.savez(out, **res)
    print(" ".join(f"{k} {t:.2f}us" for k, t in times.items()), flush=True)


if sys.argv[1] == "--one":
    one(sys.argv[2], sys.argv[3])
else:
    a, b = sys.argv[1], sys.argv[2]
    os.makedirs("gpurun_out", exist_ok=True)
    for rnd in range(2):
        for tag, lib in (("A", a), ("B", b)):
            r = subprocess.run([sys.executable, __file__, "--one", lib, f"/tmp/kami_ab_{tag}.npz"], capture_output=True, text=True, timeout=280)
            print(tag, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
    A, B = np.load("/tmp/kami_ab_A.npz"), np.load("/tmp/kami_ab_B.npz")
    for k in A.files:
        same = np.array_equal(A[k].view(np.uint32), B[k].view(np.uint32))
        print(k, "bit-identical" if same else f"DIFFERENT: max |d| {np.abs(A[k] - B[k]).max():.3e}")
