"""tower2s_kernel (two workgroups per board pair, output channels split, halves exchanged per layer) against tower2b_kernel
on the same weights and planes: each in a process of its own (KAMI_WIDE_VARIANT is read once), outputs compared,
forward time of both.   python tools/split_check.py [R] [B] [dtype]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
from kami_amd import NN, weights as W
R, B, dtype, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
F, C = 119, 256
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype=dtype)
nn.load_weights(W.random_weights(F, C, R, seed=5, peaky=10.0), 1)
x = np.random.default_rng(B).random((B, 8, 8, F), dtype=np.float32)
p, vf, lg = nn.infer_full(x)
p2, vf2, lg2 = nn.infer_full(x)
assert np.array_equal(lg, lg2) and np.array_equal(vf, vf2), "not deterministic"
import ctypes as C
from kami_amd import _lib as L
lib = L.load()
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms = 0.0
for _ in range(3):
    t = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 40, C.byref(t)) == 0
    ms = t.value
np.savez(out, p=p, vf=vf, lg=lg, ms=ms)
""" % ROOT
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
res = {}
for v in ("6", "7"):
    path = os.path.join(tempfile.gettempdir(), f"split_{v}.npz")
    env = dict(os.environ)
    env["KAMI_WIDE_VARIANT"] = v             # 6: tower2b_kernel (slices summed 0,1,2,3), 7: tower2s_kernel
    r = subprocess.run([sys.executable, "-c", CHILD, str(R), str(B), dtype, path], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode:
        print("variant", v, "failed:", r.stderr[-1500:]); sys.exit(1)
    res[v] = np.load(path)
a, b = res["6"], res["7"]
print(f"20x256-like: R={R} B={B} {dtype}: tower2b_kernel {float(a['ms']):.3f} ms   tower2s_kernel {float(b['ms']):.3f} ms per forward")
for k in ("lg", "p", "vf"):
    d = np.abs(a[k] - b[k])
    print(f"  {k}: max |diff| {d.max():.3e}  mean {d.mean():.3e}  (max |value| {np.abs(a[k]).max():.3e})  identical rows {int((d.reshape(len(d), -1).max(1) == 0).sum())}/{len(d)}")
