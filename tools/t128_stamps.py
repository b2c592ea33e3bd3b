"""tower128_kernel, diagnostic build (kami_amd/csrc/build/libkamihip_diag.so, -DKAMI_WIDE_DIAG): clocks of a layer's MFMA loop
and of its wave-local boundary, median over the workgroups, per layer."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
L.LIB_PATH = os.path.abspath(os.environ.get("KAMI_AB_LIB", "kami_amd/csrc/build/libkamihip_diag.so"))
lib = L.load(); raw = C.CDLL(L.LIB_PATH)
F, Cc, R, B, dt = (119, 256, 20, 512, "f16") if len(sys.argv) > 1 and sys.argv[1] == "256" else (119, 128, 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, "bf16")
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 50, C.byref(ms)) == 0
st = np.zeros((256, 64, 4), np.uint64)
assert raw.kh_debug_t128_stamps(st.ctypes.data_as(C.c_void_p), st.size) == 0
full = st.astype(np.int64)
st = st[:, :1 + 2 * R].astype(np.int64)
loop = np.median(st[:, :, 1] - st[:, :, 0], axis=0); bnd = np.median(st[:, :, 2] - st[:, :, 1], axis=0)
gap = np.median(st[:, 1:, 0] - st[:, :-1, 2], axis=0)
print(f"forward {ms.value*1e3:.1f} us; per layer (median over workgroups): MFMA loop {loop.astype(int).tolist()}")
print(f"boundary {bnd.astype(int).tolist()}")
print(f"  of which ReLU / skip / round / image writes {np.median(st[:, :, 3] - st[:, :, 1], axis=0).astype(int).tolist()}")
print(f"  of which next layer's shifts {np.median(st[:, :, 2] - st[:, :, 3], axis=0).astype(int).tolist()}")
print(f"kernel span per workgroup (first loop start -> last boundary end): {int(np.median(st[:, -1, 2] - st[:, 0, 0]))} clocks; sum loops {int(loop.sum())} boundaries {int(bnd.sum())}")

NLh = 1 + 2 * R
hs = full[:, NLh], full[:, NLh + 1]
if np.median(hs[0][:, 0]) > 0:
    t0 = full[:, NLh - 1, 2]                                   # end of the last boundary
    names = (("heads start", hs[0][:, 0]), ("policy conv 1 done", hs[0][:, 1]), ("policy conv 2 done", hs[0][:, 2]), ("softmax sums done", hs[0][:, 3]),
             ("rows through LDS, stores issued", hs[1][:, 0]), ("value FC done, stores drained", hs[1][:, 1]))
    print("heads inside the launch (clocks since the last boundary's end, median over workgroups): " + "; ".join(f"{n} {int(np.median(v - t0))}" for n, v in names))
