import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
lib = L.load()
for F, Cc, R, B, dt in ((119, 128, 10, 512, "bf16"), (119, 256, 20, 256, "f16"), (119, 256, 20, 512, "f16"), (119, 128, 10, 2048, "bf16")):
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
    nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
    x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
    d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
    lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
    lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
    r = []
    for _ in range(5):
        ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 60, C.byref(ms)) == 0; r.append(ms.value)
    print(f"{R}x{Cc} B={B} {dt}: {np.median(r[1:])*1e3:.1f} us", flush=True)
