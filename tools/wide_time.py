"""Forward time of one wide-net configuration (filters residuals batch dtype) under the library KAMI_AB_LIB names."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
if os.environ.get("KAMI_AB_LIB"): L.LIB_PATH = os.path.abspath(os.environ["KAMI_AB_LIB"])
lib = L.load()
Cc, R, B, dt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
F = 119
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
best = 1e9
for _ in range(4):
    ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, max(10, 60 * 256 // B), C.byref(ms)) == 0, L.last_error()
    best = min(best, ms.value)
flops = (1152 * F * Cc + 2304 * R * Cc * Cc + 16512 * Cc + 1228800) * B
print(f"{os.path.basename(os.environ.get('KAMI_AB_LIB', 'product')):22s} {R}x{Cc} B={B} {dt}: {best*1e3:8.1f} us  frac {flops/best/1e9/2500:.3f}", flush=True)
