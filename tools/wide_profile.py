"""One wide-net configuration under rocprofv3 --kernel-trace --stats: per-layer-kernel durations.
   cd /tmp; rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/wide_profile.py 128 10 1024 bf16"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
lib = L.load()
Cc, R, B, dt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
REPS = int(sys.argv[5]) if len(sys.argv) > 5 else 100          # counter passes serialise kernels: a few launches are enough
F = 119
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
for _ in range(3):
    ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, max(2, REPS * 512 // B), C.byref(ms)) == 0
    print(f"{R}x{Cc} B={B} {dt}: {ms.value*1e3:.1f} us", flush=True)
