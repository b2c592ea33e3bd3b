"""Sustained load for tools/clock_probe.sh: one configuration's device-resident forward in a loop for N seconds.
    python tools/load_loop.py <filters> <residuals> <batch> <dtype> <seconds>"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W, _lib as L
Cc, R, B, dt, secs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5])
F = 119
lib = L.load()
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in, d_p, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B * 4672 * 4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B * 256 * 4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms = C.c_float()
reps = max(20, int(0.25 / (B * W.flops_per_eval(F, Cc, R) / 1.0e15)))          # about a quarter second per call
t0 = time.perf_counter(); last = 0.0
while time.perf_counter() - t0 < secs:
    assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, reps, C.byref(ms)) == 0, L.last_error()
    last = ms.value
flops = W.flops_per_eval(F, Cc, R) * B
print(f"{R}x{Cc} batch {B} {dt}: {last * 1e3:.1f} us per forward, {flops / (last * 1e-3) / 1e12:.0f} TFLOP/s = {flops / (last * 1e-3) / 1e12 / 2500:.3f} of 2 500")
