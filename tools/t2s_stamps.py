"""tower2s_kernel, diagnostic build (tools/build_diag.sh): where a layer's time goes in a compute wave and in a mover wave.
Clocks (s_memtime, 100 MHz ticks x ... no: shader clocks) relative to the layer's start, median over workgroups."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["KAMI_WIDE_VARIANT"] = "7"
from kami_amd import NN, weights as W, _lib as L
L.LIB_PATH = os.path.abspath(os.environ.get("KAMI_AB_LIB", "kami_amd/csrc/build/libkamihip_diag.so"))
lib = L.load(); raw = C.CDLL(L.LIB_PATH)
F, Cc, R, B, dt = 119, 256, 20, 256, "f16"
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 30, C.byref(ms)) == 0
st = np.zeros((256, 64, 8), np.uint64)
assert raw.kh_debug_t2s_stamps(st.ctypes.data_as(C.c_void_p), st.size) == 0
st = st[:, :1 + 2 * R].astype(np.int64)
NL = 1 + 2 * R
rel = st - st[:, :, 0:1]
names = ["layer start", "compute: own half done (at [X])", "compute: past [X]", "compute: all slices done (at [R])", "compute: past [W]",
         "mover: own half published (flag stored)", "mover: partner's flag seen", "mover: partner's half in LDS"]
print(f"forward {ms.value*1e3:.1f} us; clocks since the layer's start, median over 256 workgroups, layers 2..{NL-1}")
for k in range(1, 8):
    v = np.median(rel[:, 2:, k], axis=0)
    print(f"  {names[k]:45s} median over layers {int(np.median(v)):7d}   (layer 2: {int(v[0])}, last: {int(v[-1])})")
print(f"  layer length (start -> next start): {int(np.median(st[:, 3:, 0] - st[:, 2:-1, 0]))}")
