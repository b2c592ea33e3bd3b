"""Why does bench.py see a slower kernel than tools/ab_bench.py?  Vary weights / buffers / torch one at a time."""
import sys, os, ctypes as C, numpy as np
if os.environ.get("PROBE_TORCH_FIRST"):
    import torch
    torch.cuda.init()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
F, Cc, R, B = 119, 64, 6, 512
lib = L.load()
def timeit(nn, d_in, d_p, d_v, n=5):
    r = []
    for _ in range(n):
        ms = C.c_float()
        assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 300, C.byref(ms)) == 0
        r.append(ms.value * 1e3)
    return np.median(r[1:])
def dev_buffers(nn, x):
    d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
    lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
    lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
    return d_in, d_p, d_v
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
for name, blob in (("ab weights (seed 1, peaky 20)", W.random_weights(F, Cc, R, seed=1, peaky=20.0)),
                   ("bench weights (seed 20240607)", W.random_weights(F, Cc, R, seed=20240607)),
                   ("seed 1, not peaky", W.random_weights(F, Cc, R, seed=1))):
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype="bf16")
    nn.load_weights(blob, 1)
    print(f"{name}: {timeit(nn, *dev_buffers(nn, x)):.2f} us", flush=True)
import torch
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype="bf16")
nn.load_weights(W.random_weights(F, Cc, R, seed=1, peaky=20.0), 1)
print(f"after import torch, hipMalloc buffers: {timeit(nn, *dev_buffers(nn, x)):.2f} us", flush=True)
tx = torch.from_numpy(x).cuda(); tp = torch.empty((B, 4672), device="cuda"); tv = torch.empty((B, 256), device="cuda")
torch.cuda.synchronize()
print(f"torch buffers: {timeit(nn, C.c_void_p(tx.data_ptr()), C.c_void_p(tp.data_ptr()), C.c_void_p(tv.data_ptr())):.2f} us", flush=True)
