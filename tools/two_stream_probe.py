"""Does a second stream hide the gap between dependent launches?  The headline kernel, K evaluations of independent
batches: on ONE stream, and alternating between TWO (each with its own output buffers).
    python tools/two_stream_probe.py [lib.so]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from kami_amd import _lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from kami_amd import NN, weights as W
lib = L.load()
B, F = 512, 119
nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16")
nn.load_weights(W.random_weights(F, 64, 6, seed=3, peaky=3.0), 1)
x = [torch.rand((B, 8, 8, F), device="cuda") for _ in range(2)]
p = [torch.empty((B, 4672), device="cuda") for _ in range(3)]
v = [torch.empty((B, 256), device="cuda") for _ in range(3)]
st = [torch.cuda.Stream() for _ in range(3)]


def run(nstreams, K):
    for i in range(K):
        s = i % nstreams
        rc = lib.kh_infer_device(nn.handle, C.c_void_p(x[i % 2].data_ptr()), B, C.c_void_p(p[s].data_ptr()), C.c_void_p(v[s].data_ptr()),
                                 C.c_void_p(st[s].cuda_stream))
        assert rc == 0, L.last_error()


for ns in (1, 2, 3, 1, 2, 3):
    run(ns, 20000); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(ns, 20000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {dt / 20000 * 1e6:.2f} us per evaluation, {B * 20000 / dt / 1e6:.2f} M evals/s", flush=True)
