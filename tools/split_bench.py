"""Experiment: one 512-board step issued as S sub-batches on S streams (chip-level pipelining)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kami_amd import NN, weights as W, _lib as L
F, Cc, R, B = 119, 64, 6, 512
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype="bf16")
nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
lib = L.load()
x = torch.rand((B, 8, 8, F), device="cuda"); pol = torch.empty((B, 4672), device="cuda"); vf = torch.empty((B, 256), device="cuda")
for S in (1, 2, 4, 8, 16, 1, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    sub = B // S
    assert sub * S == B
    def step():
        for i, st in enumerate(streams):
            rc = lib.kh_infer_device(nn.handle, C.c_void_p(x[i*sub:].data_ptr()), sub, C.c_void_p(pol[i*sub:].data_ptr()),
                                     C.c_void_p(vf[i*sub:].data_ptr()), C.c_void_p(st.cuda_stream))
            assert rc == 0, L.last_error()
    for _ in range(30): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 400
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S}: {dt/K*1e6:.2f} us/step  {B*K/dt/1e6:.2f} M evals/s")
