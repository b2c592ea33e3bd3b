"""Compact ingest on device: records -> policy.  Fused (encoder inside the forward kernel) vs
encode kernel + forward kernel, 512 positions, 6x64 bf16, HIP events via back-to-back launches."""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from kami_amd import NN, weights as W, _lib as L
B, F = 512, 30
nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16")
nn.load_weights(W.random_weights(F, 64, 6, seed=1), 1)
lib = L.load()
rng = np.random.default_rng(0)
boards = np.zeros(B, dtype=L.BOARD_DTYPE)
boards["piece_occ"] = rng.integers(0, 2**63, (B, 6), dtype=np.uint64); boards["color_occ"] = rng.integers(0, 2**63, (B, 2), dtype=np.uint64)
boards["ply"] = rng.integers(0, 300, B); boards["ctm"] = rng.integers(0, 2, B); boards["castle_rights"] = rng.integers(0, 16, B)
d_b = torch.from_numpy(boards.view(np.uint8).reshape(B, 80)).cuda()
planes = torch.empty((B, 8, 8, F), device="cuda"); pol = torch.empty((B, 4672), device="cuda"); vf = torch.empty((B, 256), device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def fused():
    assert lib.kh_encode_infer_device(nn.handle, p(d_b), B, p(pol), p(vf), st) == 0, L.last_error()
def split():
    assert lib.kh_encode_device(nn.handle, p(d_b), B, p(planes), st) == 0
    assert lib.kh_infer_device(nn.handle, p(planes), B, p(pol), p(vf), st) == 0
for name, fn in (("fused", fused), ("encode+infer", split), ("fused", fused), ("encode+infer", split)):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 400
    for _ in range(K): fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:13s}: {dt / K * 1e6:.2f} us per 512 positions  {B * K / dt / 1e6:.2f} M evals/s")
