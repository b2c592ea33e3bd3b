"""kh_encode_infer_device with the board records in device memory vs in page-locked HOST memory (the kernel reads
them over PCIe itself): what zero-copy ingest costs per launch.  Also the launch + poll floor of an empty stream."""
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
host = torch.from_numpy(boards.view(np.uint8).reshape(B, 80).copy())
d_b = host.cuda()
h_b = host.pin_memory()
pol = torch.empty((B, 4672), device="cuda"); vf = torch.empty((B, 256), device="cuda")
st = torch.cuda.Stream(); sp = C.c_void_p(st.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def run(src, n):
    for _ in range(n):
        assert lib.kh_encode_infer_device(nn.handle, p(src), B, p(pol), p(vf), sp) == 0, L.last_error()
for name, src in (("device", d_b), ("pinned host", h_b), ("device", d_b), ("pinned host", h_b)):
    run(src, 300); st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); run(src, 2000); e1.record(st); st.synchronize()
    print(f"records in {name:12s}: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per launch (back to back)")
    # one launch at a time, host-side latency: launch -> poll until done
    lat = []
    for _ in range(300):
        t0 = time.perf_counter(); run(src, 1)
        while not st.query(): pass
        lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"   launch -> polled completion, median {lat[150] * 1e6:.1f} us, p10 {lat[30] * 1e6:.1f} us")
