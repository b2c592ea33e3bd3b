"""Throughput of the board -> plane encode kernel (Env::observe, env.h:202-262) against its
HBM-write roofline: 80 B read + 7 680 B written per position."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
nn = NN(filters=8, residuals=0)
lib = L.load()
rng = np.random.default_rng(0)
boards = np.zeros(N, dtype=L.BOARD_DTYPE)
boards["piece_occ"] = rng.integers(0, 2**63, (N, 6), dtype=np.uint64)
boards["color_occ"] = rng.integers(0, 2**63, (N, 2), dtype=np.uint64)
boards["ply"] = rng.integers(0, 400, N); boards["halfmove_clock"] = rng.integers(0, 60, N)
boards["ctm"] = rng.integers(0, 2, N); boards["castle_rights"] = rng.integers(0, 16, N)
d_b = C.c_void_p(); d_p = C.c_void_p()
assert lib.kh_dev_alloc(nn.handle, boards.nbytes, C.byref(d_b)) == 0
assert lib.kh_dev_alloc(nn.handle, N * 7680, C.byref(d_p)) == 0
lib.kh_memcpy_h2d(nn.handle, d_b, boards.ctypes.data_as(C.c_void_p), boards.nbytes)
ms = C.c_float()
assert lib.kh_time_encode_device(nn.handle, d_b, N, d_p, 20, C.byref(ms)) == 0
bytes_per = 80 + 7680
gbs = N * bytes_per / (ms.value * 1e-3) / 1e9
print({"positions": N, "ms_per_launch": round(ms.value, 4), "positions_per_s": round(N / (ms.value * 1e-3)),
       "GB_per_s": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000, 3), "frac_of_6.29TBs_measured": round(gbs / 6290, 3)})
