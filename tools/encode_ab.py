"""Same-process A/B of builds of the encode kernel: positions/s and GB/s at several batch sizes.
    python tools/encode_ab.py name=lib.so ..."""
import sys, os, ctypes as C, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import _lib as L
libs = []
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    lib = C.CDLL(os.path.abspath(path))
    for n, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, n); fn.restype = res; fn.argtypes = args
    cfg = L.Config(8, 8, 30, 4672, 8, 0, L.KH_BF16, 0, 0)
    h = C.c_void_p(); assert lib.kh_create(C.byref(cfg), C.byref(h)) == 0
    libs.append((name, lib, h))
rng = np.random.default_rng(0)
for N in (512, 8192, 1 << 17, 1 << 20):
    boards = np.zeros(N, dtype=L.BOARD_DTYPE)
    boards["piece_occ"] = rng.integers(0, 2**63, (N, 6), dtype=np.uint64); boards["color_occ"] = rng.integers(0, 2**63, (N, 2), dtype=np.uint64)
    boards["ply"] = rng.integers(0, 400, N); boards["ctm"] = rng.integers(0, 2, N)
    iters = max(20, min(5000, (1 << 24) // N))
    res = {}
    bufs = []
    for name, lib, h in libs:
        d_b, d_p = C.c_void_p(), C.c_void_p()
        assert lib.kh_dev_alloc(h, boards.nbytes, C.byref(d_b)) == 0 and lib.kh_dev_alloc(h, N * 7680, C.byref(d_p)) == 0
        lib.kh_memcpy_h2d(h, d_b, boards.ctypes.data_as(C.c_void_p), boards.nbytes)
        bufs.append((d_b, d_p)); res[name] = []
    ms = C.c_float()
    for rnd in range(6):
        for (name, lib, h), (d_b, d_p) in zip(libs, bufs):
            assert lib.kh_time_encode_device(h, d_b, N, d_p, iters, C.byref(ms)) == 0
            if rnd: res[name].append(ms.value)
    print(f"N={N}: " + "  ".join(f"{n} {statistics.median(t) * 1e3:.2f} us = {N * 7760 / statistics.median(t) / 1e6:.0f} GB/s" for n, t in res.items()), flush=True)
    for (name, lib, h), (d_b, d_p) in zip(libs, bufs):
        lib.kh_dev_free(h, d_b); lib.kh_dev_free(h, d_p)
