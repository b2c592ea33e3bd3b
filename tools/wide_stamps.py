"""Where a workgroup of the wide-net 3x3 skip layer spends its time (diagnostic build: KAMI_DIAG=1 python
kami_amd/build.py, library selected with KAMI_AB_LIB).  Stamps (s_memtime, shader clocks): 0 entry, 1 image
staged (before the barrier), 2 barrier passed, 3 last MFMA issued, 4 ring drained, 5 stores issued."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
if os.environ.get("KAMI_AB_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["KAMI_AB_LIB"])
lib = L.load()
raw = C.CDLL(L.LIB_PATH)
for Cc, R, B, dt in ((128, 10, 1024, "bf16"), (256, 20, 512, "f16")):
    F = 119
    nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
    nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
    x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
    d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
    lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
    lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
    ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 50, C.byref(ms)) == 0
    four = os.environ.get("KAMI_WIDE_VARIANT") == "4"
    nwg = ((B + 3) // 4) * (Cc // 128) if four else (B // 2) * (Cc // 64)
    st = np.zeros((2048, 8), np.uint64)
    assert raw.kh_debug_wide_stamps(st.ctypes.data_as(C.c_void_p), 2048 * 8) == 0
    st = st[:min(nwg, 2048)].astype(np.int64)
    t0 = st[:, 0].min()
    d = np.diff(st[:, :6], axis=1)
    span = st[:, 5].max() - t0
    print(f"{R}x{Cc} B={B} {dt}: {ms.value*1e3:.1f} us per forward; last skip layer: {nwg} workgroups, launch span {span} clocks")
    print("  median clocks per phase: stage %d | barrier wait %d | MFMA loop %d | ring drain %d | epilogue %d | total %d" %
          (*np.median(d, axis=0), np.median(st[:, 5] - st[:, 0])))
    hw = st[:, 7]
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; xcc = (hw >> 20) & 0xf   # HW_ID fields (gfx9: CU_ID 11:8, SH 12, SE 15:13)
    # rounds: workgroups ordered by start time; how many start before the first one ends
    order = np.argsort(st[:, 0])
    first_end = st[:, 5].min()
    print("  workgroups started before the first one finished:", int((st[:, 0] < first_end).sum()),
          "| start-time quartiles (clocks after first):", np.percentile(st[:, 0] - t0, [25, 50, 75, 100]).astype(int))
    # gap between a workgroup's end and the next start on the same (xcc, se, cu) slot
    key = (hw & 0xffffff00)
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        idx = idx[np.argsort(st[idx, 0])]
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append(st[b, 0] - st[a, 5])
    if gaps:
        print("  end -> next start on the same CU: median %d clocks (n=%d)" % (np.median(gaps), len(gaps)))

