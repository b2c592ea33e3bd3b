"""Phase anatomy of the headline kernel from a -DKAMI_TOWER_STAMP=1 build (tools/build_variants.sh stamp:-DKAMI_TOWER_STAMP=1):
s_memtime stamps per wave at the phase boundaries, median over the 256 workgroups; in-kernel clock from s_memrealtime.
    python tools/tower_stamps.py kami_amd/csrc/build/libkamihip_stamp.so [--dtype bf16] [--F 119] [--B 512]"""
import sys, os, ctypes as C, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import _lib as L, weights as W

ap = argparse.ArgumentParser()
ap.add_argument("lib"); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--F", type=int, default=119); ap.add_argument("--B", type=int, default=512)
ap.add_argument("--v8", action="store_true", help="tower8_kernel build: compute waves' stamps in the first value row, helper waves' in the second")
a = ap.parse_args()
lib = C.CDLL(os.path.abspath(a.lib))
for name, (res, args) in L.SYMBOLS.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
cfg = L.Config(8, 8, a.F, 4672, 64, 6, L.DTYPES[a.dtype], 0, 0)
h = C.c_void_p()
assert lib.kh_create(C.byref(cfg), C.byref(h)) == 0
blob = W.random_weights(a.F, 64, 6, seed=3, peaky=3.0)
assert lib.kh_load_weights(h, blob.ctypes.data_as(C.c_void_p), blob.size, 1) == 0
x = np.random.default_rng(1).random((a.B, 8, 8, a.F), dtype=np.float32)
d_in, d_p, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
lib.kh_dev_alloc(h, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(h, a.B * 4672 * 4, C.byref(d_p)); lib.kh_dev_alloc(h, a.B * 256 * 4, C.byref(d_v))
lib.kh_memcpy_h2d(h, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
ms = C.c_float()
for _ in range(3):      # >= 2 s of back-to-back launches before the launch that is read
    assert lib.kh_time_infer_device(h, d_in, a.B, d_p, d_v, 20000, C.byref(ms)) == 0, lib.kh_last_error()
print(f"stamped build: {ms.value * 1e3:.2f} us per launch (do not quote: shares only)")
v = np.empty((a.B, 256), np.float32)
lib.kh_memcpy_d2h(h, v.ctypes.data_as(C.c_void_p), d_v, v.nbytes)
nwg = a.B // 2
rows = v.view(np.uint64).reshape(a.B, 128)
st = rows[0::2][:nwg].reshape(nwg, 4, 32).astype(np.int64)      # [wg][wave][stamp]
if a.v8:
    hs = rows[1::2][:nwg].reshape(nwg, 4, 32).astype(np.int64)
    t0 = st[:, :, 0].min(axis=1)[:, None]
    print("cycles since the workgroup's first compute-wave stamp (median over workgroups and the role's two waves):")
    for role, w0, items in (("plane waves", 2, ((0, "entry"), (10, "quarter 0 loads issued"), (11, "quarter 1 loads issued"), (12, "quarter 2 loads issued"), (13, "quarter 3 loads issued"), (6, "plane loads issued"), (14, "quarter 0 landed (STAMP=2 builds)"), (15, "quarter 1 landed"), (16, "quarter 2 landed"), (17, "quarter 3 landed"), (8, "quarter 0 converted"), (2, "[B0] passed"), (5, "stem steps done"), (19, "all steps done"), (22, "group done"))),
                            ("stream waves", 0, ((0, "entry"), (2, "[B0] passed"), (19, "all steps done"), (10, "fc: row requested behind [BL]"), (11, "fc: [BS1] passed"), (12, "fc: [BS2] passed"), (13, "fc: sums done"), (21, "value fc done (tanh, stores issued)"), (22, "group done"), (23, "drained")))):
        print(f" {role}:")
        for k, nm in items:
            print(f"   {nm:34s} {np.median(hs[:, w0:w0 + 2, k] - t0):8.0f}")
    print("compute waves:")
t = st[:, :, :24]
rt = st[:, :, 31] - st[:, :, 30]
clk = (t[:, :, 23] - t[:, :, 0]) / np.maximum(rt, 1) * 100.0
print(f"in-kernel clock: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}); "
      f"kernel span per wave: median {np.median(t[:, :, 23] - t[:, :, 0]):.0f} cycles = {np.median(rt) / 100.0:.2f} us")
names = ["entry", "ring issued+params", "ingest done", "stem start", "stem gemm done", "tower start"] + \
        [f"blk{r} conv{c} gemm done" for r in range(6) for c in (1, 2)] + \
        ["value conv done", "policy conv1 done", "policy conv2+logits", "softmax+store issued", "value fc + barrier", "drained"]
d = np.diff(t, axis=2)                    # [wg][wave][23]
print(f"{'phase ending at':28s} {'median':>8s} {'p10':>8s} {'p90':>8s}   cumulative(median)")
cum = 0
for k in range(23):
    m = np.median(d[:, :, k]); cum += m
    print(f"{names[k + 1]:28s} {m:8.0f} {np.percentile(d[:, :, k], 10):8.0f} {np.percentile(d[:, :, k], 90):8.0f}   {cum:8.0f}")
# skew of the workgroups' starts and ends on the chip (memtime is chip-wide)
s0 = t[:, 0, 0]; e0 = t[:, 0, 23]
print(f"workgroup starts spread over {s0.max() - s0.min()} cycles, ends over {e0.max() - e0.min()} cycles; first start -> last end {e0.max() - s0.min()} cycles")
