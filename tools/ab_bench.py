"""In-process A/B of tower-kernel experiment variants (KAMI_TOWER_DBG bits), interleaved rounds."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L
if os.environ.get("KAMI_AB_LIB"):      # time another build of the library on the same box (tools only)
    L.LIB_PATH = os.path.abspath(os.environ["KAMI_AB_LIB"])
# a variant is "DBG" or "DBGsSTAGGER" (KAMI_TOWER_DBG bits, KAMI_TOWER_STAGGER value)
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
def select(v):
    d, _, st = v.partition("s")
    os.environ["KAMI_TOWER_DBG"] = d
    if st: os.environ["KAMI_TOWER_STAGGER"] = st
    else: os.environ.pop("KAMI_TOWER_STAGGER", None)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
F, Cc, R, B = 119, 64, 6, 512
nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype="bf16")
nn.load_weights(W.random_weights(F, Cc, R, seed=1, peaky=20.0), 1)
lib = L.load()
x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        select(v)
        ms = C.c_float()
        assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, 300, C.byref(ms)) == 0
        res[v].append(ms.value * 1e3)
for v in variants:
    a = np.array(res[v][1:])
    print(f"DBG {v:>9s}: median {np.median(a):.2f} us  min {a.min():.2f}  max {a.max():.2f}")
# correctness of the variants that claim to be result-preserving (pass e.g. "check" as 3rd arg)
if len(sys.argv) > 3:
    outs = {}
    for v in variants:
        select(v)
        p, vf, _ = nn.infer_full(x, want_logits=False)
        outs[v] = (p.copy(), vf.copy())
    for v in variants[1:]:
        print(f"DBG {v}: policy bit-identical to DBG {variants[0]}: {np.array_equal(outs[v][0], outs[variants[0]][0])}, value: {np.array_equal(outs[v][1], outs[variants[0]][1])}")
