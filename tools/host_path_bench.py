"""Throughput of the HOST-buffer entry points (what kami's selfplay.cpp calls): PCIe-inclusive."""
import sys, os, time, threading, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, _lib as L

def rate(fn, n, iters=30):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    return n * iters / (time.perf_counter() - t0)

for F in (119, 30):
    nn = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16")
    nn.load_weights(W.random_weights(F, 64, 6, seed=1), 1)
    for B in (16, 64, 512):
        x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
        pol = np.empty((B, 4672), np.float32); val = np.empty(B, np.float32)
        print(f"F={F} B={B}: kh_infer {rate(lambda: nn.infer(x, B, pol, val), B):,.0f} evals/s (1 thread)")
        def worker(res, i):
            xx = x.copy(); pp = np.empty_like(pol); vv = np.empty_like(val)
            res[i] = rate(lambda: nn.infer(xx, B, pp, vv), B, 20)
        for T in (4,):
            res = [0] * T
            th = [threading.Thread(target=worker, args=(res, i)) for i in range(T)]
            [t.start() for t in th]; [t.join() for t in th]
            print(f"F={F} B={B}: kh_infer {sum(res):,.0f} evals/s ({T} threads)")
    if F == 30:
        B = 512
        rng = np.random.default_rng(1)                       # random occupancies: the encoder's cost does not depend on them
        boards = np.zeros(B, dtype=L.BOARD_DTYPE)
        boards["piece_occ"] = rng.integers(0, 2**63, (B, 6), dtype=np.uint64); boards["color_occ"] = rng.integers(0, 2**63, (B, 2), dtype=np.uint64)
        boards["ply"] = rng.integers(0, 300, B); boards["ctm"] = rng.integers(0, 2, B); boards["castle_rights"] = rng.integers(0, 16, B)
        offs = (np.arange(B + 1) * 20).astype(np.int32)
        acts = np.tile(np.array([584, 657, 730, 803, 876] * 4, np.int32), B)
        print(f"F=30 B=512: kh_encode_infer {rate(lambda: nn.encode_infer(boards), B):,.0f} evals/s")
        print(f"F=30 B=512: kh_encode_infer_legal {rate(lambda: nn.infer_legal(boards, offs, acts), B):,.0f} evals/s (reference value copy-out: pinned block + copy engine)")
        nn2 = NN(8, 8, F, 4672, filters=64, residuals=6, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
        nn2.load_weights(W.random_weights(F, 64, 6, seed=1), 1)
        print(f"F=30 B=512: kh_encode_infer_legal {rate(lambda: nn2.infer_legal(boards, offs, acts), B):,.0f} evals/s (one value per position: ONE launch, no copy engine)")
        for B2 in (16, 64):
            b2, o2, a2 = boards[:B2], offs[:B2 + 1], acts[:offs[B2]]
            print(f"F=30 B={B2}: kh_encode_infer_legal {rate(lambda: nn2.infer_legal(b2, o2, a2), B2, 200):,.0f} evals/s (one launch)")
