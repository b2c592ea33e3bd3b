"""Where the time of a coalesced launch goes (KAMI_CO_TRACE=1 prints the queue's own stamps when an engine is destroyed)."""
import sys, os, time
os.environ["KAMI_CO_TRACE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W, search as S, _lib as L
F, C, R = 30, 64, 6
for games, threads, leaves, target, wait in ((256, 14, 2, 512, 200), (256, 14, 2, 512, 80), (256, 8, 2, 512, 80), (256, 14, 4, 512, 200), (256, 14, 4, 512, 80), (256, 8, 4, 512, 80), (256, 14, 8, 512, 80)):
    nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
    nn.load_weights(W.random_weights(F, C, R, seed=1, peaky=5.0), 1)
    pool = S.Pool(nn, games=games, threads=threads, nodes=800, leaves_per_tree=leaves, seed=1, pipeline=True, coalesce_target=target, coalesce_wait_us=wait)
    pool.run(min_evals=20000, max_seconds=10.0)
    s0 = pool.run(min_evals=0, max_seconds=0.0)
    st = pool.run(min_evals=10**12, max_seconds=2.0)
    print(f"games {games} threads {threads} leaves {leaves}: {(st.evals - s0.evals) / (st.seconds - s0.seconds):,.0f} evals/s", flush=True)
    del pool
    nn.close()
# the synchronous call alone, one thread, for comparison
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
nn.load_weights(W.random_weights(F, C, R, seed=1, peaky=5.0), 1)
rng = np.random.default_rng(0)
B = 512
boards = np.zeros(B, dtype=L.BOARD_DTYPE)
boards["piece_occ"] = rng.integers(0, 2**63, (B, 6), dtype=np.uint64); boards["color_occ"] = rng.integers(0, 2**63, (B, 2), dtype=np.uint64)
nact = rng.integers(10, 50, B); offs = np.concatenate([[0], np.cumsum(nact)]).astype(np.int32); acts = rng.integers(0, 4672, int(offs[-1])).astype(np.int32)
for _ in range(50): nn.infer_legal(boards, offs, acts)
t0 = time.perf_counter()
for _ in range(300): nn.infer_legal(boards, offs, acts)
print(f"kh_encode_infer_legal, 1 thread, 512: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call (python overhead included)")
