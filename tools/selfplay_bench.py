"""Leaf evaluations per second through the WHOLE host path: search trees -> compact records ->
kh_encode_infer_legal -> priors -> expansion (tools/host_path_bench.py times the engine calls alone)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, search as S, _lib as L
F, C, R = 30, 64, 6
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
nn.load_weights(W.random_weights(F, C, R, seed=1, peaky=5.0), 1)
for games, threads, leaves, nodes in ((256, 1, 2, 800), (256, 4, 2, 800),      # BASELINE configs[1]: 256 games, 800 sims/move, batch-512 eval
                                     (512, 1, 1, 64), (2048, 8, 1, 64), (4096, 16, 1, 64), (8192, 16, 1, 64), (4096, 16, 2, 64),
                                     (2048, 16, 4, 64), (16384, 16, 1, 64), (8192, 32, 1, 64)):
    pool = S.Pool(nn, games=games, threads=threads, nodes=nodes, leaves_per_tree=leaves, seed=1)
    pool.run(min_evals=20000, max_seconds=10.0)          # warm-up
    s0 = pool.run(min_evals=0, max_seconds=0.0)
    e0, t0, g0, b0 = s0.evals, s0.seconds, s0.engine_seconds, s0.batches
    st = pool.run(min_evals=10**12, max_seconds=4.0)     # timed: 4 s of play
    de, dt = st.evals - e0, st.seconds - t0
    print(f"games {games:5d} threads {threads:2d} leaves/tree {leaves}: {de / dt:12,.0f} leaf-evals/s  mean batch {de / max(1, st.batches - b0):7.1f}  "
          f"in the engine call {100 * (st.engine_seconds - g0) / (dt * threads):4.1f} % of worker time "
          f"({1e6 * (st.engine_seconds - g0) / max(1, st.batches - b0):6.0f} us per call)  moves {st.moves}  games finished {st.games_finished} "
          f"(W {st.white_wins} / B {st.black_wins} / D {st.draws})", flush=True)
    del pool
