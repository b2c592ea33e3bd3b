"""Leaf evaluations per second through the WHOLE host path: search trees -> compact records ->
kh_encode_infer_legal / kh_submit_encode_infer_legal -> priors -> expansion (tools/host_path_bench.py times the engine
calls alone).  `pipe` rows keep two halves of every worker's trees in flight through the engine's coalescing queue."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, search as S, _lib as L
F, C, R = 30, 64, 6
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
nn.load_weights(W.random_weights(F, C, R, seed=1, peaky=5.0), 1)
CASES = [  # games, threads, leaves/tree, visits/move, pipeline, coalesce target, wait us
    # BASELINE configs[1] literally: 256 games, 800 sims/move, batch-512 eval
    (256, 1, 2, 800, 0, 0, 0), (256, 4, 2, 800, 0, 0, 0),                       # blocking calls (round 1's schedule)
    (256, 14, 2, 800, 1, 512, 200), (256, 14, 2, 800, 1, 512, 80),               # 512 positions in flight, queue + 2 launch lanes
    (256, 14, 4, 800, 1, 512, 200), (256, 14, 4, 800, 1, 512, 80), (256, 8, 4, 800, 1, 512, 80),   # 1024 in flight: two batch-512 evaluations
    (256, 14, 8, 800, 1, 512, 80),
    (256, 14, 2, 800, 3, 512, 80), (256, 14, 2, 800, 4, 512, 80), (256, 14, 2, 800, 4, 256, 40), (256, 14, 4, 800, 3, 512, 80), (256, 14, 4, 800, 4, 512, 80),   # three / four sets per worker
    # many trees, one leaf each (the reference's schedule, more games)
    (8192, 16, 1, 64, 0, 0, 0), (8192, 14, 1, 64, 1, 1024, 100), (4096, 14, 2, 64, 1, 1024, 100),
]
if os.environ.get("SP_CASE"):          # one ad-hoc case: "games,threads,leaves,visits,pipeline,target,wait"
    CASES = [tuple(int(x) for x in os.environ["SP_CASE"].split(","))]
elif len(sys.argv) > 2:
    CASES = CASES[int(sys.argv[1]):int(sys.argv[2])]
elif len(sys.argv) > 1:
    CASES = CASES[:int(sys.argv[1])]
for games, threads, leaves, nodes, pipe, target, wait in CASES:
    nn.set_coalesce(0, 0)
    pool = S.Pool(nn, games=games, threads=threads, nodes=nodes, leaves_per_tree=leaves, seed=1, pipeline=int(pipe),
                  coalesce_target=target, coalesce_wait_us=wait)
    pool.run(min_evals=20000, max_seconds=10.0)          # warm-up
    s0 = pool.run(min_evals=0, max_seconds=0.0)
    e0, t0, g0, b0 = s0.evals, s0.seconds, s0.engine_seconds, s0.batches
    l0, r0 = nn.coalesce_stats()
    st = pool.run(min_evals=10**12, max_seconds=3.0)     # timed: 3 s of play
    l1, r1 = nn.coalesce_stats()
    de, dt = st.evals - e0, st.seconds - t0
    launch = f"engine launches of {(r1 - r0) / max(1, l1 - l0):6.1f} positions" if l1 > l0 else f"engine calls of {de / max(1, st.batches - b0):6.1f} positions"
    print(f"games {games:5d} threads {threads:2d} leaves/tree {leaves} visits {nodes:3d} {('pipe%d' % max(2, pipe)) if pipe else 'sync '} target {target:4d}/{wait:3d}us: "
          f"{de / dt:12,.0f} leaf-evals/s  {launch}  worker submissions of {de / max(1, st.batches - b0):6.1f}  "
          f"waiting on the engine {100 * (st.engine_seconds - g0) / (dt * threads * (max(2, pipe) if pipe else 1)):4.1f} % "
          f"moves {st.moves} games finished {st.games_finished}", flush=True)
    del pool
