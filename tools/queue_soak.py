"""Soak of the engine's queue under the self-play pool: long runs of the configurations the bench times for seconds, results
checked for sanity (evaluations counted, no engine error), and a two-engine pool.  python tools/queue_soak.py [seconds per case]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import NN, weights as W, search as S, _lib as L
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
F, C, R = 30, 64, 6
blob = W.random_weights(F, C, R, seed=1, peaky=5.0)
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
nn.load_weights(blob, 1)
for games, threads, leaves, nodes, pipe, target, wait in ((256, 14, 2, 800, 1, 512, 80), (256, 14, 4, 800, 3, 512, 80), (8192, 14, 1, 64, 1, 1024, 100), (64, 8, 8, 200, 4, 256, 40)):
    pool = S.Pool(nn, games=games, threads=threads, nodes=nodes, leaves_per_tree=leaves, seed=7, pipeline=pipe, coalesce_target=target, coalesce_wait_us=wait)
    t0 = time.perf_counter()
    st = pool.run(min_evals=10**12, max_seconds=secs)
    print(f"games {games} threads {threads} leaves {leaves} pipeline {pipe}: {st.evals:,} evaluations in {time.perf_counter() - t0:.1f} s = {st.evals / st.seconds / 1e6:.2f} M/s, "
          f"{st.moves} moves, {st.games_finished} games finished", flush=True)
    assert st.evals > 1e6 * secs
    del pool
nn2 = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
nn2.load_weights(blob, 1)
pool = S.Pool([nn, nn2], games=512, threads=14, nodes=400, leaves_per_tree=2, seed=9, pipeline=1, coalesce_target=512, coalesce_wait_us=80)
st = pool.run(min_evals=10**12, max_seconds=secs)
print(f"two engines, 512 games: {st.evals:,} evaluations = {st.evals / st.seconds / 1e6:.2f} M/s", flush=True)
pool.publish_weights(blob, 2)
st = pool.run(min_evals=10**12, max_seconds=5.0)
print("after a weight publish:", st.evals, "evaluations in all; soak ok", flush=True)
