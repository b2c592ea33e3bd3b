"""One NN::train call per configuration with KAMI_TRAIN_TRACE=1: where a call's time goes (set-up, steps, read-back, installing
the trained weights in the serving layouts)."""
import sys, os, time
os.environ["KAMI_TRAIN_TRACE"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W
rng = np.random.default_rng(0)
for F, C, R, tb, n in ((30, 64, 6, 8, 256), (30, 128, 10, 32, 256)):
    nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16")
    nn.load_weights(W.random_weights(F, C, R, seed=1), 0)
    x = rng.random((n, 8, 8, F), dtype=np.float32)
    p = np.zeros((n, 4672), np.float32); p[np.arange(n), rng.integers(0, 4672, n)] = 1.0
    v = rng.choice(np.array([-1, 0, 1], np.float32), n)
    nn.train(x[:tb], p[:tb], v[:tb], epochs=1, batchsize=tb)
    for _ in range(3):
        t0 = time.perf_counter(); nn.train(x, p, v, epochs=2, batchsize=tb); dt = time.perf_counter() - t0
        print(f"{R}x{C} batch {tb}: {2 * (n // tb)} steps, whole call {dt * 1e3:.1f} ms", flush=True)
