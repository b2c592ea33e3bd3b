"""SGD steps per second of kh_train (csrc/train.hip) at the reference's training batch sizes.
`whole call`: steps / wall time of one NN::train call of 2 epochs over n samples (what round 1 reported: includes the
call's fixed work — workspace, parameter upload, recording the step graph, installing the trained weights in the engine);
`marginal`: (t(4 epochs) - t(2 epochs)) / extra steps: what one more SGD step costs."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W
rng = np.random.default_rng(0)
CASES = ((30, 64, 6, 8, 256), (30, 64, 6, 64, 1024), (30, 256, 2, 8, 128), (30, 128, 10, 32, 256), (30, 256, 20, 32, 128), (30, 256, 20, 64, 256))
for F, C, R, tb, n in CASES:
    nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16")
    nn.load_weights(W.random_weights(F, C, R, seed=1), 0)
    x = rng.random((n, 8, 8, F), dtype=np.float32)
    p = np.zeros((n, 4672), np.float32); p[np.arange(n), rng.integers(0, 4672, n)] = 1.0
    v = rng.choice(np.array([-1, 0, 1], np.float32), n)
    nn.train(x[:tb], p[:tb], v[:tb], epochs=1, batchsize=tb)          # warm-up
    t0 = time.perf_counter()
    first, last = nn.train(x, p, v, epochs=2, batchsize=tb)
    dt = time.perf_counter() - t0
    steps = 2 * (n // tb)
    t0 = time.perf_counter()
    nn.train(x, p, v, epochs=4, batchsize=tb)
    dt4 = time.perf_counter() - t0
    marg = steps / max(dt4 - dt, 1e-9)
    print(f"{R}x{C} F={F} batch {tb}: whole call {steps / dt:7.1f} steps/s {steps * tb / dt:9.0f} samples/s | marginal {marg:7.1f} steps/s "
          f"({1e3 / marg:.2f} ms per step, fixed {1e3 * (dt - steps / marg):.1f} ms per call)  loss {first:.3f} -> {last:.3f}", flush=True)
