"""One training configuration for rocprofv3 --kernel-trace --stats:  python3 tools/train_profile.py C R batch [n]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W
F = 30
C, R, tb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8 * tb
rng = np.random.default_rng(0)
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16")
nn.load_weights(W.random_weights(F, C, R, seed=1), 0)
x = rng.random((n, 8, 8, F), dtype=np.float32)
p = np.zeros((n, 4672), np.float32); p[np.arange(n), rng.integers(0, 4672, n)] = 1.0
v = rng.choice(np.array([-1, 0, 1], np.float32), n)
nn.train(x[:tb], p[:tb], v[:tb], epochs=1, batchsize=tb)
t0 = time.perf_counter()
nn.train(x, p, v, epochs=2, batchsize=tb)
dt = time.perf_counter() - t0
print(f"{R}x{C} batch {tb}: {2 * (n // tb) / dt:.1f} steps/s")
