import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from kami_amd import NN, weights as W
rng = np.random.default_rng(0)
F, C, R, tb, n = 30, 64, 6, 8, 64
nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16")
nn.load_weights(W.random_weights(F, C, R, seed=1), 0)
x = rng.random((n, 8, 8, F), dtype=np.float32)
p = np.zeros((n, 4672), np.float32); p[np.arange(n), rng.integers(0, 4672, n)] = 1.0
v = rng.choice(np.array([-1, 0, 1], np.float32), n)
print(nn.train(x, p, v, epochs=2, batchsize=tb))
