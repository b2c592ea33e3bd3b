"""kh_train over several batches / epochs against the float64 restatement of tests/test_gpu_train.py, both conv paths;
repeated runs of one configuration (is a difference a race or arithmetic?), with and without the recorded graph."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from kami_amd import NN, weights as W
_first = NN(filters=8, residuals=0)      # the engine's HIP runtime up before torch is imported by the restatement
import test_gpu_train as T
F, C, R = 30, 64, 1
rng = np.random.default_rng(1)
blob = W.random_weights(F, C, R, seed=6, peaky=3.0)
N = 24
x = rng.random((N, 8, 8, F), dtype=np.float32)
obs_p = np.zeros((N, 4672), np.float32)
for i in range(N):
    idx = rng.choice(4672, 25, replace=False); v = rng.random(25).astype(np.float32); obs_p[i, idx] = v / v.sum()
obs_v = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), N)
for n, batch, epochs in ((12, 6, 2), (20, 8, 2)):
    want = T._float64_run(blob, F, C, R, x[:n], obs_p[:n], obs_v[:n], 0.005, epochs, batch)
    for valu, nograph in (("1", ""), ("1", ""), ("1", "1"), ("1", "1"), ("0", ""), ("1", "")):
        os.environ["KAMI_TRAIN_VALU"] = valu
        if nograph: os.environ["KAMI_TRAIN_NOGRAPH"] = "1"
        else: os.environ.pop("KAMI_TRAIN_NOGRAPH", None)
        nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="f32"); nn.load_weights(blob, 0)
        nn.train(x[:n], obs_p[:n], obs_v[:n], mlr=5, epochs=epochs, batchsize=batch)
        got = nn.get_weights(); off = 0; worst = (0, "")
        for tname, shape in W.tensor_specs(F, C, R):
            k = int(np.prod(shape))
            if "running" not in tname:
                e = float(np.abs(got[off:off+k] - want[off:off+k]).max()) / max(1e-3, float(np.abs(want[off:off+k]).max()))
                if e > worst[0]: worst = (e, tname)
            off += k
        print(n, batch, epochs, "valu" if valu == "1" else "mfma", "nograph" if nograph else "graph  ", worst, flush=True)
        nn.close()
