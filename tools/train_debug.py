"""Debug aid: SGD steps of kh_train against a plain PyTorch (CPU, float64) restatement of nn.cpp:59-105, for the matrix-core
kernels and (KAMI_TRAIN_VALU=1) the order-exact VALU kernels:  python tools/train_debug.py F C R B steps"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.nn.functional as Fn
from kami_amd import NN, weights as W

def torch_step(blob, F, C, R, x, obs_p, obs_v, lr):
    ts, off = {}, 0
    x, obs_p, obs_v = x.astype(np.float64), obs_p.astype(np.float64), obs_v.astype(np.float64)
    for name, shape in W.tensor_specs(F, C, R):
        k = int(np.prod(shape))
        t = torch.tensor(blob[off:off + k].reshape(shape).astype(np.float64))
        if "running" not in name: t.requires_grad_(True)
        ts[name] = t; off += k
    def convbn(h, conv, bn, pad):
        h = Fn.conv2d(h, ts[conv + ".weight"], ts[conv + ".bias"], padding=pad)
        return Fn.batch_norm(h, ts[bn + ".running_mean"], ts[bn + ".running_var"], ts[bn + ".weight"], ts[bn + ".bias"], True, 0.1, 1e-5)
    h = torch.tensor(x).permute(0, 3, 1, 2)
    h = torch.relu(convbn(h, "conv1", "batchnorm1", 1))
    for i in range(R):
        r = f"residual{i}"
        t = torch.relu(convbn(h, r + ".conv1", r + ".batchnorm1", 1))
        h = h + torch.relu(convbn(t, r + ".conv2", r + ".batchnorm2", 1))
    ph = torch.relu(convbn(h, "policyconv", "pbatchnorm", 0))
    ph = Fn.conv2d(ph, ts["policyconv2.weight"], ts["policyconv2.bias"])
    ph = ph.permute(0, 2, 3, 1).flatten(1)
    p = torch.exp(torch.log_softmax(ph, 1))
    vh = torch.relu(convbn(h, "valueconv", "vbatchnorm", 0)).flatten(1)
    v = torch.tanh(Fn.linear(vh, ts["valuefc.weight"], ts["valuefc.bias"]))
    loss = -(torch.tensor(obs_p) * torch.log(p + 0.001)).sum() + Fn.mse_loss(v, torch.tensor(obs_v).reshape(-1, 1).expand_as(v))
    loss.backward()
    out = []
    for name, shape in W.tensor_specs(F, C, R):
        t = ts[name]
        out.append((t - lr * t.grad).detach().numpy().ravel() if t.requires_grad else t.numpy().ravel())
    return np.concatenate(out), float(loss)

F, C, R, B, steps = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (30, 16, 1, 8, 1)
rng = np.random.default_rng(0)
blob = W.random_weights(F, C, R, seed=4, peaky=3.0)
x = rng.random((B, 8, 8, F), dtype=np.float32)
obs_p = np.zeros((B, 4672), np.float32)
for i in range(B):
    idx = rng.choice(4672, 30, replace=False); v = rng.random(30).astype(np.float32); obs_p[i, idx] = v / v.sum()
obs_v = rng.choice(np.array([-1, 0, 1], np.float32), B)
want = blob.astype(np.float64)
for _ in range(steps):
    want, tl = torch_step(want, F, C, R, x, obs_p, obs_v, 0.005)
    # the running statistics are buffers the functional batch_norm updated in place inside torch_step's tensors: redo by hand
res = {}
for valu in ("0", "1"):
    os.environ["KAMI_TRAIN_VALU"] = valu
    nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="f32")
    nn.load_weights(blob, 0)
    nn.train(x, obs_p, obs_v, mlr=5, epochs=steps, batchsize=B)
    res[valu] = nn.get_weights()
print(f"F={F} C={C} R={R} B={B} steps={steps}: max over tensors of |err| / tensor scale, parameters only (running statistics aside)")
off = 0
worst = {"0": 0.0, "1": 0.0, "d": 0.0}
for name, shape in W.tensor_specs(F, C, R):
    k = int(np.prod(shape))
    if "running" not in name:
        b = want[off:off + k]; scale = max(1e-3, np.abs(b).max())
        e0 = np.abs(res["0"][off:off + k] - b).max() / scale; e1 = np.abs(res["1"][off:off + k] - b).max() / scale
        d = np.abs(res["0"][off:off + k] - res["1"][off:off + k]).max() / scale
        worst["0"] = max(worst["0"], e0); worst["1"] = max(worst["1"], e1); worst["d"] = max(worst["d"], d)
        if max(e0, e1) > 1e-6: print(f"  {name:32s} matrix cores {e0:.2e}  VALU {e1:.2e}  between them {d:.2e}")
    off += k
print(f"worst: matrix cores vs float64 {worst['0']:.2e} | VALU vs float64 {worst['1']:.2e} | matrix cores vs VALU {worst['d']:.2e}")
