import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_net_fixture(path):
    z = np.load(path, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "x_u8" in d:
        d["x"] = d.pop("x_u8").astype(np.float32) / 256.0
    for k in ("features", "filters", "residuals", "generation"):
        d[k] = int(d[k])
    d["name"] = os.path.basename(path)[:-4]
    return d


NET_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "net_*.npz")))


@pytest.fixture(params=NET_FIXTURES, ids=[os.path.basename(p)[:-4] for p in NET_FIXTURES])
def net_fixture(request):
    return load_net_fixture(request.param)


@pytest.fixture(scope="session")
def observe_fixture():
    z = np.load(os.path.join(GOLDEN, "observe_playouts.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


# Observed parity maxima of the -m gpu run (max |dlogp|, |dlogit|, |dvalue| per dtype and configuration): collected by
# the tests through `record_maxima`, written to gpurun_out/parity_maxima.json at session end; the copy kept under
# profiles/ is what the tolerances in tests/test_gpu_parity.py::TOL are derived from (<= 2x observed).
_MAXIMA = {}


def record_maxima(key, **vals):
    cur = _MAXIMA.setdefault(key, {})
    for k, v in vals.items():
        cur[k] = max(float(v), cur.get(k, 0.0))


def pytest_sessionfinish(session, exitstatus):
    if not _MAXIMA:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_maxima.json"), "w") as f:
            json.dump({k: _MAXIMA[k] for k in sorted(_MAXIMA)}, f, indent=1)
    except OSError:
        pass
