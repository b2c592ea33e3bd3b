"""The N > 1 path on CPU: two processes, gloo backend (the GPU run uses the same code with RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np

from kami_amd import dist as kd
from kami_amd.replay import ReplayBuffer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_is_a_partition():
    for n in (0, 1, 7, 512, 2048, 2051):
        for world in (1, 2, 3, 8):
            parts = [kd.shard(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            pos = 0
            for s, c in parts:
                assert s == pos
                pos += c
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_replay_buffer_matches_reference_semantics():
    rb = ReplayBuffer(4, 3, 5, seed=0)                     # replaybuffer.h:10-92
    for i in range(7):                                     # wraps: ring of 5
        rb.add(np.full(4, i, np.float32), np.full(3, -i, np.float32), float(i))
    assert rb.count() == 7 and rb.size() == 5
    assert sorted(rb.result_buffer.tolist()) == [2, 3, 4, 5, 6]
    x, p, r = rb.select_batch(64)
    assert x.shape == (64, 4) and p.shape == (64, 3) and r.shape == (64,)
    assert np.array_equal(x[:, 0], r) and np.array_equal(p[:, 0], -r)
    rb.clear()
    assert rb.count() == 0


def test_two_rank_gloo(tmp_path):
    port = 29000 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=300, env=env, capture_output=True)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [r["world"] for r in res] == [2, 2]
    # weak-scaling shard: disjoint, covering, balanced
    assert res[0]["start"] == 0 and res[0]["count"] + res[1]["count"] == 2051
    assert res[1]["start"] == res[0]["count"]
    # the timed region is bracketed by barriers: every rank sees at least the slowest rank's sleep,
    # and the reported time is the max over ranks
    assert all(r["dt"] >= 0.099 for r in res)
    assert all(abs(r["dt_max"] - 0.10) < 1e-9 for r in res)
    # replay merge: root (3 own records) received rank 1's 4 records, rank-major order
    assert res[0]["inserted"] == 4 and res[0]["total"] == 7 and res[1]["inserted"] == 0
    assert res[0]["first_col"] == [0, 1, 2, 100, 101, 102, 103]
    assert res[0]["results"] == [0, 1, 2, 0, 1, 2, 3]
    # weight broadcast (selfplay.cpp:282-283 across processes): every rank ends with rank 0's blob
    from kami_amd import weights as W
    ref = W.random_weights(30, 8, 1, seed=77)
    assert all(r["wgen"] == 41 and r["wn"] == ref.size for r in res)
    assert all(abs(r["wsum"] - float(ref.astype(np.float64).sum())) < 1e-9 for r in res)
    # compact-record merge: rank-major on the root, nothing elsewhere
    assert res[0]["compact"] == [list(range(0, 8)), list(range(10, 22))] and res[1]["compact"] == []
