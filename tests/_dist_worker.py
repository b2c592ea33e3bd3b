"""Worker for tests/test_dist_gloo.py: run under torch.distributed.run with the gloo backend."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kami_amd import dist as kd           # noqa: E402
from kami_amd.replay import ReplayBuffer  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = kd.env_rank()
    dist = kd.init("gloo")
    assert kd.collective_device(dist) == "cpu" and kd.collective_device(None) == "cpu"    # tensor placement follows the backend
    start, count = kd.shard(2048 + 3, rank, world)       # BASELINE config 4: 2048 games, uneven tail
    kd.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))                         # ranks finish at different times
    kd.barrier(dist)
    dt = time.perf_counter() - t0
    dt_max = kd.max_over_ranks(dist, 0.05 * (rank + 1))
    rb = ReplayBuffer(8, 5, 64, seed=rank)
    for i in range(3 + rank):                             # rank r holds 3 + r fresh records
        rb.add(np.full(8, 100 * rank + i, np.float32), np.full(5, rank, np.float32), float(i))
    ins = rb.gather(dist, root=0)
    from kami_amd.replay import gather_compact
    payload = bytes([10 * rank + k for k in range(4 * (2 + rank))])       # rank r holds 2 + r records of 4 bytes
    parts = gather_compact(dist, payload, 4, root=0)
    from kami_amd import weights as W
    blob = W.random_weights(30, 8, 1, seed=77) if rank == 0 else None
    got, gen = kd.broadcast_weights(dist, blob, 41 if rank == 0 else -1, src=0)
    wsum = float(np.asarray(got, dtype=np.float64).sum())
    res = {"rank": rank, "world": world, "start": start, "count": count, "dt": dt, "dt_max": dt_max,
           "inserted": ins, "wgen": gen, "wsum": wsum, "wn": int(got.size), "total": rb.count(), "results": rb.result_buffer[:rb.count()].tolist(),
           "first_col": rb.input_buffer[:rb.count(), 0].tolist(), "compact": [list(x) for x in parts]}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
