"""Worker for the RCCL tests (tests/test_gpu_parity.py): the replay merge, the weight broadcast and the bench's
max-over-ranks through torch.distributed's "nccl" backend (= RCCL on ROCm) with DEVICE tensors.  Runs as one rank per
GPU under torch.distributed.run when the box has >= 2 GPUs, and as a world of ONE rank on a one-GPU box (the group is
still an RCCL communicator: same backend, same tensor placement rules, no peer)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kami_amd import dist as kd, weights as W            # noqa: E402
from kami_amd.replay import ReplayBuffer, gather_compact  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = kd.env_rank()
    dist = kd.init("nccl", single_rank_group=True)
    assert dist is not None and str(dist.get_backend()).lower() == "nccl"
    assert kd.collective_device(dist) == "cuda"
    kd.barrier(dist)
    dt_max = kd.max_over_ranks(dist, 0.05 * (rank + 1))
    rb = ReplayBuffer(8, 5, 64, seed=rank)
    for i in range(3 + rank):
        rb.add(np.full(8, 100 * rank + i, np.float32), np.full(5, rank, np.float32), float(i))
    ins = rb.gather(dist, root=0)                                          # replaybuffer.h:36-84 across ranks
    payload = bytes([10 * rank + k for k in range(4 * (2 + rank))])
    parts = gather_compact(dist, payload, 4, root=0)
    blob = W.random_weights(30, 8, 1, seed=77) if rank == 0 else None
    got, gen = kd.broadcast_weights(dist, blob, 41 if rank == 0 else -1, src=0)   # selfplay.cpp:282-283 across ranks
    res = {"rank": rank, "world": world, "backend": str(dist.get_backend()), "dt_max": dt_max, "inserted": ins, "total": rb.count(),
           "compact": [list(x) for x in parts], "wgen": gen, "wn": int(got.size), "wsum": float(np.asarray(got, np.float64).sum())}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    kd.barrier(dist)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
