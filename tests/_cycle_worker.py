"""Worker for test_selfplay_train_cycle_two_ranks: one generation of kami_amd/cycle.py per rank under
torch.distributed.run (gloo, both ranks on this box's one GPU)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kami_amd import NN, weights as W, _lib as L, dist as kd, search as S, cycle   # noqa: E402
from kami_amd.replay import ReplayBuffer                                           # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = kd.env_rank()
    dist = kd.init("gloo")
    F, C, R = 30, 16, 1
    nn = NN(8, 8, F, 4672, filters=C, residuals=R, dtype="bf16", value_mode=L.KH_VALUE_PER_SAMPLE0)
    nn.load_weights(W.random_weights(F, C, R, seed=5, peaky=3.0), 0)          # same start on every rank
    start, count = kd.shard(256, rank, world)                                   # the trees are sharded over the ranks
    pool = S.Pool(nn, games=count, threads=2, nodes=16, seed=100 + rank)
    replay = ReplayBuffer(cycle.OBSIZE, cycle.PSIZE, 4096, seed=rank)
    out = cycle.generation(nn, pool, replay, play_evals=60000, play_seconds=60.0, epochs=2, batchsize=8, sample=128, dist=dist)
    w = nn.get_weights()
    out.update(rank=rank, world=world, games=count, wsum=float(np.asarray(w, np.float64).sum()), replay_count=replay.count())
    out = {k: (int(v) if isinstance(v, (np.integer,)) else v) for k, v in out.items()}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    kd.barrier(dist)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
