"""One generation of kami's outer loop on this stack: self-play -> replay records -> NN::train -> new
generation (kami/selfplay.cpp:58-304 without the gating match of evaluate.cpp).

    play      kami_amd.search.Pool          MCTS trees -> kh_encode_infer_legal           (rows f1, f2)
    collect   gather_compact + ReplayBuffer finished games' positions, merged over ranks as 664-byte records (row f3)
    train     NN.train (kh_train)           the reference's SGD loop on the device        (row f4)
    publish   dist.broadcast_weights        every rank's evaluator gets the new generation (row f3)

Multi-GPU: every rank plays its own shard of the trees with its own engine, rank 0 trains."""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import dist as kd
from .replay import ReplayBuffer

OBSIZE, PSIZE = 8 * 8 * 30, 4672


def records_to_arrays(nn, records):
    """ks_record[] -> (planes [n,1920] fp32 via the device encoder, dense visit rows [n,4672], values [n])."""
    n = len(records)
    boards = np.frombuffer(b"".join(bytes(r.board) for r in records), dtype=L.BOARD_DTYPE) if n else np.zeros(0, L.BOARD_DTYPE)
    planes = nn.encode(boards).reshape(n, OBSIZE) if n else np.zeros((0, OBSIZE), np.float32)
    mcts = np.zeros((n, PSIZE), np.float32)
    vals = np.zeros(n, np.float32)
    for i, r in enumerate(records):
        mcts[i, list(r.actions[:r.nact])] = r.visits[:r.nact]
        vals[i] = r.value
    return planes, mcts, vals


def generation(nn, pool, replay: ReplayBuffer, *, play_evals: int, play_seconds: float = 60.0, sample: int | None = None,
               mlr: int = 5, epochs: int = 8, batchsize: int = 8, dist=None, device: str | None = None):
    """Play, collect, train (rank 0), publish.  Returns a dict of what happened.  The collectives' tensors live where
    the group's backend needs them (kd.collective_device: device memory for RCCL, host memory for gloo)."""
    device = device or kd.collective_device(dist)
    import ctypes as C
    from . import search as S
    from .replay import gather_compact
    st = pool.run(min_evals=play_evals, max_seconds=play_seconds)
    mine = pool.drain()
    # merge over the ranks as COMPACT records (664 B each); the root expands them (device encoder) into its ring
    payload = b"".join(bytes(r) for r in mine)
    merged = 0
    rank0 = dist is None or dist.get_rank() == 0
    for r, blob in enumerate(gather_compact(dist, payload, C.sizeof(S.Record), root=0, device=device)):
        recs = (S.Record * (len(blob) // C.sizeof(S.Record))).from_buffer_copy(blob)
        planes, mcts, vals = records_to_arrays(nn, recs)
        for i in range(len(vals)):
            replay.add(planes[i], mcts[i], float(vals[i]))                  # selfplay.cpp:176-184
        if dist is not None and r != dist.get_rank():
            merged += len(vals)
    if not rank0:
        merged = 0
    vals = mine
    rank = dist.get_rank() if dist is not None else 0
    out = {"evals": st.evals, "games_finished": st.games_finished, "records": len(vals), "merged": merged,
           "generation_before": nn.get_generation()}
    have = min(replay.count(), replay.size())
    if rank == 0 and have >= batchsize:
        n = sample or (have // batchsize) * batchsize
        src = replay._rng.integers(0, have, n)                               # replaybuffer.h:61-84, over the written slots
        first, last = nn.train(replay.input_buffer[src].reshape(n, 8, 8, 30), replay.mcts_buffer[src], replay.result_buffer[src],
                               mlr=mlr, epochs=epochs, batchsize=batchsize)
        out.update(first_loss=first, last_loss=last, trained_on=n)
    if dist is not None:
        blob, gen = kd.broadcast_weights(dist, nn.get_weights() if rank == 0 else None, nn.get_generation() if rank == 0 else 0,
                                         src=0, device=device)
        if rank != 0:
            nn.load_weights(blob, gen)
    out["generation_after"] = nn.get_generation()
    return out
