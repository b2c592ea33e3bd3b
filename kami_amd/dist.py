"""One process per GPU: the little that the leaf-evaluation path needs from torch.distributed.

Leaf evaluations are independent, so ranks shard units (boards / games) and never
exchange data on the evaluation path; the only collectives are the barrier and the
max-over-ranks of the elapsed time that the benchmark contract asks for, and — outside the timed
path — the replay-record gather (replay.py).  Backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str | None = None, single_rank_group: bool = False):
    """Initialise the default process group when WORLD_SIZE > 1; returns torch.distributed or None.
    single_rank_group: also build the group for a lone process (a world of one still goes through the backend —
    how the RCCL code path is exercised on a one-GPU box)."""
    rank, local_rank, world = env_rank()
    if world <= 1 and not single_rank_group:
        return None
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def collective_device(dist) -> str:
    """Where a collective's tensors must live: RCCL ("nccl") only takes device tensors, gloo host tensors.
    Every helper below derives it from the group's backend instead of trusting a caller's default."""
    if dist is None:
        return "cpu"
    return "cuda" if str(dist.get_backend()).lower() == "nccl" else "cpu"


def shard(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n_units over world ranks -> (start, count) of this rank."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value: float, device: str | None = None) -> float:
    if dist is None:
        return value
    import torch
    device = device or collective_device(dist)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_weights(dist, blob, generation: int, src: int = 0, device: str | None = None):
    """New parameters from the trainer rank to every evaluator rank: the multi-process counterpart
    of `model->read(path)` after a candidate is accepted (selfplay.cpp:282-283).  One collective
    of the whole blob (2 MB for 6x64, 95 MB for 20x256) plus the generation; RCCL over xGMI with
    device tensors, gloo on CPU.  Returns (blob, generation) on every rank."""
    import numpy as np
    import torch
    if dist is None:
        return np.ascontiguousarray(blob, dtype=np.float32), generation
    device = device or collective_device(dist)
    meta = torch.tensor([generation, 0 if blob is None else int(np.asarray(blob).size)], dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src)
    n = int(meta[1].item())
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy(), int(meta[0].item())
