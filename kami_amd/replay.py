"""Replay records: the reference's in-process ring (kami/replaybuffer.h:10-92) and its multi-GPU
merge.  A record is (observation[obsize], mcts policy[psize], result) in fp32 — 26 372 bytes at
F = 30.  `add` / `select_batch` / `count` / `size` / `clear` keep the reference's meaning
(uniform selection WITH replacement over the whole ring, replaybuffer.h:61-84, including slots
not yet written — they read as zeros here where the reference reads uninitialised memory).

`gather` is new (the reference is single-process): every rank contributes the records it added
since the last gather; they are gathered to the root and merged into its ring in rank-major order (the reference's
insertion order is thread-racy, so any fixed order is acceptable — SURVEY §8e).  It runs over
torch.distributed: RCCL/xGMI with device tensors on GPUs, gloo on CPU.  Never on the timed
evaluation path.
"""
from __future__ import annotations

import numpy as np


class ReplayBuffer:
    def __init__(self, obsize: int, psize: int, bufsize: int, seed: int | None = None):
        self.obsize, self.psize, self.bufsize = obsize, psize, bufsize
        self.input_buffer = np.zeros((bufsize, obsize), np.float32)
        self.mcts_buffer = np.zeros((bufsize, psize), np.float32)
        self.result_buffer = np.zeros((bufsize,), np.float32)
        self.write_index = 0
        self.total = 0
        self._fresh = []                 # ring slots written since the last gather
        self._rng = np.random.default_rng(seed)

    def clear(self) -> None:             # replaybuffer.h:31-34
        self.total = 0
        self.write_index = 0
        self._fresh = []

    def add(self, inp, mcts, result: float) -> None:      # replaybuffer.h:36-56
        i = self.write_index
        self.input_buffer[i] = inp
        self.mcts_buffer[i] = mcts
        self.result_buffer[i] = result
        self._fresh.append(i)
        self.write_index = (i + 1) % self.bufsize
        self.total += 1

    def size(self) -> int:
        return self.bufsize

    def count(self) -> int:
        return self.total

    def select_batch(self, n: int):      # replaybuffer.h:61-84
        src = self._rng.integers(0, self.bufsize, n)
        return self.input_buffer[src].copy(), self.mcts_buffer[src].copy(), self.result_buffer[src].copy()

    # ---- multi-GPU merge ---------------------------------------------------------------
    def _pack_fresh(self) -> np.ndarray:
        idx = np.asarray(self._fresh[-self.bufsize:], dtype=np.int64)
        rec = np.empty((len(idx), self.obsize + self.psize + 1), np.float32)
        rec[:, :self.obsize] = self.input_buffer[idx]
        rec[:, self.obsize:self.obsize + self.psize] = self.mcts_buffer[idx]
        rec[:, -1] = self.result_buffer[idx]
        return rec

    def gather(self, dist, root: int = 0, device: str | None = None) -> int:
        """Merge every rank's fresh records into `root`'s ring.  Returns how many records the
        root inserted (0 on the other ranks)."""
        import torch
        mine = self._pack_fresh()
        self._fresh = []
        if dist is None:
            return 0
        from .dist import collective_device
        device = device or collective_device(dist)
        world, rank = dist.get_world_size(), dist.get_rank()
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[rank] = mine.shape[0]
        dist.all_reduce(counts)
        width = self.obsize + self.psize + 1
        nmax = int(counts.max().item())
        if nmax == 0:
            return 0
        pad = torch.zeros((nmax, width), dtype=torch.float32, device=device)
        pad[:mine.shape[0]] = torch.from_numpy(mine).to(device)
        # gather TO THE ROOT (only the trainer rank consumes the records: an all-gather would move world x the bytes,
        # which matters at 26 KB per dense row); fixed shapes, one collective
        outs = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
        dist.gather(pad, gather_list=outs, dst=root)
        inserted = 0
        if rank == root:
            for r in range(world):
                if r == root:
                    continue            # the root's own records are already in its ring
                recs = outs[r][:int(counts[r].item())].cpu().numpy()
                for rec in recs:
                    self.add(rec[:self.obsize], rec[self.obsize:self.obsize + self.psize], float(rec[-1]))
                    inserted += 1
            self._fresh = []
        return inserted


def gather_compact(dist, payload: bytes, record_bytes: int, root: int = 0, device: str | None = None):
    """Merge fixed-size compact records (ks_record, 664 bytes: board + sparse visit distribution + value —
    include/kami_search.h) over the ranks: 40x less traffic than the dense 26 372-byte rows `gather` moves.
    Every rank passes its own records; returns the list of every rank's payload (rank-major) on `root`, and
    [] elsewhere.  One size all-reduce + one gather of uint8 to the root (RCCL over xGMI with device tensors: the
    root's seven direct links carry one rank's payload each; gloo on CPU)."""
    import torch
    if dist is None:
        return [payload]
    from .dist import collective_device
    device = device or collective_device(dist)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert len(payload) % record_bytes == 0
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = len(payload)
    dist.all_reduce(counts)
    nmax = int(counts.max().item())
    if nmax == 0:
        return [b""] * world if rank == root else []
    pad = torch.zeros(nmax, dtype=torch.uint8, device=device)
    if payload:
        pad[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, gather_list=outs, dst=root)
    if rank != root:
        return []
    return [outs[r][:int(counts[r].item())].cpu().numpy().tobytes() for r in range(world)]
