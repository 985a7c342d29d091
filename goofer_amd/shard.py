"""Sharding of independent notes across ranks (SURVEY.md §8 e): no data-path collective.

``note_range``   weak scaling — rank r renders ids [r*k, (r+1)*k).
``assign_lpt``   a fixed job (e.g. the 10 k-note render): greedy longest-processing-time assignment by
                 frame count, deterministic, so every rank computes the same plan without communicating.
``reduce_timing`` the only collectives of a run: MAX of the elapsed time, SUM of the frames rendered.
"""
from __future__ import annotations

import heapq

import numpy as np


def note_range(rank: int, world: int, per_rank: int) -> range:
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return range(rank * per_rank, (rank + 1) * per_rank)


def assign_lpt(frame_counts, world: int) -> list:
    """Greedy LPT: notes sorted by (-frames, id), each to the currently lightest rank (ties -> lowest
    rank).  Returns ``world`` sorted id lists; every id appears exactly once."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    heap = [(0, r) for r in range(world)]
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(frame_counts[i]), r))
    return [sorted(x) for x in out]


def reduce_timing(elapsed_s: float, frames: int, device=None):
    """(max elapsed over ranks, total frames).  Works on any initialised process group (nccl = RCCL on
    the GPU box, gloo in the CPU tests); a single process returns its own numbers."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), int(frames)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(f.item()))
