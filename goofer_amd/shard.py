"""Sharding of independent notes across ranks (SURVEY.md §8 e): no data-path collective.

``note_range``   weak scaling — rank r renders ids [r*k, (r+1)*k).
``assign_lpt``   a fixed job (e.g. the 10 k-note render): greedy longest-processing-time assignment by
                 frame count, deterministic, so every rank computes the same plan without communicating.
``reduce_timing`` the only collectives of a run: MAX of the elapsed time, SUM of the frames rendered.
``gather_audio``  optional: ragged gather of the finished notes to one rank (RCCL over xGMI on the GPU box) — a job that
                 wants one process to write every wav; never part of the timed render.
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


def gather_audio(mix, lengths, dst: int = 0, group=None):
    """Ragged gather of finished audio to rank ``dst`` (SURVEY.md §8 e, "optional"): ``mix`` is this rank's concatenated
    fp32 notes, ``lengths`` their sample counts.  Returns on ``dst`` a list over ranks of (mix, lengths) and None elsewhere.
    Two collectives: an all-gather of the per-rank sizes (so every rank knows the padded shape) and one gather of the audio
    padded to the largest rank — point-to-point traffic into ``dst`` over its direct xGMI links, no ring."""
    import torch
    import torch.distributed as dist
    lengths = [int(v) for v in lengths]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [(mix, lengths)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = mix.device
    sizes = torch.tensor([int(mix.numel()), len(lengths)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [tuple(int(v) for v in t.tolist()) for t in all_sizes]
    max_n, max_k = max(s[0] for s in all_sizes), max(s[1] for s in all_sizes)
    pad = torch.zeros(max_n, dtype=mix.dtype, device=dev)
    pad[:mix.numel()] = mix
    lens = torch.zeros(max_k, dtype=torch.int64, device=dev)
    lens[:len(lengths)] = torch.tensor(lengths, dtype=torch.int64, device=dev)
    got_a = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    got_l = [torch.empty_like(lens) for _ in range(world)] if rank == dst else None
    dist.gather(pad, got_a, dst=dst, group=group)
    dist.gather(lens, got_l, dst=dst, group=group)
    if rank != dst:
        return None
    return [(got_a[r][:all_sizes[r][0]], [int(v) for v in got_l[r][:all_sizes[r][1]].tolist()]) for r in range(world)]
