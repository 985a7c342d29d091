"""Sharding of independent notes across ranks (SURVEY.md §8 e): no data-path collective.

``note_range``   weak scaling — rank r renders ids [r*k, (r+1)*k).
``assign_lpt``   a fixed job (e.g. the 10 k-note render): greedy longest-processing-time assignment by
                 frame count, deterministic, so every rank computes the same plan without communicating.
``reduce_timing`` the only collectives of a run: MAX of the elapsed time, SUM of the frames rendered.
``gather_audio``  optional: ragged gather of the finished notes to one rank (grouped send / recv: RCCL over xGMI on the GPU
                 box) — a job that wants one process to write every wav; never part of the timed render.
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


def reduce_timing(elapsed_s: float, frames: int, device=None, always: bool = False):
    """(max elapsed over ranks, total frames).  Works on any initialised process group (nccl = RCCL on
    the GPU box, gloo in the CPU tests); a single process returns its own numbers — unless ``always`` asks for the two
    all-reduces anyway (the one-GPU test of the RCCL path: the same calls the 8-GPU run makes)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always):
        return float(elapsed_s), int(frames)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(f.item()))


def gather_audio(mix, lengths, dst: int = 0, group=None, always: bool = False):
    """Ragged gather of finished audio to rank ``dst`` (SURVEY.md §8 e, "optional"): ``mix`` is this rank's concatenated
    fp32 notes, ``lengths`` their sample counts.  Returns on ``dst`` a list over ranks of (mix, lengths) and None elsewhere.
    One all-gather of the per-rank sizes (sample count, note count), then grouped point-to-point transfers of exactly those
    sizes: every other rank sends its audio and its note lengths straight to ``dst``, which posts the matching receives in
    one batch (ncclGroupStart / End under RCCL: seven concurrent transfers over the seven direct xGMI links of ``dst`` on an
    8-GPU node, no ring, nothing padded)."""
    import torch
    import torch.distributed as dist
    lengths = [int(v) for v in lengths]
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always):
        return [(mix, lengths)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = mix.device
    sizes = torch.tensor([int(mix.numel()), len(lengths)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [tuple(int(v) for v in t.tolist()) for t in all_sizes]
    lens = torch.tensor(lengths, dtype=torch.int64, device=dev)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if rank != dst:
        ops = []
        if mix.numel():
            ops.append(dist.P2POp(dist.isend, mix.contiguous(), peer(dst), group))
        if lens.numel():
            ops.append(dist.P2POp(dist.isend, lens, peer(dst), group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    got_a = [mix if r == dst else torch.empty(all_sizes[r][0], dtype=mix.dtype, device=dev) for r in range(world)]
    got_l = [lens if r == dst else torch.empty(all_sizes[r][1], dtype=torch.int64, device=dev) for r in range(world)]
    ops = []
    for r in range(world):
        if r == dst:
            continue
        if all_sizes[r][0]:
            ops.append(dist.P2POp(dist.irecv, got_a[r], peer(r), group))
        if all_sizes[r][1]:
            ops.append(dist.P2POp(dist.irecv, got_l[r], peer(r), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [(got_a[r], [int(v) for v in got_l[r].tolist()]) for r in range(world)]
