"""CPU: note sharding across ranks, incl. a world_size-2 gloo run of the timing reduction."""
import os
import socket

import numpy as np
import pytest

from goofer_amd import shard
from goofer_amd import synthetic as syn


def test_note_range_partitions():
    seen = []
    for r in range(8):
        seen += list(shard.note_range(r, 8, 1024))
    assert seen == list(range(8192))
    with pytest.raises(ValueError):
        shard.note_range(8, 8, 4)


def test_lpt_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    frames = np.exp(rng.uniform(np.log(18), np.log(520), 10000)).astype(int)     # config 4: 0.1 .. 3 s notes
    parts = shard.assign_lpt(frames, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(10000))
    loads = np.array([frames[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.001            # >= 6x at 8 GPUs needs <= 33 % imbalance; LPT gives ~0
    assert shard.assign_lpt(frames, 8) == parts          # deterministic: every rank derives the same plan


def test_notes_are_rank_independent():
    from goofer_amd.workload import assembled_note
    a = assembled_note(3, 1500)
    b = assembled_note(3, 1500)
    assert a["n"] == b["n"] and np.array_equal(a["f0"], b["f0"]) and np.array_equal(a["knots"], b["knots"])
    assert a["params"]["seed"][0][0] == 5000 + 1500


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.note_range(rank, world, 3)
    frames = sum(1 + int(float(syn.config_note(3, i)[1]["length"]) * 44.1 + 4410) // 256 for i in ids)
    dist.barrier()
    t, f = shard.reduce_timing(0.5 + rank, frames)
    q.put((rank, list(ids), frames, t, f))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_ranks_reduce_timing():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ids0, f0, t0, tot0), (r1, ids1, f1, t1, tot1) = res
    assert ids0 == [0, 1, 2] and ids1 == [3, 4, 5]
    assert t0 == t1 == 1.5                       # MAX over ranks
    assert tot0 == tot1 == f0 + f1               # SUM of frames


def _gather_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [5, 3] if rank == 0 else [2, 7, 4]                     # ragged notes, different counts per rank
    mix = torch.arange(sum(lens), dtype=torch.float32) + 100.0 * rank
    got = shard.gather_audio(mix, lens, dst=0)
    if rank == 0:
        q.put([(a.tolist(), l) for a, l in got])
    else:
        assert got is None
        q.put(None)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_ranks_gather_finished_audio():
    """The optional ragged gather of finished notes to one rank (what RCCL does over xGMI on the GPU box)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = next(r for r in res if r is not None)
    assert got[0] == ([float(v) for v in range(8)], [5, 3])
    assert got[1] == ([100.0 + v for v in range(13)], [2, 7, 4])


def _job_worker(rank, world, port, q):
    """What bench.py --job-notes does on every rank: the same LPT assignment from the frame counts of the whole job, derived
    without communication; then the per-rank frame totals are all-gathered (imbalance) and the timing reduced."""
    import torch
    import torch.distributed as dist
    from goofer_amd import synthetic as syn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    est = [syn.config_note_frames(4, i) for i in range(200)]
    mine = shard.assign_lpt(est, world)[rank]
    frames = sum(est[i] for i in mine)
    t = torch.tensor([float(frames)], dtype=torch.float64)
    allf = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allf, t)
    elapsed, total = shard.reduce_timing(1.0 + 0.25 * rank, frames)
    q.put((rank, mine, frames, [int(v.item()) for v in allf], elapsed, total))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_ranks_fixed_job_assignment():
    """A fixed job (BASELINE config 4: log-uniform note lengths) sharded over two ranks: every note rendered exactly once,
    the shares within 1 % of each other in frames, the same plan on both ranks, no data-path collective."""
    import torch.multiprocessing as mp
    from goofer_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ids0, f0, all0, t0, tot0), (r1, ids1, f1, all1, t1, tot1) = res
    assert sorted(ids0 + ids1) == list(range(200)) and not set(ids0) & set(ids1)
    assert all0 == all1 == [f0, f1]
    assert max(f0, f1) / (0.5 * (f0 + f1)) < 1.01
    assert t0 == t1 == 1.25 and tot0 == tot1 == f0 + f1
    est = [syn.config_note_frames(4, i) for i in range(200)]
    assert f0 + f1 == sum(est)


def test_frame_estimate_of_a_job_matches_the_planner():
    """config_note_frames (what the ranks balance a fixed job by) is the planner's frame count of the same note."""
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    for cfg, ids in ((3, [0, 1, 7]), (4, [0, 1, 2, 3, 50, 51, 999]), (5, [0, 1])):
        geo = syn.config_geometry(cfg)
        for i in ids:
            src, req, _ = syn.config_note(cfg, i)
            p = S.plan_note(S.decode_request(*syn.request_args(req)), src["sr"], src["y_len"], src["env_pack"]["knot_vals_log"].shape[1],
                            src["formants"], geo["hop"])
            assert 1 + p.n_out // geo["hop"] == syn.config_note_frames(cfg, i), (cfg, i)


@pytest.mark.parametrize("argv", [["--config", "4", "--job-notes", "1600", "--sub-batch", "128"], ["--config", "3", "--notes", "40"]])
def test_eight_rank_dress_rehearsal_of_bench(argv):
    """`bench.py --rehearse` under 8 gloo ranks, launched the way the driver launches the GPU bench: assignment, sub-batching,
    host planning, barrier + MAX/SUM reductions, per-rank gather, one JSON line from rank 0 (no GPU; the device step is a sleep)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(here, "bench.py"), "--rehearse", "--gpus", "8", "--steps", "2"] + argv
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=here)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                    # rank 0 alone prints, once
    line = json.loads(lines[0])
    assert line["rehearsal"] and line["n_gpus"] == 8 and len(line["per_rank_frames"]) == 8 and len(line["setup_seconds_per_rank"]) == 8
    job = "--job-notes" in argv
    assert line["scaling"] == ("strong" if job else "weak")
    total = sum(line["per_rank_frames"])
    if job:
        assert total == sum(syn.config_note_frames(4, i) for i in range(1600)) and line["imbalance"] < 1.01
    else:
        assert total == sum(syn.config_note_frames(3, i) for i in range(8 * 40)) and line["imbalance"] < 1.05
    assert abs(line["value"] - total * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`bench.py --gpus 2` started WITHOUT torch.distributed.run (WORLD_SIZE unset) must not measure one rank and print
    n_gpus 1: it starts the ranks itself as a child process, before anything touches a GPU, and exits with the child's status.
    A WORLD_SIZE that contradicts --gpus is refused."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(here, "bench.py"), "--rehearse", "--gpus", "2", "--steps", "2", "--notes", "12"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=here)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["per_rank_frames"]) == 2
    assert sum(line["per_rank_frames"]) == sum(syn.config_note_frames(3, i) for i in range(24))
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1"), cwd=here)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)
