"""Two batches in flight: steps alternate between two handles (own scratch, own streams), so the assembly of one batch can
run beside the walkers / gain pass of the other.  Usage (GPU box): python scripts/pipeline_time.py [notes] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctxs = [Context(0), Context(0)]
split = "--split" in sys.argv       # the two handles hold the two HALVES of one batch of `notes` notes (ids 0..n/2, n/2..n)
wls = [SamplerWorkload(c, 3, list(range(notes // 2 * i, notes // 2 * (i + 1))) if split else list(range(notes))) for i, c in enumerate(ctxs)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(n, depth):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for k in range(n):
        i = k % depth
        with torch.cuda.stream(streams[i]):
            wls[i].step()
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


for depth in (1, 2, 1, 2):
    run(4, depth)
    ms = run(steps, depth)
    print(f"depth {depth}: {ms:.3f} ms per launch" + (f" = {2 * ms:.3f} ms per {notes}-note batch" if split else ""))
