"""Experiment: one 1024-note step as TWO half batches on two handles / streams, joined at the end of every step (no overlap
between steps), against the same notes as one batch on one handle.  Usage (GPU box): python scripts/split_step.py [notes] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = Context(0)
wl = SamplerWorkload(ctx, 3, list(range(notes)))


def timed(fn, k):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / k)
    return best


one = timed(wl.step, steps)
ctxs = [Context(0) for _ in range(parts)]
cut = [notes * k // parts for k in range(parts + 1)]
halves = [SamplerWorkload(c, 3, list(range(cut[k], cut[k + 1]))) for k, c in enumerate(ctxs)]
streams = [torch.cuda.Stream() for _ in range(parts)]


def split_step():
    cur = torch.cuda.current_stream()
    for st, h in zip(streams, halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            h.step()
    for st in streams:
        cur.wait_stream(st)


two = timed(split_step, steps)
print("one batch of %d notes: %.3f ms;  %d parts on %d handles, joined every step: %.3f ms (%.1f %%)" % (notes, one, parts, parts, two, 100.0 * (two / one - 1.0)))
