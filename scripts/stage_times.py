"""Per-stage HIP-event times of the default bench workload (config 3, 1024 notes), one line per stage.
Usage (on the GPU box): python scripts/stage_times.py [notes] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
wl = SamplerWorkload(ctx, 3, list(range(notes)))
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
ctx.profile_begin(steps)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    wl.step()
t1.record()
torch.cuda.synchronize()
st = ctx.profile_end()
tot = 0.0
for k, v in st["ms"].items():
    if v > 0:
        print(f"{k:16s} {v / st['steps']:.3f} ms")
        tot += v / st["steps"]
print(f"sum of synth stages (two streams, they overlap) {tot:.3f} ms;  step (assemble + synth, wall) {t0.elapsed_time(t1) / steps:.3f} ms")
