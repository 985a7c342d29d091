"""Per-stage HIP-event times of the default bench workload (config 3, 1024 notes), one line per stage.
Usage (on the GPU box): python scripts/stage_times.py [notes] [steps] [--serial] [--opt name=value ...]
--serial turns the side stream off (option "overlap" = 0), so every stage is timed alone on the chip."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

serial = "--serial" in sys.argv
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
argv = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a]
notes = int(argv[0]) if len(argv) > 0 else 1024
steps = int(argv[1]) if len(argv) > 1 else 10
config = int(argv[2]) if len(argv) > 2 else 3
ctx = Context(0)
if serial:
    ctx.set_option("overlap", 0)
for k, v in opts:
    ctx.set_option(k, int(v))
wl = SamplerWorkload(ctx, config, list(range(notes)))
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
