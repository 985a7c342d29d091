"""N plain steps of the default workload (no profile events): for kernel traces of the step as the bench times it.
Usage (GPU box): python scripts/plain_steps.py [notes] [steps] [config] [name=value ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
argv = [a for a in sys.argv[1:] if "=" not in a]
notes = int(argv[0]) if len(argv) > 0 else 1024
steps = int(argv[1]) if len(argv) > 1 else 20
config = int(argv[2]) if len(argv) > 2 else 3
ctx = Context(0)
for k, v in opts:
    ctx.set_option(k, int(v))
wl = SamplerWorkload(ctx, config, list(range(notes)))
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    wl.step()
torch.cuda.synchronize()
print("step %.3f ms (wall, %d steps)" % ((time.perf_counter() - t0) / steps * 1e3, steps))
