"""Step time of the default bench workload under combinations of the library's tuning options.
Usage (GPU box): python scripts/tune_step.py name=v1,v2,... [name=...] [--steps N] [--config C] [--zeros]   (cartesian product; profiling off)"""
import itertools
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

steps = 30
config = 3
axes = []
args = sys.argv[1:]
for i, a in enumerate(args):
    if a == "--steps":
        steps = int(args[i + 1])
    elif a == "--config":
        config = int(args[i + 1])
    elif "=" in a:
        k, v = a.split("=")
        axes.append((k, [int(x) for x in v.split(",")]))
ctx = Context(0)
wl = SamplerWorkload(ctx, config, list(range(1024)))
if "--zeros" in args:
    o = wl.renderer.run(wl.prep, seed=0, keep_stems=True)
    print("exact zeros: uv %.3f  bre %.3f  harm %.3f" % tuple(float((o[k] == 0).float().mean()) for k in ("uv", "bre", "harm")))
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        ctx.set_option(k, v)
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(steps):
            wl.step()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) / steps)
    print(" ".join("%s=%d" % (k, v) for (k, _), v in zip(axes, combo)), "-> %.3f ms" % best, flush=True)
