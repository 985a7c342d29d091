"""A/B of a library option on the default workload: bit equality of the mix and step / stage times.
Usage (GPU box): python scripts/ab_option.py name v0 v1 [notes] [config]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

name, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
notes = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
config = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = Context(0)
wl = SamplerWorkload(ctx, config, list(range(notes)))
outs = {}
for v in (v0, v1):
    ctx.set_option(name, v)
    o = wl.renderer.run(wl.prep, seed=0, keep_stems=True)
    torch.cuda.synchronize()
    outs[v] = {k: o[k].clone() for k in ("mix", "uv", "bre", "harm")}
for k in ("mix", "uv", "bre", "harm"):
    same = torch.equal(outs[v0][k], outs[v1][k])
    d = (outs[v0][k].double() - outs[v1][k].double())
    print(k, "bit-identical" if same else "DIFFERENT: max |d| = %g, rms d = %g (rms signal %g, max %g)" % (
        float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(outs[v0][k].double().pow(2).mean().sqrt()), float(outs[v0][k].abs().max())))
for rep in range(2):
    for v in (v0, v1):
        ctx.set_option(name, v)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        ctx.profile_begin(20)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            wl.step()
        t1.record()
        torch.cuda.synchronize()
        st = ctx.profile_end()
        print("%s=%d: step %.3f ms; " % (name, v, t0.elapsed_time(t1) / 20) + ", ".join("%s %.3f" % (k, x / st["steps"]) for k, x in st["ms"].items() if x > 0))
# ... and the plain step (no event records), alternating, five rounds
best = {v0: 1e9, v1: 1e9}
for rep in range(5):
    for v in (v0, v1):
        ctx.set_option(name, v)
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(30):
            wl.step()
        t1.record()
        torch.cuda.synchronize()
        best[v] = min(best[v], t0.elapsed_time(t1) / 30)
print("plain step, best of five alternating rounds: " + ", ".join("%s=%d %.3f ms" % (name, v, best[v]) for v in (v0, v1)))
