"""Step time of the default workload against the share of unvoiced audio: the synthetic notes of BASELINE config 3 are fully
voiced behind their 50 ms offset, so the noise walker skips the unvoiced stem's transform on every frame (its gain is exactly
zero there).  This script re-runs the same batch with (a) the skipping switched off, (b) a fraction of every source's
voicing mask zeroed in blocks (consonant-like gaps), skipping on.  Usage (GPU box): python scripts/voicing_sweep.py [notes]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from goofer_amd import synthetic as syn
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def time_step(wl, steps=20):
    for _ in range(10):
        wl.step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        out = wl.step()
    t1.record()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["mix"]).all())
    return t0.elapsed_time(t1) / steps


ctx = Context(0)
wl = SamplerWorkload(ctx, 3, list(range(notes)))
time_step(wl)
for rep in range(2):
    print(f"fully voiced, skipping on : {time_step(wl):.3f} ms")
    ctx.set_option("skip_zero", 0)
    print(f"fully voiced, skipping off: {time_step(wl):.3f} ms")
    ctx.set_option("skip_zero", 1)
del wl
for share in (0.1, 0.2, 0.4):
    wl = SamplerWorkload(ctx, 3, list(range(notes)), unvoiced_share=share)
    voiced = float((wl.prep["f0"] > 0).float().mean())
    print(f"{share:.0%} of the source unvoiced in 50 ms gaps (assembled mask {voiced:.0%} voiced): {time_step(wl):.3f} ms")
    del wl
