import sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden
from goofer_amd.device import Context
from oracle import goofer_ref as R
ctx = Context(0); ctx.plan(44100, 1024, 256)
f0s = [np.asarray(golden("sampler_" + nm)["f0_new"], dtype=np.float32) for nm in ("default", "t12g50")]
lens = [len(f) for f in f0s]
off = ctx.tensor(ctx.offsets(lens))
p = ctx.pulse_train(ctx.tensor(np.concatenate(f0s)), off).cpu().numpy()
print("pulse nan", np.isnan(p).sum(), "inf", np.isinf(p).sum(), "max", np.nanmax(np.abs(p)))
o = 0
for f in f0s:
    ref = R.pulse_train(f, 44100)
    d = np.abs(p[o:o+len(f)] - ref)
    print(" note err max", d.max(), "argmax", d.argmax(), "ref@", ref[d.argmax()], "got", p[o+d.argmax()])
    o += len(f)
Ts = [1 + n // 256 for n in lens]
S = ctx.rfft_frames(ctx.tensor(p), off, ctx.tensor(ctx.offsets(Ts)), sum(Ts)).cpu().numpy()
print("S nan", np.isnan(S).sum(), np.isinf(S).sum())
