import sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden
from goofer_amd.device import Context
from goofer_amd.core import note_params_from_kwargs, _fit
ctx = Context(0); ctx.plan(44100, 1024, 256)
for name in ["default", "t12g50", "tm12gm50", "formants"]:
    g = golden("sampler_" + name); kw = json.loads(str(g["kw"]))
    env = np.asarray(g["env_new"], dtype=np.float32); n = len(g["mask_new"]); T = 1 + n // 256
    F = np.stack([_fit(r, env.shape[1]) for r in np.asarray(g["formants_new"], dtype=np.float64)], 1)
    par = note_params_from_kwargs(1, **kw)
    print(name, "env", env.shape, "T", T, "n", n, "fs", par["formant_shift"], par["f_shift"], "f0 range", g["f0_new"].min(), g["f0_new"].max(),
          "env finite", np.isfinite(env).all(), "F finite", np.isfinite(F).all())
    phi = np.random.default_rng(int(g["seed"][0])).uniform(0, 2*np.pi, size=(513, T)).astype(np.float32).T
    for use_phi, use_F in ((False, False), (True, False), (False, True), (True, True)):
        out = ctx.synth_batch(ctx.rows_from(env.T), [env.shape[1]], ctx.tensor(np.asarray(g["f0_new"], dtype=np.float32)),
                              ctx.tensor(np.asarray(g["mask_new"], dtype=np.float32)), [n], par,
                              formants=ctx.tensor(F) if use_F else None, phi=ctx.rows_from(phi) if use_phi else None)
        torch.cuda.synchronize()
        print("   phi", use_phi, "F", use_F, {k: int(torch.isnan(out[k]).sum()) for k in ("harm", "uv", "bre")})
