import sys, json, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden
from goofer_amd.device import Context
from goofer_amd.core import note_params_from_kwargs, _fit
ctx = Context(0); ctx.plan(44100, 1024, 256)
SAMPLER = ["default", "t12g50", "tm12gm50", "formants", "formants_flip", "L0", "L1", "L2", "br_es_neg", "br_es_pos",
           "vel60", "vel150", "R1", "FV1_P50", "negcut"]
def load(names):
    envs, f0s, masks, forms, params, phis, env_len, lens = [], [], [], [], [], [], [], []
    for name in names:
        g = golden("sampler_" + name); kw = json.loads(str(g["kw"]))
        env = np.asarray(g["env_new"], dtype=np.float32); n = len(g["mask_new"]); T = 1 + n // 256
        envs.append(env.T); env_len.append(env.shape[1])
        f0s.append(np.asarray(g["f0_new"], dtype=np.float32)); masks.append(np.asarray(g["mask_new"], dtype=np.float32))
        forms.append(np.stack([_fit(r, env.shape[1]) for r in np.asarray(g["formants_new"], dtype=np.float64)], 1))
        params.append(note_params_from_kwargs(1, **kw))
        phis.append(np.random.default_rng(int(g["seed"][0])).uniform(0, 2*np.pi, size=(513, T)).astype(np.float32).T)
        lens.append(n)
    return envs, f0s, masks, forms, params, phis, env_len, lens
for names in (SAMPLER[:2], SAMPLER[:3], SAMPLER, SAMPLER[1:3]):
    envs, f0s, masks, forms, params, phis, env_len, lens = load(names)
    for use_phi, use_F in ((False, False), (True, True)):
        out = ctx.synth_batch(ctx.rows_from(np.concatenate(envs)), env_len, ctx.tensor(np.concatenate(f0s)),
                              ctx.tensor(np.concatenate(masks)), lens, np.concatenate(params),
                              formants=ctx.tensor(np.concatenate(forms)) if use_F else None,
                              phi=ctx.rows_from(np.concatenate(phis)) if use_phi else None)
        torch.cuda.synchronize()
        off = np.concatenate([[0], np.cumsum(lens)])
        rep = []
        for i, nm in enumerate(names):
            c = {k: int(torch.isnan(out[k][off[i]:off[i+1]]).sum()) for k in ("harm", "uv", "bre")}
            if any(c.values()): rep.append((nm, lens[i], env_len[i], c))
        print(len(names), "phi/F", use_phi, rep)
print("---- intermediates for [default, t12g50]")
envs, f0s, masks, forms, params, phis, env_len, lens = load(SAMPLER[:2])
out = ctx.synth_batch(ctx.rows_from(np.concatenate(envs)), env_len, ctx.tensor(np.concatenate(f0s)),
                      ctx.tensor(np.concatenate(masks)), lens, np.concatenate(params))
torch.cuda.synchronize()
for k in ("f0", "pulse", "S_harm", "S_uv", "S_breath", "frames", "env_harm", "env_noise", "mask_short", "note_mag", "note_peak", "onset_cnt", "frame_note", "row_src"):
    a = ctx.debug_fetch(k)
    fin = np.isfinite(a.view(np.float32) if a.dtype == np.complex64 else a) if a.dtype.kind in "fc" else np.ones(1, bool)
    print(k, a.shape, "nonfinite", int((~fin).sum()), "first", np.flatnonzero(~fin)[:3] if (~fin).any() else "", a[:4] if a.size < 40 else "")
