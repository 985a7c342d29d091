"""GPU: the noise branch that production (and bench.py) runs — Philox-4x32-7 phases + hardware sin / cos on the device —
as opposed to the injected-phase branch of the parity tests.  GOOFER.py:1148-1157: phi ~ U(0, 2 pi) per (bin, frame),
U = cos phi + i sin phi, S_uv = U * env_noise, S_breath = S_uv * HP (+ brightness / 5-tap blur on voiced frames).

The spectra are observable on the one-kernel-per-step path (option "stems" = 0, goofer_debug_fetch); the stem walkers that
the default path runs keep them in registers, so they are tied to these checks by (a) bit-equality of the two paths in
Philox mode and (b) the statistics of the finished unvoiced stem against the oracle's over many seeds."""
import numpy as np
import pytest

from conftest import rms_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SR, N_FFT, HOP, NB = 44100, 1024, 256, 513


@pytest.fixture(scope="module")
def ctx():
    from goofer_amd.device import Context
    c = Context(0)
    c.plan(SR, N_FFT, HOP)
    yield c
    c.close()


def _batch(ctx, n_notes=6, n=9000, voiced=False, seed_env=3):
    from goofer_amd.device import default_params
    rng = np.random.default_rng(seed_env)
    T = 1 + n // HOP
    f = np.arange(NB) * (SR / N_FFT)
    envs, f0s, masks = [], [], []
    for j in range(n_notes):
        e = np.exp(-f / 3000.0)[None, :] * (1.0 + 0.3 * rng.random((T, 1))) * (1.0 + 0.5 * np.exp(-0.5 * ((f - 900.0 - 200 * j) / 200.0) ** 2))[None, :]
        envs.append(e.astype(np.float32))
        m = np.ones(n, np.float32) if voiced else np.zeros(n, np.float32)
        masks.append(m)
        f0s.append((180.0 + 15.0 * j) * m)
    par = default_params(n_notes)
    par["seed"][:, 0] = np.arange(n_notes)                        # note ids key the phases
    return envs, f0s, masks, par, T, n


def _run(ctx, envs, f0s, masks, par, n, seed, stems):
    ctx.set_option("stems", stems)
    try:
        out = ctx.synth_batch(ctx.rows_from(np.concatenate(envs)), [e.shape[0] for e in envs], ctx.tensor(np.concatenate(f0s)),
                              ctx.tensor(np.concatenate(masks)), [n] * len(envs), par, seed=seed)
        torch.cuda.synchronize()
        res = {k: out[k].cpu().numpy() for k in ("harm", "uv", "bre", "mix")}
        if not stems:
            F = sum(e.shape[0] for e in envs)
            for name in ("S_uv", "S_breath"):
                res[name] = ctx.debug_fetch(name).reshape(F, (NB + 15) & ~15)[:, :NB].copy()      # rows are 128-byte aligned
    finally:
        ctx.set_option("stems", 1)
    return res


def test_unit_phasors_uniform_phases_and_one_u_for_both_stems(ctx):
    from oracle import goofer_ref as R
    envs, f0s, masks, par, T, n = _batch(ctx, n_notes=8, n=16000, voiced=False)
    r = _run(ctx, envs, f0s, masks, par, n, seed=1234, stems=0)
    env_n = np.concatenate([R.gauss1d(e.T.astype(np.float32), 1.75, axis=0).T for e in envs]).astype(np.float32)   # GOOFER.py:993
    U = r["S_uv"] / env_n
    # |U| = 1 to fp32 rounding (hardware sin / cos: 1.5e-6 absolute on each component)
    assert np.max(np.abs(np.abs(U) - 1.0)) < 1e-5
    # phases uniform on the circle: chi-square of a 64-cell histogram over ~250 k draws (63 dof: mean 63, sd 11.2)
    ph = np.angle(U[:, 1:NB - 1]).ravel()                          # DC / Nyquist imaginary parts are dropped by irfft anyway
    cells, _ = np.histogram(ph, bins=64, range=(-np.pi, np.pi))
    exp = ph.size / 64.0
    chi2 = float(np.sum((cells - exp) ** 2 / exp))
    assert chi2 < 63 + 6 * 11.2, chi2
    # first trigonometric moments vanish like 1 / sqrt(N)
    assert abs(np.mean(np.exp(1j * ph))) < 5.0 / np.sqrt(ph.size)
    assert abs(np.mean(np.exp(2j * ph))) < 5.0 / np.sqrt(ph.size)
    # no correlation along bins, along frames, or between notes (same frame index, different note id)
    Un = U.reshape(8, -1, NB)
    for a, b in ((Un[:, :, 1:-2], Un[:, :, 2:-1]), (Un[:, :-1, 1:-1], Un[:, 1:, 1:-1]), (Un[:-1, :, 1:-1], Un[1:, :, 1:-1])):
        c = np.mean(a * np.conj(b))
        assert abs(c) < 5.0 / np.sqrt(a.size), abs(c)
    # the breath spectrum is the SAME U times the real high-pass (unvoiced frames: no brightness, no blur)  GOOFER.py:1155-1157
    freqs = np.fft.rfftfreq(N_FFT, 1.0 / SR).astype(np.float32)
    hp = 1.0 / (1.0 + np.exp(-np.clip((freqs[None, :] - 0.0) / 5.0, -60, 60)))          # f0 = 0 on unvoiced frames
    ratio = r["S_breath"][:, 1:] / r["S_uv"][:, 1:]
    assert np.max(np.abs(ratio.imag)) < 1e-6
    assert np.max(np.abs(ratio.real - hp[:, 1:])) < 1e-5
    # another batch seed draws other phases; the same seed redraws the same
    r2 = _run(ctx, envs, f0s, masks, par, n, seed=1235, stems=0)
    assert abs(np.mean((r2["S_uv"] / env_n)[:, 1:-1] * np.conj(U[:, 1:-1]))) < 5.0 / np.sqrt(U[:, 1:-1].size)
    r3 = _run(ctx, envs, f0s, masks, par, n, seed=1234, stems=0)
    assert np.array_equal(r3["S_uv"], r["S_uv"])


def test_voiced_breath_spectrum_is_brightened_blurred_same_phasors(ctx):
    from oracle import goofer_ref as R
    envs, f0s, masks, par, T, n = _batch(ctx, n_notes=3, n=6000, voiced=True)
    r = _run(ctx, envs, f0s, masks, par, n, seed=77, stems=0)
    freqs = np.fft.rfftfreq(N_FFT, 1.0 / SR).astype(np.float32)
    bright = ctx.table(4)                                         # plan table 4: breath brightness curve (GOOFER.py:43)
    o = 0
    for j, e in enumerate(envs):
        Tn = e.shape[0]
        f0f = np.full(Tn, f0s[j][0], np.float32)
        hp = 1.0 / (1.0 + np.exp(-np.clip((freqs[None, :] - f0f[:, None]) / 5.0, -60, 60)))
        want = R.gauss1d((r["S_uv"][o:o + Tn] * hp * bright[None, :]).T, 0.5, axis=0).T          # GOOFER.py:1159-1173
        assert rms_err(r["S_breath"][o:o + Tn].view(np.float32), want.astype(np.complex64).view(np.float32)) < 2e-6 * np.abs(want).max()
        o += Tn


@pytest.mark.parametrize("voiced", [False, True])
def test_stem_walkers_draw_the_same_noise_as_the_checked_path(ctx, voiced):
    """Philox mode, default path (stem walkers) vs the path whose spectra the tests above inspect: the same bits."""
    envs, f0s, masks, par, T, n = _batch(ctx, n_notes=5, n=12000, voiced=voiced)
    if voiced:
        for m, f in zip(masks, f0s):
            m[:3000] = 0.0                                          # a transition: both noise stems are live somewhere
            f[:3000] = 0.0
    ctx.set_option("td_blur", 0)                                  # the 5-tap bin blur as a pass over the bins, like the checked path
    try:
        a = _run(ctx, envs, f0s, masks, par, n, seed=99, stems=1)
        b = _run(ctx, envs, f0s, masks, par, n, seed=99, stems=0)
    finally:
        ctx.set_option("td_blur", 1)
    for k in ("harm", "uv", "bre", "mix"):
        assert np.array_equal(a[k], b[k]), k
    # the default: that blur folded into the synthesis window of the voiced frames (a circular convolution of the spectrum is a
    # product in time).  Same audio up to the reference's 'reflect' handling of the two spectrum edges, which the product form
    # continues Hermitian and a six-bin correction in front of the transform puts back (goofer_plan: blur_edge): fp32 rounding
    d = _run(ctx, envs, f0s, masks, par, n, seed=99, stems=1)
    for k, tol in (("harm", 2e-6), ("uv", 2e-6), ("bre", 2e-6), ("mix", 2e-6)):        # (uv: through the peak gain only)
        err = float(np.max(np.abs(d[k] - a[k])))
        assert err <= tol * max(1.0, float(np.abs(a["mix"]).max())), (k, err)
    if voiced:
        assert not np.array_equal(d["bre"], a["bre"])             # (it IS another formulation)
    assert np.abs(a["uv"]).max() > 0 and np.all(np.isfinite(a["mix"]))
    assert (np.abs(a["bre"]).max() > 0) == voiced                 # an all-unvoiced note has no breath stem (mask smooths to 0)
    # and with the exact-zero transform skipping switched off (every frame runs both inverse transforms): the same values
    ctx.set_option("skip_zero", 0)
    try:
        c = _run(ctx, envs, f0s, masks, par, n, seed=99, stems=1)
    finally:
        ctx.set_option("skip_zero", 1)
    for k in ("harm", "uv", "bre", "mix"):
        assert np.array_equal(d[k], c[k]), k


def test_unvoiced_stem_band_power_matches_the_oracle_over_64_seeds(ctx):
    """Finished unvoiced stem of the default (walker, Philox) path against the oracle's (numpy default_rng phases) on the same
    envelope: power in eight frequency bands, averaged over 64 seeds.  Each side is a periodogram average over 7 analysis frames
    x 64 seeds (448 chi-square-2 terms per bin, 4.7 % per bin) and at most 64 bins per band — fewer effective bins where the
    envelope falls steeply inside a band — so the ratio of the two has a standard error of 0.8-1.5 %; the bound is 5 %
    (a wrong phase law, correlated bins or a wrong envelope blur move a band by tens of percent)."""
    from goofer_amd.device import default_params
    from oracle import goofer_ref as R
    n = 9600
    T = 1 + n // HOP
    f = np.arange(NB) * (SR / N_FFT)
    env = (np.exp(-f / 2500.0) * (1.0 + np.exp(-0.5 * ((f - 1500.0) / 300.0) ** 2)))[None, :].repeat(T, 0).astype(np.float32)
    mask = np.zeros(n, np.float32)
    f0 = np.zeros(n, np.float32)
    seeds = 64
    par = default_params(seeds)
    par["normalize"] = 0.0                                            # no peak normalisation: absolute power is compared
    par["seed"][:, 0] = np.arange(seeds)
    out = ctx.synth_batch(ctx.rows_from(np.concatenate([env] * seeds)), [T] * seeds, ctx.tensor(np.concatenate([f0] * seeds)),
                          ctx.tensor(np.concatenate([mask] * seeds)), [n] * seeds, par, seed=4321)
    torch.cuda.synchronize()
    uv = out["uv"].cpu().numpy().reshape(seeds, n)
    edges = np.linspace(0, NB - 1, 9).astype(int)

    def band_power(x):
        S = np.abs(np.fft.rfft(x[:, 1024:1024 + 7168].reshape(x.shape[0], 7, 1024) * np.hanning(1024), axis=-1)) ** 2
        P = S.mean(axis=(0, 1))
        return np.array([P[a:b].sum() for a, b in zip(edges[:-1], edges[1:])])

    ref = []
    for s in range(seeds):
        phi = np.random.default_rng(9000 + s).uniform(0.0, 2.0 * np.pi, size=(NB, T)).astype(np.float32)
        rec, harm, u, b = R.synthesize(env.T, f0, mask, np.empty(n, bool), SR, phi=phi, normalize=0.0)
        ref.append(u)
    pd, pr = band_power(uv), band_power(np.stack(ref))
    assert np.all(np.abs(pd / pr - 1.0) < 0.05), pd / pr
    # and the two agree in total power to 1 %
    assert abs(np.sum(uv.astype(np.float64) ** 2) / np.sum(np.stack(ref).astype(np.float64) ** 2) - 1.0) < 0.02


def test_unstored_zero_hops_give_the_same_mix():
    """Round 6: over a hop whose mask gain is one constant the noise walker does not STORE the stem that constant zeroes (1 KB per
    hop and stem); a byte per hop tells k_note_finish to take zeros.  With skip_zero 0 no hop is ever flat, every sample is stored
    and read: the two must give the same mix and — when the stems are asked for — the same stems, on sources with interior
    unvoiced gaps, fractional mask plateaus and ramps (hard sources), loops, reversed sources and notes of a few hundred samples
    (so that the finish pass's 16-byte groups straddle hop boundaries at every alignment)."""
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=HOP)
        jobs = []
        cases = [("C4", "100", "g10", "30", "900", "80", "40", "100", "0", "!120", "AA#5#AF#3#/+"),
                 ("A3", "60", "L1fa20", "10", "701", "120", "30", "90", "0", "!100", "AB#9#"),
                 ("G4", "100", "R1U20", "20", "653", "90", "50", "100", "0", "!120", "AA#20#"),
                 ("F4", "100", "L0", "40", "2903", "100", "60", "100", "0", "!90", "AA#60#"),
                 ("C5", "100", "t30", "40", "7", "2", "60", "100", "0", "!120", "AA"),
                 ("B3", "100", "FV1", "40", "333", "11", "60", "100", "0", "!120", "AA")]
        for k, args in enumerate(cases):
            src = syn.make_hard_source(78000 + k, seconds=0.5) if k % 2 else syn.with_unvoiced_gaps(syn.make_source(78000 + k, seconds=0.5), 0.4, 600 + k)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]), S.decode_request(*args)))
        for i in (0, 1, 5, 130):
            src, req, _ = syn.config_note(3, i, hard=bool(i % 2))
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
        prep = r.prepare(jobs, note_ids=list(range(len(jobs))))
        # the stems' arrays come from torch's caching allocator: make sure what it hands out is NaN, not fresh zero pages, so that
        # a sample read where nothing was stored shows
        junk = [torch.full((int(prep["samples"]) + 4096,), float("nan"), device="cuda") for _ in range(6)]
        del junk
        res = {}
        for skip in (1, 0):
            ctx.set_option("skip_zero", skip)
            try:
                for keep in (False, True):
                    o = r.run(prep, seed=21, keep_stems=keep)
                    ctx.check()
                    res[(skip, keep)] = {k: o[k].cpu().numpy().copy() for k in (("harm", "uv", "bre", "mix") if keep else ("mix",))}
            finally:
                ctx.set_option("skip_zero", 1)
        for keep in (False, True):
            for k in res[(1, keep)]:
                assert np.array_equal(res[(1, keep)][k], res[(0, keep)][k]), (keep, k)
        assert np.array_equal(res[(1, False)]["mix"], res[(1, True)]["mix"])
        uv, bre = res[(1, True)]["uv"], res[(1, True)]["bre"]
        assert np.isfinite(res[(1, True)]["mix"]).all() and (uv == 0).mean() > 0.2 and (bre == 0).mean() > 0.02 and np.abs(uv).max() > 0
    finally:
        ctx.close()
