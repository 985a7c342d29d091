"""Lean hand-off of goofer_render_batch (goofer_assembly.lean, assemble.hip: lean_out): the f0 / mask kernel writes the smoothed
decimated voicing mask (smooth_mask_ds, GOOFER.py:556-562) and the per-frame (f0, mask) picks (GOOFER.py:1104-1106) itself and the
per-sample mask is never materialised.  Option "lean" 0 restores the written mask + k_mask_short + the map kernel's picks: the two
must agree bit for bit — knots, picks, every stem and the mix — on every kind of note the assembly knows."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from goofer_amd import synthetic as syn  # noqa: E402


def _jobs(hop_config=3):
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    jobs = []

    def add(src, args):
        jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*args)))

    # (pitch, velocity, flags, offset, length, consonant, cutoff, volume, modulation, tempo, pitch string)
    cases = [("C4", "100", "g10", "30", "900", "80", "40", "100", "0", "!120", "AA#5#AF#3#/+"),
             ("A3", "60", "L1fa20", "10", "700", "120", "30", "90", "0", "!100", "AB#9#"),          # velocity stretch (vel_active)
             ("D4", "160", "L2B30", "50", "500", "60", "-200", "100", "0", "!140", "AA"),           # faster consonant, negative cutoff
             ("G4", "100", "R1U20", "20", "650", "90", "50", "100", "0", "!120", "AA#20#"),         # reversed source
             ("E4", "100", "FV1", "0", "400", "50", "20", "100", "0", "!120", "AA#3#AC"),           # force-voiced
             ("F4", "100", "L0", "40", "2900", "100", "60", "100", "0", "!90", "AA#60#"),           # long loop
             ("C5", "100", "t30", "40", "5", "2", "60", "100", "0", "!120", "AA"),                  # a few hundred samples
             ("B3", "100", "es40br-30", "40", "1161", "100", "60", "100", "0", "!120", "AA")]
    for k, args in enumerate(cases):
        src = syn.make_source(77000 + k, seconds=0.5)
        if k % 2:
            src = syn.with_unvoiced_gaps(src, 0.4, 500 + k)
        add(src, args)
    for i in (0, 1, 5, 130, 700):
        src, req, _ = syn.config_note(hop_config, i)
        add(syn.with_unvoiced_gaps(src, 0.3, 900 + i) if i % 2 else src, syn.request_args(req))
    for i in range(12):
        src, req, _ = syn.config_note(4, i)
        add(src, syn.request_args(req))
    return jobs


def _run(renderer, prep, lean):
    ctx = renderer.ctx
    ctx.set_option("lean", lean)
    try:
        out = renderer.run(prep, seed=11, keep_stems=True)
        ctx.check()
        res = {k: out[k].cpu().numpy().copy() for k in ("harm", "uv", "bre", "mix")}
        res["short"] = ctx.debug_fetch("mask_short")
        res["picks"] = ctx.debug_fetch("picks").reshape(-1, 2).copy()
    finally:
        ctx.set_option("lean", 1)
    return res


def test_lean_handoff_equals_written_mask():
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=256)
        jobs = _jobs()
        prep = r.prepare(jobs, note_ids=list(range(len(jobs))))
        prep["mask"].fill_(float("nan"))                     # lean must not read it
        a = _run(r, prep, 1)
        assert not np.isfinite(prep["mask"].cpu().numpy()).any(), "the lean path wrote the per-sample mask"
        b = _run(r, prep, 0)
        assert np.isfinite(prep["mask"].cpu().numpy()).all()
        so = np.asarray(prep["sample_off"], dtype=np.int64)
        for k in range(len(jobs)):
            n = int(so[k + 1] - so[k])
            base, ns = int(so[k]) // 4 + k, (n + 3) // 4
            assert np.array_equal(a["short"][base:base + ns], b["short"][base:base + ns]), k
        assert np.array_equal(a["picks"], b["picks"])
        for key in ("harm", "uv", "bre", "mix"):
            assert np.array_equal(a[key], b[key]), key
        assert np.isfinite(a["mix"]).all() and np.abs(a["uv"]).max() > 0 and np.abs(a["bre"]).max() > 0
        # both kinds of mask smoothing were exercised: flat stretches and transitions
        live = np.concatenate([a["short"][int(so[k]) // 4 + k: int(so[k]) // 4 + k + (int(so[k + 1] - so[k]) + 3) // 4] for k in range(len(jobs))])
        assert (live == 0).mean() > 0.02 and ((live > 1e-3) & (live < 0.999)).mean() > 0.01
    finally:
        ctx.close()


def test_lean_handoff_at_config5_geometry():
    """n_fft 2048 / hop 96 (the spectra pipeline: per-frame skip bits from the knots, k_irfft_ola1's gains): lean on / off."""
    from goofer_amd import sampler as S
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    geo = syn.config_geometry(5)
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=geo["hop"])
        jobs = []
        for k, i in enumerate([3, 11, 40, 77, 130]):
            src, req, _ = syn.config_note(5, i)
            if k % 2 == 0:
                src = syn.with_unvoiced_gaps(src, 0.35, 4000 + i)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
        prep = r.prepare(jobs, note_ids=list(range(len(jobs))))
        a = _run(r, prep, 1)
        b = _run(r, prep, 0)
        assert np.array_equal(a["picks"], b["picks"])
        for key in ("harm", "uv", "bre", "mix"):
            assert np.array_equal(a[key], b[key]), key
    finally:
        ctx.close()
