"""n_fft 2048 (BASELINE config 5 is 2048 / 96 at 96 kHz), goofer_amd/csrc/stems_ring.hip, against the one-kernel-per-reference-step
pipeline (k_rfft_frames -> k_harm_shape, k_noise_spectra -> k_irfft_ola1 with three spectra in HBM), which tests/test_gpu_synth.py
and tests/test_gpu_sampler.py pin to the oracle:

  k_rfft_shape   the framewise rFFT and the harmonic shaping as one kernel (option "rfft_shape" 1)
  k_stem_ring    the ring walkers: no spectrum in HBM at all (option "ring_walkers" 1)

Both are off by default: on the MI355X they are slower than the kernels they fuse, and stay as measured
alternatives with their parity tests.

Both evaluate the same expressions in the same order as the kernels they replace, so every stem and the mix must be the same bits."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

KEYS = ("harm", "uv", "bre", "rec", "mix")


@pytest.fixture(scope="module")
def ctx():
    from goofer_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _batch(ctx, hop, seed, warp):
    from goofer_amd.device import default_params
    rng = np.random.default_rng(seed)
    nb = 1025
    lens = [1, hop - 1, hop, 2048 + 3, 5 * 2048 + 17, 9 * hop, 20000, 2047, 4096, 31000]
    envs, f0s, masks, env_len = [], [], [], []
    for n in lens:
        T = 1 + n // hop
        rows = max(1, T + int(rng.integers(0, 3)) - 1)                # fewer / equal / more envelope rows than frames
        envs.append((1.0 + rng.random((rows, nb))).astype(np.float32))
        env_len.append(rows)
        m = (rng.random(n) > 0.3).astype(np.float32)
        m[n // 3:n // 2] = 1.0                                        # flat and transition stretches of the smoothed mask
        m[(2 * n) // 3:(5 * n) // 6] = 0.0
        masks.append(m)
        f0s.append((200.0 + 50.0 * rng.random(n)).astype(np.float32) * m)
    par = default_params(len(lens))
    par["seed"][:, 0] = rng.integers(0, 2 ** 32, len(lens), dtype=np.uint64).astype(np.uint32)
    par["apply_brightness"][3] = 0
    par["cut_below_f0"][4] = 0
    formants = None
    if warp:
        par["formant_shift"][1::3] = 1.12                             # uniform warp ('g')
        par["f_shift"][2::3] = (1.1, 0.9, 1.05, 0.95)                 # formant-anchored warp ('fa'..'fd')
        par["f_shift"][6] = (1.9, 0.4, 1.0, 1.0)                      # crossing anchors
        R = int(np.sum(env_len))
        formants = np.sort(rng.uniform(300.0, 4500.0, size=(R, 4)), axis=1)
        formants[rng.random(R) < 0.1] = np.nan                        # frames without a formant estimate
        formants = ctx.tensor(formants.astype(np.float64))
    args = (ctx.rows_from(np.concatenate(envs)), env_len, ctx.tensor(np.concatenate(f0s)), ctx.tensor(np.concatenate(masks)), lens, par)
    return args, formants, sum(1 + n // hop for n in lens)


def _run(ctx, args, formants, phi, seed, **kw):
    out = ctx.synth_batch(*args, formants=formants, phi=phi, seed=seed, **kw)
    torch.cuda.synchronize()
    ctx.check()
    return {k: out[k].cpu().numpy().copy() for k in KEYS}


@pytest.mark.parametrize("hop", [96, 512, 1024, 200])
@pytest.mark.parametrize("warp", [False, True])
def test_ring_walkers_equal_the_spectra_pipeline(ctx, hop, warp):
    ctx.plan(96000, 2048, hop)
    try:
        args, formants, F = _batch(ctx, hop, 2048 + hop, warp)
        for inject in (False, True):
            phi = None
            if inject:
                ph = np.random.default_rng(3).uniform(0.0, 2.0 * np.pi, size=(F, 1025)).astype(np.float32)
                phi = ctx.rows_from(ph)
            got = {}
            for ring in (1, 0):
                ctx.set_option("ring_walkers", ring)
                got[ring] = _run(ctx, args, formants, phi, 17)
            for k in KEYS:
                assert np.array_equal(got[1][k], got[0][k]), (k, inject)
                assert np.all(np.isfinite(got[1][k])), k
            assert np.abs(got[1]["harm"]).max() > 0 and np.abs(got[1]["uv"]).max() > 0 and np.abs(got[1]["bre"]).max() > 0
    finally:
        ctx.set_option("ring_walkers", 0)
        ctx.plan(44100, 1024, 256)


@pytest.mark.parametrize("hop", [96, 512, 200])
@pytest.mark.parametrize("warp", [False, True])
def test_fused_rfft_shaping_equals_the_two_kernels(ctx, hop, warp):
    ctx.plan(96000, 2048, hop)
    try:
        args, formants, _ = _batch(ctx, hop, 4096 + hop, warp)
        got = {}
        for fused in (1, 0):
            ctx.set_option("rfft_shape", fused)
            got[fused] = _run(ctx, args, formants, None, 23)
        for k in KEYS:
            assert np.array_equal(got[1][k], got[0][k]), k
            assert np.all(np.isfinite(got[1][k])), k
        assert np.abs(got[1]["harm"]).max() > 0
        ctx.set_option("overlap", 0)                                  # everything on one stream (the frame picks come from the map kernel either way)
        try:
            one = _run(ctx, args, formants, None, 23)
        finally:
            ctx.set_option("overlap", 1)
        for k in KEYS:
            assert np.array_equal(got[1][k], one[k]), k
    finally:
        ctx.set_option("rfft_shape", 0)
        ctx.plan(44100, 1024, 256)


def test_ring_walkers_without_the_side_stream_and_without_skipping(ctx):
    """overlap 0: all three stems in the launches of one stream; skip_zero 0: every noise frame transformed.  Same bits."""
    ctx.plan(96000, 2048, 96)
    try:
        args, formants, _ = _batch(ctx, 96, 77, True)
        ctx.set_option("ring_walkers", 1)
        base = _run(ctx, args, formants, None, 5)
        for opt in ("overlap", "skip_zero"):
            ctx.set_option(opt, 0)
            try:
                other = _run(ctx, args, formants, None, 5)
            finally:
                ctx.set_option(opt, 1)
            for k in KEYS:
                assert np.array_equal(base[k], other[k]), (opt, k)
        ctx.set_option("ring_walkers", 0)
        ref = _run(ctx, args, formants, None, 5)
        for k in KEYS:
            assert np.array_equal(base[k], ref[k]), k
    finally:
        ctx.set_option("ring_walkers", 0)
        ctx.plan(44100, 1024, 256)


def test_ring_walkers_on_config5_notes_of_the_sampler():
    """goofer_render_batch on BASELINE config 5 notes (96 kHz, br / es flags, sources with unvoiced gaps): walkers on / off."""
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    geo = syn.config_geometry(5)
    c = Context(0)
    try:
        r = Renderer(c, hop=geo["hop"])
        jobs = []
        for k, i in enumerate([3, 11, 40, 77, 130, 500, 901]):
            src, req, _ = syn.config_note(5, i)
            if k % 2 == 0:
                src = syn.with_unvoiced_gaps(src, 0.35, 4000 + i)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
        prep = r.prepare(jobs, note_ids=list(range(len(jobs))))
        outs = {}
        for ring in (1, 0):
            c.set_option("ring_walkers", ring)
            o = r.run(prep, seed=5, keep_stems=True)
            c.check()
            outs[ring] = {k: o[k].cpu().numpy().copy() for k in ("harm", "uv", "bre", "mix")}
        for k in ("harm", "uv", "bre", "mix"):
            assert np.array_equal(outs[1][k], outs[0][k]), k
        assert np.isfinite(outs[1]["mix"]).all() and np.abs(outs[1]["mix"]).max() > 0
        c.set_option("rfft_shape", 1)                                 # ... and the spectra path with transform and shaping fused
        o = r.run(prep, seed=5, keep_stems=True)
        c.check()
        for k in ("harm", "uv", "bre", "mix"):
            assert np.array_equal(outs[0][k], o[k].cpu().numpy()), k
    finally:
        c.close()
