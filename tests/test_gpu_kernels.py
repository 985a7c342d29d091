"""GPU: every single-kernel C-ABI entry point against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

from conftest import golden, rel_rms, rms_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from goofer_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _off(c, lengths):
    return c.tensor(c.offsets(lengths))


def test_plan_tables_match_reference(ctx):
    g = golden("tables")
    for sr, n_fft, hop in ((44100, 1024, 256), (96000, 2048, 96), (48000, 512, 128)):
        ctx.plan(sr, n_fft, hop)
        t = f"{sr}_{n_fft}"
        assert np.max(np.abs(ctx.table(0) - g["win_" + t])) <= 6e-8          # libm cos vs numpy: <=1 ulp
        assert np.array_equal(ctx.table(1), g["freqs_" + t][:, 0])
        assert np.array_equal(ctx.table(2), g["boost_" + t][:, 0])
        assert np.array_equal(ctx.table(3), g["bright_harm_" + t][:, 0])
        assert np.array_equal(ctx.table(4), g["bright_breath_" + t][:, 0])


def test_rfft_and_irfft_against_reference(ctx):
    g = golden("stft_istft")
    for tag in g["cases"]:
        n_fft, hop = (int(v) for v in g[f"geo_{tag}"])
        ctx.plan(44100, n_fft, hop)
        x, S_ref, y_ref = g[f"x_{tag}"], g[f"S_{tag}"], g[f"y_{tag}"]
        n = len(x)
        T = 1 + n // hop
        assert S_ref.shape[1] == T
        S = ctx.rfft_frames(ctx.tensor(x), _off(ctx, [n]), _off(ctx, [T]), T)
        torch.cuda.synchronize()
        S = S.cpu().numpy().T
        assert rel_rms(S, S_ref) < 5e-7, (tag, rel_rms(S, S_ref))
        # inverse: device irfft+OLA of the REFERENCE spectrum vs the reference's istft
        St = ctx.tensor(np.ascontiguousarray(S_ref.T))
        n_dev = hop * (T - 1)
        if n_dev == 0:
            continue
        y = ctx.irfft_ola(St, _off(ctx, [n_dev]), _off(ctx, [T]), n_dev).cpu().numpy()
        assert n_dev <= len(y_ref) and not y_ref[n_dev:].any()
        assert rms_err(y, y_ref[:n_dev]) < 3e-7 * max(1.0, np.abs(y_ref).max()), tag


def test_rfft_ragged_batch_equals_per_note(ctx):
    from oracle import goofer_ref as R
    ctx.plan(44100, 1024, 256)
    rng = np.random.default_rng(5)
    lens = [4000, 1, 700, 256, 2, 12345, 511, 513]
    xs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    Ts = [1 + n // 256 for n in lens]
    S = ctx.rfft_frames(ctx.tensor(np.concatenate(xs)), _off(ctx, lens), _off(ctx, Ts), sum(Ts)).cpu().numpy()
    o = 0
    win = R.sqrt_hann(1024)
    for x, T in zip(xs, Ts):
        ref = R.stft(x, 1024, 256, win).T
        assert ref.shape[0] == T
        assert rel_rms(S[o:o + T], ref) < 5e-7
        o += T
    # round trip through the device inverse: istft(stft(x)) == x away from the zero-padded tail
    y = ctx.irfft_ola(ctx.tensor(S), _off(ctx, lens), _off(ctx, Ts), sum(lens)).cpu().numpy()
    o = 0
    for x, n in zip(xs, lens):
        valid = 256 * (n // 256)
        if valid:
            assert np.max(np.abs(y[o:o + valid] - x[:valid])) < 2e-5
        assert not y[o + valid:o + n].any()
        o += n


def test_pulse_train_against_reference(ctx):
    g = golden("pulse_train")
    ctx.plan(44100, 1024, 256)
    names = list(g["names"])
    f0s = [g["f0_" + k] for k in names]
    lens = [len(f) for f in f0s]
    out = ctx.pulse_train(ctx.tensor(np.concatenate(f0s)), _off(ctx, lens)).cpu().numpy()
    o = 0
    for k, n in zip(names, lens):
        ref = g["pulse_" + k]
        err = np.max(np.abs(out[o:o + n] - ref))
        assert err < 5e-6, (k, err)     # a one-sample onset slip would be O(1)
        o += n
    ctx.plan(96000, 2048, 96)
    f = g["f0_sr96"]
    out = ctx.pulse_train(ctx.tensor(f), _off(ctx, [len(f)])).cpu().numpy()
    assert np.max(np.abs(out - g["pulse_sr96"])) < 5e-6


def test_pulse_train_other_lf_models_against_reference(ctx):
    """goofer_pulse_model: Ra / Rg / Rk other than gf.synthesize's constants (GOOFER.py:474, 508-519) against pulse trains the
    reference wrote — the shape table (T0 <= 2048) and the on-the-fly path (f0 18.3 Hz: T0 = 2410) — through the handle and
    through core.pulse_train_numba, which puts the constants back; a new plan starts from the constants as well."""
    from goofer_amd import core
    g = golden("pulse_train_lf")
    sr = int(g["sr"])
    ctx.plan(sr, 1024, 256)
    names = [str(k) for k in g["names"]]
    f0s = [g["f0_" + k] for k in names]
    lens = [len(f) for f in f0s]
    d_f0, off = ctx.tensor(np.concatenate(f0s)), _off(ctx, lens)
    try:
        for m, (Ra, Rg, Rk) in enumerate(g["models"]):
            ctx.pulse_model(Ra, Rg, Rk)
            out = ctx.pulse_train(d_f0, off).cpu().numpy()
            o = 0
            for k, n in zip(names, lens):
                err = np.max(np.abs(out[o:o + n] - g["pulse_%s_%d" % (k, m)]))
                assert err < 5e-6, (k, m, err)
                o += n
    finally:
        ctx.pulse_model()
    default = ctx.pulse_train(d_f0, off).cpu().numpy()
    assert np.max(np.abs(default[:lens[0]] - g["pulse_glide_3"])) < 5e-6          # model 3 is the constants
    Ra, Rg, Rk = g["models"][1]
    got = core.pulse_train_numba(f0s[0], sr, Ra=Ra, Rg=Rg, Rk=Rk, ctx=ctx)
    assert np.max(np.abs(got - g["pulse_glide_1"])) < 5e-6
    assert ctx.lf == (0.02, 1.7, 0.8)
    assert np.array_equal(ctx.pulse_train(d_f0, off).cpu().numpy(), default)
    ctx.pulse_model(Ra, Rg, Rk)
    ctx.plan(sr, 2048, 512)
    ctx.plan(sr, 1024, 256)
    assert np.array_equal(ctx.pulse_train(d_f0, off).cpu().numpy(), default)
    with pytest.raises(RuntimeError):
        ctx.pulse_model(float("nan"), 1.7, 0.8)


def test_pulse_train_many_random_notes_vs_oracle(ctx):
    from oracle import goofer_ref as R
    ctx.plan(44100, 1024, 256)
    rng = np.random.default_rng(17)
    f0s = []
    for i in range(70):   # > 64: more than one wave of note-lanes
        n = int(rng.integers(50, 9000))
        t = np.arange(n) / 44100
        base = rng.uniform(60, 900)
        f = base * 2 ** (rng.uniform(-0.5, 0.5) * np.sin(2 * np.pi * rng.uniform(1, 9) * t))
        f[rng.uniform(size=n) < 0.02] = 0
        if i % 7 == 0:
            f[:] = [440.0, 441.0, 220.5, 110.25, 882.0][i % 5]   # knife-edge rationals
        f0s.append(f.astype(np.float32))
    lens = [len(f) for f in f0s]
    out = ctx.pulse_train(ctx.tensor(np.concatenate(f0s)), _off(ctx, lens)).cpu().numpy()
    o = 0
    for f, n in zip(f0s, lens):
        ref = R.pulse_train(f, 44100)
        assert np.max(np.abs(out[o:o + n] - ref)) < 2e-6
        o += n


def _onset_lists(c, lens):
    cnt = c.debug_fetch("onset_cnt")
    idx = c.debug_fetch("onset_idx")
    off = np.concatenate([[0], np.cumsum(lens)])
    return [idx[off[k] // 2 + 16 * k: off[k] // 2 + 16 * k + cnt[k]].copy() for k in range(len(lens))]


def test_pulse_onsets_parallel_scan_equals_sequential_walk(ctx):
    """Option pulse_scan: 1 (default) takes the onsets from the parallel fp64 phase scan wherever its rounding band cannot move
    floor(phase) and walks the other notes sequentially inside the same kernel; 0 runs the sequential walk kernel on every
    note; 2 makes the scan kernel walk every note.  The three give the same onset samples (array_equal) and the same pulse bits on long ordinary notes, exact-rational
    f0 (every crossing lands on an integer phase: all of those notes must take the walk), negative and oversized increments,
    all-zero notes, unvoiced heads, and lengths around the 512-sample round."""
    from oracle import goofer_ref as R
    ctx.plan(44100, 1024, 256)
    rng = np.random.default_rng(41)
    f0s, must_walk = [], []
    for i in range(44):
        n = int(rng.integers(20000, 70000)) if i < 30 else [1, 2, 511, 512, 513, 1023, 1024, 1025, 7, 64, 4096, 4097, 3000, 5000][i - 30]
        t = np.arange(n) / 44100
        f = rng.uniform(60, 900) * 2 ** (rng.uniform(-0.5, 0.5) * np.sin(2 * np.pi * rng.uniform(0.2, 9) * t))
        f[:int(0.08 * n)] = 0                                  # unvoiced head: the phase stays exactly 0
        f[rng.uniform(size=n) < 0.01] = 0
        walk = False
        if i % 6 == 0:
            f[:] = [441.0, 220.5, 882.0, 110.25, 440.0][(i // 6) % 5]
            walk = n >= 401                                    # at least one onset on an integer phase (440 Hz: 2205 samples)
            if (i // 6) % 5 == 4:
                walk = n >= 2205
        if i % 6 == 1:
            f[n // 3:n // 3 + 20] = -500.0
            walk = True
        if i % 6 == 2 and n > 100:
            f[int(rng.integers(50, n))] = 90000.0
        if i == 9:
            f[:] = 0
        f0s.append(f.astype(np.float32))
        must_walk.append(walk)
    lens = [len(f) for f in f0s]
    d_f0, off = ctx.tensor(np.concatenate(f0s)), _off(ctx, lens)
    res = {}
    try:
        for mode in (0, 1, 2):
            ctx.set_option("pulse_scan", mode)
            s0, w0 = ctx.counter("pulse_scanned_notes"), ctx.counter("pulse_fallback_notes")
            pulse = ctx.pulse_train(d_f0, off).cpu().numpy()
            res[mode] = (pulse, _onset_lists(ctx, lens), ctx.counter("pulse_scanned_notes") - s0, ctx.counter("pulse_fallback_notes") - w0)
    finally:
        ctx.set_option("pulse_scan", 1)
    assert res[0][2:] == (0, 0)
    assert res[2][2:] == (len(lens), len(lens))
    assert res[1][2] == len(lens)
    assert sum(must_walk) <= res[1][3] <= sum(must_walk) + 3, (res[1][3], sum(must_walk))   # both branches taken in one batch
    for mode in (1, 2):
        assert np.array_equal(res[mode][0], res[0][0]), mode
        for k, (a, b) in enumerate(zip(res[mode][1], res[0][1])):
            assert np.array_equal(a, b), (mode, k)
    assert sum(len(a) for a in res[0][1]) > 5000
    o = 0
    for k, (f, n) in enumerate(zip(f0s, lens)):                 # ... and they are the reference's onsets
        if k % 3 == 0:
            assert np.max(np.abs(res[1][0][o:o + n] - R.pulse_train(f, 44100))) < 4e-6, k
        o += n


def test_pulse_train_negative_and_oversized_increments(ctx):
    """R_i = max(R_{i-1}, floor(phase_i)): a falling phase records nothing until it passes the old maximum again, and an
    increment above 1 records several onsets on one sample (the reference's `while`, GOOFER.py:492)."""
    from oracle import goofer_ref as R
    ctx.plan(44100, 1024, 256)
    rng = np.random.default_rng(23)
    f0s = []
    for i in range(9):
        n = int(rng.integers(700, 4000))
        f = rng.uniform(150, 700) * np.ones(n)
        for _ in range(4):
            a = int(rng.integers(0, n - 60))
            f[a:a + int(rng.integers(1, 60))] = -rng.uniform(100, 3000)       # phase runs backwards
        for _ in range(3):
            f[int(rng.integers(0, n))] = rng.uniform(50000, 140000)          # 1-3 onsets on one sample
        if i == 0:
            f[:] = -200.0                                                    # never reaches 1
        f0s.append(f.astype(np.float32))
    lens = [len(f) for f in f0s]
    out = ctx.pulse_train(ctx.tensor(np.concatenate(f0s)), _off(ctx, lens)).cpu().numpy()
    o = 0
    for f, n in zip(f0s, lens):
        ref = R.pulse_train(f, 44100)
        assert np.max(np.abs(out[o:o + n] - ref)) < 4e-6
        o += n
    assert not out[:lens[0]].any()


def test_gauss_bins(ctx):
    from goofer_amd.core import gaussian_taps
    g = golden("gauss")
    ctx.plan(44100, 1024, 256)
    env = g["env"]
    rows = ctx.rows_from(env.T)
    for s in (0.5, 1.75, 2.0, 3.4, 7.0):
        out = ctx.gauss_bins(rows, gaussian_taps(s)).cpu().numpy().T
        ref = g["ax0_s%g" % s]
        assert np.max(np.abs(out - ref.astype(np.float32))) <= 1.2e-7 * np.abs(ref).max()
    small = g["small"]      # radius 8 on 9 bins: multiple reflections
    out = ctx.gauss_bins(ctx.rows_from(small.T), gaussian_taps(2.0)).cpu().numpy().T
    assert np.max(np.abs(out - g["small_s2"].astype(np.float32))) <= 1.2e-7 * np.abs(small).max()


def test_warp_bins_against_reference(ctx):
    g = golden("warps")
    ctx.plan(44100, 1024, 256)
    env = g["env"]
    rows = ctx.rows_from(env.T)
    for r in (0.75, 1.25, 1.5, 0.5):
        out = ctx.warp_bins(rows, ratio=r).cpu().numpy().T
        ref = g["shift_%g" % r]
        assert np.max(np.abs(out - ref)) <= 2e-7 * np.abs(ref).max(), r
    F = g["formants"]
    dF = ctx.tensor(np.ascontiguousarray(F.T))
    for i in range(3):
        out = ctx.warp_bins(rows, formants=dF, f_shift=g[f"ratios_{i}"]).cpu().numpy().T
        ref = g[f"warp_{i}"]
        assert np.max(np.abs(out - ref)) <= 2e-7 * np.abs(ref).max(), i


def test_warp_bins_crossing_anchors_follow_numpy_interp(ctx):
    """Unsorted anchors: np.interp's answer depends on its guess chain; the kernel reproduces it."""
    from oracle import goofer_ref as R
    from goofer_amd import synthetic as syn
    ctx.plan(44100, 1024, 256)
    src = syn.make_source(123, seconds=0.3)
    env = R.decode_env_from_knots(src["env_pack"])
    T = env.shape[1]
    rng = np.random.default_rng(9)
    F = np.stack([src["formants"][i][:T] for i in (1, 2, 3, 4)], 0)
    F[0] = rng.uniform(600, 1100, T)
    F[1] = rng.uniform(700, 1400, T)
    F[2, ::5] = 0.0
    F[3, ::7] = rng.uniform(100, 21000, len(F[3, ::7]))
    rows = ctx.rows_from(env.T)
    dF = ctx.tensor(np.ascontiguousarray(F.T))
    for ratios in ([1.3, 0.8, 1.1, 0.9], [1.5, 0.6, 0.7, 1.4], [0.7, 1.3, 1.3, 0.6]):
        sh = F * np.asarray(ratios)[:, None]
        crossing = np.mean(np.any(np.diff(np.where((F > 50) & (F < 22050) & (sh > 50), sh, np.nan), axis=0) < 0, axis=0))
        ref = R.warp_env_by_formants(env, F, sh, 44100)
        out = ctx.warp_bins(rows, formants=dF, f_shift=ratios).cpu().numpy().T
        assert crossing > 0.1
        assert np.max(np.abs(out - ref)) <= 2e-7 * np.abs(ref).max(), ratios
        both = ctx.warp_bins(rows, formants=dF, f_shift=ratios, ratio=1.25).cpu().numpy().T
        # two chained fp32 interpolation stages (round 5: the per-bin lerps run in fp32): 4 ulp of the row maximum
        assert np.max(np.abs(both - R.shift_formants(ref, 1.25, 44100))) <= 5e-7 * np.abs(ref).max()


def test_knot_decode(ctx):
    from goofer_amd import core
    g = golden("knots")
    pack = {"mode": "knots", "knot_vals_log": g["knot_vals_log"], "hz_knots": g["hz_knots"],
            "n_bins": 513, "n_fft": 1024, "sr": 44100}
    out = core.decode_env_from_knots(pack, ctx=ctx)
    assert out.dtype == np.float32 and out.shape == g["decoded"].shape
    np.testing.assert_allclose(out, g["decoded"], rtol=1e-6)


def test_analysis_envelope_and_knot_encode(ctx):
    """|stft| + 1e-8 -> sigma 2 blur -> sigma 0.5 blur -> K search -> fp16 knots, against the reference."""
    from goofer_amd import core
    g = golden("knots")
    env, pack = core.envelope_features(g["an_x"], 44100, ctx=ctx)
    assert env.dtype == np.float64 and env.shape == g["an_env"].shape
    np.testing.assert_allclose(env, g["an_env"], rtol=2e-6, atol=1e-9)
    assert np.array_equal(pack["hz_knots"], g["an_hz_knots"])                  # same K chosen
    a, b = pack["knot_vals_log"].astype(np.float32), g["an_knot_vals_log"].astype(np.float32)
    assert pack["knot_vals_log"].dtype == np.float16 and a.shape == b.shape
    assert np.mean(a == b) > 0.995 and np.max(np.abs(a - b)) <= 0.008          # <= 1 fp16 ulp on rare rounding ties
    # a smooth envelope picks a small K, exactly like the reference
    p2 = core.compress_env_to_knots(g["sm_env"], 44100, 1024, ctx=ctx)
    assert np.array_equal(p2["hz_knots"], g["sm_hz_knots"])
    a, b = p2["knot_vals_log"].astype(np.float32), g["sm_knot_vals_log"].astype(np.float32)
    assert np.mean(a == b) > 0.995 and np.max(np.abs(a - b)) <= 0.008
    with pytest.raises(NotImplementedError):
        core.extract_features(g["an_x"], 44100, ctx=ctx)
    out = core.extract_features(g["an_x"], 44100, ctx=ctx,
                                pitch_tracker=lambda y, sr, hop, T: (np.full(T, 220.0), {k: [500.0 * k] * T for k in range(1, 6)}))
    assert out[1].shape == (len(g["an_x"]),) and out[2].min() == 1.0 and out[4]["mode"] == "knots"


def test_onepole_cascade_vs_reference(ctx):
    """K10: time-varying one-pole cascades as an affine scan, against the reference's sequential fp32 loops."""
    g = golden("post_chain")
    ctx.plan(44100, 1024, 256)
    x, f0 = ctx.tensor(g["x"]), ctx.tensor(g["f0"])
    for i in range(4):
        cf, order, hp = g[f"dyn_args_{i}"]
        y = ctx.onepole_cascade(x, f0, float(cf), int(order), "highpass" if hp else "lowpass").cpu().numpy()
        ref = g[f"dyn_{i}"]
        assert y.dtype == np.float32 and y.shape == ref.shape
        scale = max(1e-6, float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))
        assert rms_err(y, ref) / scale < 2e-6, (i, rms_err(y, ref) / scale)
    # several notes in one launch, lengths that are not tile multiples, in/out of the oracle's restatement
    from oracle import sampler_ref as SR
    lens = [1, 7, 2049, 2943]
    xs, fs = g["x"][:sum(lens)], g["f0"][:sum(lens)]
    y = ctx.onepole_cascade(ctx.tensor(xs), ctx.tensor(fs), 1.0, 12, "highpass", f0_mode=1, lengths=lens).cpu().numpy()
    o = 0
    for n in lens:
        ref = SR.dynamic_filter(xs[o:o + n], np.maximum(fs[o:o + n], 120.0), 44100, 1.0, order=6, btype="highpass")
        ref = SR.dynamic_filter(ref, np.maximum(fs[o:o + n], 120.0), 44100, 1.0, order=6, btype="highpass")
        assert np.max(np.abs(y[o:o + n] - ref)) < 1e-6 * max(1.0, float(np.max(np.abs(ref)))), n
        o += n


def test_stretch_rows_vs_oracle(ctx):
    """goofer_stretch_rows = gf.stretch_feature (np.interp on normalised coordinates) for 1-D arrays and row matrices,
    growing and shrinking, including the single-knot and single-output corner cases."""
    from oracle import goofer_ref as R
    ctx.plan(44100, 1024, 256)
    rng = np.random.default_rng(21)
    for n_in, factor in ((1000, 1.37), (1000, 0.41), (70001, 1.003), (5, 3.2), (1, 4.0), (3, 0.34)):
        x = rng.standard_normal(n_in).astype(np.float32)
        ref = R.stretch_feature(x, factor)
        got = ctx.stretch_rows(ctx.tensor(x), len(ref)).cpu().numpy()
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref.astype(np.float32))) <= 1e-6 * max(1.0, float(np.max(np.abs(ref)))) if len(ref) else True
    for T, factor in ((37, 1.6), (200, 0.55), (2, 2.5)):
        M = rng.random((ctx.n_bins, T)).astype(np.float32)
        ref = R.stretch_feature(M, factor)                          # [bins, T']
        got = ctx.stretch_rows(ctx.rows_from(M.T), ref.shape[1]).cpu().numpy().T
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref.astype(np.float32))) <= 1e-6


def test_module_level_helpers_vs_oracle(ctx):
    """gf.gaussian_filter1d / gaussian_filter / stretch_feature / create_volume_jitter as the reference exposes them
    (SillySampler.py and SillyEditor.py call them on the module)."""
    from goofer_amd import core
    from oracle import goofer_ref as R
    rng = np.random.default_rng(31)
    x = rng.standard_normal(5000)
    for sigma in (0.5, 2.0, 20.0, 73.5):
        a, b = core.gaussian_filter1d(x, sigma, ctx=ctx), R.gauss1d(x, sigma)
        assert a.dtype == np.float64 and a.shape == b.shape and np.max(np.abs(a - b)) < 1e-13
    short = rng.standard_normal(7)                                   # radius larger than the array: multiple reflections
    assert np.max(np.abs(core.gaussian_filter1d(short, 3.0, ctx=ctx) - R.gauss1d(short, 3.0))) < 1e-13
    M = rng.random((33, 40))
    for axis in (0, 1):
        assert np.max(np.abs(core.gaussian_filter1d(M, 1.75, axis=axis, ctx=ctx) - R.gauss1d(M, 1.75, axis=axis))) < 1e-13
    assert np.max(np.abs(core.gaussian_filter(M, (0.5, 0), ctx=ctx) - R.gauss2d(M, (0.5, 0)))) < 1e-13
    Z = M[:, :20] + 1j * M[:, 20:]
    assert np.max(np.abs(core.gaussian_filter1d(Z, 0.5, axis=0, ctx=ctx) - R.gauss1d(Z, 0.5, axis=0))) < 1e-13
    assert core.gaussian_filter1d(x, 0.0, ctx=ctx) is not x and np.array_equal(core.gaussian_filter1d(x, 0.0, ctx=ctx), x)
    f = rng.random(300).astype(np.float32)
    assert np.max(np.abs(core.stretch_feature(f, 1.7, ctx=ctx) - R.stretch_feature(f, 1.7))) < 1e-6
    E = rng.random((513, 21)).astype(np.float32)
    assert np.max(np.abs(core.stretch_feature(E, 0.6, ctx=ctx) - R.stretch_feature(E, 0.6))) < 1e-6
    np.random.seed(5)
    a = core.create_volume_jitter(4000, 44100, speed=150, strength=0.8, ctx=ctx)
    np.random.seed(5)
    b = R.volume_jitter_curve(4000, 44100, speed=150.0, strength=0.8)
    assert np.max(np.abs(a - b)) < 1e-12
    assert np.array_equal(core.create_volume_jitter(4000, 44100, speed=150.0, strength=0.15, vibrato=True, ctx=ctx),
                          R.volume_jitter_curve(4000, 44100, speed=150.0, strength=0.15, vibrato=True))
    assert abs(core.rms(x) - R.rms(x)) < 1e-15


def test_smooth_mask_ds_vs_reference(ctx):
    """a7 on its own (GOOFER.py:556-569): the reference's smooth_mask_ds outputs, both interpolant forms (the search-loop form
    of the separate stem-gain kernel and the walkers' flat-knot shortcut form) bit-identical, ragged batches vs the oracle."""
    from oracle import goofer_ref as R
    g = golden("mask_interp")
    ctx.plan(44100, 1024, 256)
    m = ctx.tensor(np.asarray(g["mask"], dtype=np.float32))
    for sigma, key in ((100.0, "smooth_100"), (1.0, "smooth_1")):
        a = ctx.smooth_mask_ds(m, sigma=sigma).cpu().numpy()
        b = ctx.smooth_mask_ds(m, sigma=sigma, fast_interp=True).cpu().numpy()
        assert a.dtype == np.float32 and a.shape == g[key].shape
        assert np.max(np.abs(a - g[key])) < 1e-7, sigma
        assert np.array_equal(a, b), sigma
    rng = np.random.default_rng(17)
    lens = [1, 2, 3, 4, 5, 7, 8, 9, 255, 1024, 4097, 20000]
    masks = []
    for n in lens:
        x = (rng.random(n) > 0.4).astype(np.float32)
        x[n // 3:n // 2] = 1.0
        masks.append(x)
    cat = ctx.tensor(np.concatenate(masks))
    for sigma in (100.0, 37.0):
        a = ctx.smooth_mask_ds(cat, lengths=lens, sigma=sigma).cpu().numpy()
        b = ctx.smooth_mask_ds(cat, lengths=lens, sigma=sigma, fast_interp=True).cpu().numpy()
        assert np.array_equal(a, b)
        o = 0
        for n, x in zip(lens, masks):
            want = R.smooth_mask(x, sigma, 4)
            assert np.max(np.abs(a[o:o + n] - want)) < 1e-7, (n, sigma)
            o += n


@pytest.mark.parametrize("n_fft,hop", [(768, 192), (1536, 384), (768, 100), (1536, 96)])
def test_transform_sizes_with_a_factor_three(ctx, n_fft, hop):
    """gf.stft / istft / synthesize take any n_fft (GOOFER.py:355, 392, 972).  Beside the powers of two the device path has
    768 and 1536 (M = 384 / 768 = 64 lanes x a 6- / 12-point first pass): spectra, inverse + overlap-add and the whole
    synthesis against the oracle (numpy pocketfft at those sizes)."""
    from goofer_amd import core
    from oracle import goofer_ref as R
    sr = 44100
    rng = np.random.default_rng(n_fft + hop)
    lens = [5000, 3, n_fft - 1, 2 * n_fft + 17]
    win = R.sqrt_hann(n_fft)
    ctx.plan(sr, n_fft, hop)
    xs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    Ts = [1 + n // hop for n in lens]
    S = ctx.rfft_frames(ctx.tensor(np.concatenate(xs)), _off(ctx, lens), _off(ctx, Ts), sum(Ts)).cpu().numpy()
    o = 0
    for x, T in zip(xs, Ts):
        ref = R.stft(x, n_fft, hop, win).T
        assert ref.shape == (T, n_fft // 2 + 1)
        assert rel_rms(S[o:o + T], ref) < 6e-7, (n_fft, hop, len(x))
        o += T
        if T < 2:
            continue                                          # a one-frame spectrum inverts to zero samples
        y = core.istft(ref.T, hop_length=hop, sr=sr, ctx=ctx)
        y_ref = R.istft(ref.T, hop, win)
        assert y.shape == y_ref.shape and rms_err(y, y_ref) < 4e-7 * max(1.0, float(np.abs(y_ref).max())), (n_fft, hop, len(x))
    # gf.synthesize at this geometry, injected phases
    n = 6000
    B, T = n_fft // 2 + 1, 1 + n // hop
    f = np.arange(B) * (sr / n_fft)
    env = (np.exp(-f / 3000.0)[:, None] * (1.0 + 0.2 * np.sin(np.arange(T) / 5.0))[None, :]).astype(np.float32)
    f0 = (200.0 + 20.0 * np.sin(np.arange(n) / 900.0)).astype(np.float32)
    mask = np.ones(n, dtype=np.float32)
    mask[:700] = 0.0
    f0 = f0 * mask
    phi = rng.uniform(0.0, 2.0 * np.pi, size=(B, T)).astype(np.float32)
    forms = {k: np.full(T, 600.0 * k) for k in (1, 2, 3, 4)}
    kw = dict(n_fft=n_fft, hop_length=hop, formants=forms, F1_shift=1.2, formant_shift=0.9)
    ref = R.synthesize(env, f0.astype(np.float64), mask, np.empty(n, bool), sr, phi=phi, **kw)
    got = core.synthesize(env, f0, mask, np.empty(n, bool), sr, phi=phi, ctx=ctx, **kw)
    for a, b, name in zip(got, ref, ("rec", "harm", "uv", "bre")):
        assert rms_err(a, b) < 2e-5, (n_fft, hop, name, rms_err(a, b))
    ctx.plan(44100, 1024, 256)


@pytest.mark.parametrize("n_fft,hop", [(1000, 250), (600, 150), (882, 147), (320, 80), (64, 16), (130, 40), (1022, 300), (1200, 300), (2000, 500), (1026, 256)])
def test_transform_sizes_without_a_radix_plan(ctx, n_fft, hop):
    """Any even n_fft up to 2048 (GOOFER.py:355, 392, 972 take whatever the caller passes): Bluestein's chirp-z form of the
    n_fft / 2-point transform through power-of-two transforms of 256 .. 2048 points (k_rfft_bluestein / k_irfft_bluestein), incl. odd half sizes
    (882 -> 441, 130 -> 65) and bin counts that are not 64 k + 1: spectra, inverse + overlap-add and the whole synthesis against
    the oracle (numpy pocketfft at those sizes).  Three transforms' worth of fp32 rounding instead of one: 2e-6 on the spectra."""
    from goofer_amd import core
    from oracle import goofer_ref as R
    sr = 44100
    rng = np.random.default_rng(n_fft + hop)
    lens = [5000, 3, n_fft - 1, 2 * n_fft + 17]
    win = R.sqrt_hann(n_fft)
    ctx.plan(sr, n_fft, hop)
    xs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    Ts = [1 + n // hop for n in lens]
    S = ctx.rfft_frames(ctx.tensor(np.concatenate(xs)), _off(ctx, lens), _off(ctx, Ts), sum(Ts)).cpu().numpy()
    o = 0
    for x, T in zip(xs, Ts):
        ref = R.stft(x, n_fft, hop, win).T
        assert ref.shape == (T, n_fft // 2 + 1)
        assert rel_rms(S[o:o + T], ref) < 2e-6, (n_fft, hop, len(x), rel_rms(S[o:o + T], ref))
        o += T
        if T < 2:
            continue
        y = core.istft(ref.T, hop_length=hop, sr=sr, ctx=ctx)
        y_ref = R.istft(ref.T, hop, win)
        assert y.shape == y_ref.shape and rms_err(y, y_ref) < 1.5e-6 * max(1.0, float(np.abs(y_ref).max())), (n_fft, hop, len(x))
    n = 6000
    B, T = n_fft // 2 + 1, 1 + n // hop
    f = np.arange(B) * (sr / n_fft)
    env = (np.exp(-f / 3000.0)[:, None] * (1.0 + 0.2 * np.sin(np.arange(T) / 5.0))[None, :]).astype(np.float32)
    f0 = (200.0 + 20.0 * np.sin(np.arange(n) / 900.0)).astype(np.float32)
    mask = np.ones(n, dtype=np.float32)
    mask[:700] = 0.0
    f0 = f0 * mask
    phi = rng.uniform(0.0, 2.0 * np.pi, size=(B, T)).astype(np.float32)
    forms = {k: np.full(T, 600.0 * k) for k in (1, 2, 3, 4)}
    kw = dict(n_fft=n_fft, hop_length=hop, formants=forms, F1_shift=1.2, formant_shift=0.9)
    ref = R.synthesize(env, f0.astype(np.float64), mask, np.empty(n, bool), sr, phi=phi, **kw)
    got = core.synthesize(env, f0, mask, np.empty(n, bool), sr, phi=phi, ctx=ctx, **kw)
    for a, b, name in zip(got, ref, ("rec", "harm", "uv", "bre")):
        assert rms_err(a, b) < 2e-5, (n_fft, hop, name, rms_err(a, b))
    with pytest.raises(Exception):
        ctx.plan(sr, 1001, 250)                               # odd sizes stay refused
    with pytest.raises(Exception):
        ctx.plan(sr, 2050, 512)
    ctx.plan(44100, 1024, 256)
