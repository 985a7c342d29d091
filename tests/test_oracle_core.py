"""CPU: the oracle's DSP-core restatement against the reference's golden vectors."""
import numpy as np
import pytest

from conftest import golden, rel_rms, rms_err
from oracle import goofer_ref as R


def test_tables():
    g = golden("tables")
    for sr, n_fft in ((44100, 1024), (96000, 2048), (48000, 512)):
        t = f"{sr}_{n_fft}"
        assert np.array_equal(R.sqrt_hann(n_fft), g["win_" + t])
        assert np.array_equal(R.bin_freqs(sr, n_fft), g["freqs_" + t][:, 0])
        assert np.array_equal(R.boost_curve(n_fft), g["boost_" + t][:, 0])
        h, b = R.brightness_curves(sr, n_fft)
        assert np.array_equal(h, g["bright_harm_" + t][:, 0])
        assert np.array_equal(b, g["bright_breath_" + t][:, 0])


def test_stft_istft_bit_exact():
    g = golden("stft_istft")
    for tag in g["cases"]:
        n_fft, hop = (int(v) for v in g[f"geo_{tag}"])
        win = R.sqrt_hann(n_fft)
        x = g[f"x_{tag}"]
        S = R.stft(x, n_fft, hop, win)
        assert S.dtype == np.complex64 and S.shape == g[f"S_{tag}"].shape
        assert np.array_equal(S, g[f"S_{tag}"])
        assert np.array_equal(R.istft(g[f"S_{tag}"], hop, win, length=len(x)), g[f"y_{tag}"])
        assert np.array_equal(R.istft(g[f"S_{tag}"], hop, win, length=len(x) + 300), g[f"ylong_{tag}"])
    y = R.overlap_add(g["ola_frames"], R.sqrt_hann(1024), 256, 1024 + 256 * 5)
    assert np.array_equal(y, g["ola_y"])


def test_overlap_add_native_equals_numpy(monkeypatch):
    g = golden("stft_istft")
    a = R.overlap_add(g["ola_frames"], R.sqrt_hann(1024), 256, 1024 + 256 * 5)
    monkeypatch.setattr(R, "_native", lambda: None)
    b = R.overlap_add(g["ola_frames"], R.sqrt_hann(1024), 256, 1024 + 256 * 5)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("native", [True, False])
def test_pulse_train(native, monkeypatch):
    g = golden("pulse_train")
    if not native:
        monkeypatch.setattr(R, "_native", lambda: None)
    for name in list(g["names"]) + ["sr96"]:
        sr = 96000 if name == "sr96" else 44100
        f0 = g["f0_" + name]
        if not native and name not in ("gaps", "many", "knife", "sr96"):
            continue
        p, oi, ot = R.pulse_train(f0, sr, return_onsets=True)
        ref = g["pulse_" + name]
        # onsets are exact (a one-sample slip would show as an O(1) error); shapes agree to fp32
        # rounding of the stub-imported reference's fp32 shape arithmetic (see oracle header)
        assert np.max(np.abs(p - ref)) < 5e-6, name
        if name == "knife":
            # f0/sr = 0.01 is not a binary fraction: the sequential fp64 sum lands either side of the
            # integers, so onsets jitter by one sample — the discontinuity SURVEY §7.3-1 warns about
            assert set(np.diff(oi)) <= {99, 100, 101} and len(set(np.diff(oi))) > 1
            assert np.all(ot == 100)
        if name == "silent":
            assert oi.size == 0 and not p.any()


def test_pulse_train_other_lf_models():
    """Ra / Rg / Rk other than gf.synthesize's constants (GOOFER.py:474, 508-519): four models x three f0 curves written by
    the reference, one of them with pulses longer than 2048 samples."""
    g = golden("pulse_train_lf")
    sr = int(g["sr"])
    for name in g["names"]:
        for m, (Ra, Rg, Rk) in enumerate(g["models"]):
            p = R.pulse_train(g["f0_" + str(name)], sr, Ra=float(Ra), Rg=float(Rg), Rk=float(Rk))
            assert np.max(np.abs(p - g["pulse_%s_%d" % (name, m)])) < 5e-6, (name, m)


def test_pulse_native_matches_python_bitwise(monkeypatch):
    g = golden("pulse_train")
    f0 = g["f0_many"]
    a, ai, at = R.pulse_train(f0, 44100, return_onsets=True)
    monkeypatch.setattr(R, "_native", lambda: None)
    b, bi, bt = R.pulse_train(f0, 44100, return_onsets=True)
    assert np.array_equal(ai, bi) and np.array_equal(at, bt)
    assert np.max(np.abs(a - b)) < 2e-7   # libm vs numpy transcendental last-bit differences


def test_lf_pulse():
    g = golden("pulse_train")
    for i in range(3):
        T, Rk = (float(v) for v in g[f"lf_args_{i}"])  # python floats: fp32 math under numpy weak-scalar rules
        assert np.array_equal(R.lf_pulse(T, Ra=0.02, Rg=1.7, Rk=Rk, sr=44100), g[f"lf_{i}"])


def test_gauss():
    g = golden("gauss")
    env = g["env"]
    for s in (0.5, 1.75, 2.0, 3.4, 7.0):
        out = R.gauss1d(env, s, axis=0)
        assert out.dtype == np.float64
        np.testing.assert_allclose(out, g["ax0_s%g" % s], rtol=1e-13, atol=1e-15)
    out = R.gauss2d(g["cx"], (0.5, 0))
    assert out.dtype == np.complex128
    np.testing.assert_allclose(out, g["cx_s0.5"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(R.gauss1d(g["vec"].astype(np.float32), 25.0), g["vec_s25"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(R.gauss1d(g["vec"], 4), g["vec_s4"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(R.gauss1d(g["small"], 2.0, axis=0), g["small_s2"], rtol=1e-13)


def test_mask_interp_stretch():
    g = golden("mask_interp")
    a = R.smooth_mask(g["mask"], 100, 4)
    assert a.dtype == np.float32
    assert np.max(np.abs(a - g["smooth_100"])) < 1e-7
    assert np.max(np.abs(R.smooth_mask(g["mask"], 1, 4) - g["smooth_1"])) < 1e-7
    assert np.array_equal(R.LinInterp(g["ix"], g["iy"])(g["iq"]), g["interp_extrap"])
    assert np.array_equal(R.LinInterp(g["ix"], g["iy"], fill=0)(g["iq"]), g["interp_fill0"])
    assert np.array_equal(R.stretch_feature(g["feat"], 1.7), g["stretch_2d_1.7"])
    assert np.array_equal(R.stretch_feature(g["feat"], 0.6), g["stretch_2d_0.6"])
    assert np.array_equal(R.stretch_feature(g["feat"][0], 2.3), g["stretch_1d_2.3"])


def test_knots():
    g = golden("knots")
    pack = {"mode": "knots", "knot_vals_log": g["knot_vals_log"], "hz_knots": g["hz_knots"],
            "n_bins": 513, "n_fft": 1024, "sr": 44100}
    dec = R.decode_env_from_knots(pack)
    assert dec.dtype == np.float32
    np.testing.assert_allclose(dec, g["decoded"], rtol=3e-7)
    for sr, n_fft, K in ((44100, 1024, 32), (44100, 1024, 192), (96000, 2048, 64)):
        fr, hz = R.mel_knots(sr, n_fft, K)
        assert np.array_equal(hz, g[f"mel_hz_{sr}_{K}"])
        assert np.array_equal(R.lerp_matrix(fr, hz), g[f"W_{sr}_{K}"])
    for tag in ("an", "sm"):
        p = R.compress_env_to_knots(g[tag + "_env"], 44100, 1024)
        assert np.array_equal(p["hz_knots"], g[tag + "_hz_knots"])
        assert p["knot_vals_log"].dtype == np.float16
        assert np.array_equal(p["knot_vals_log"], g[tag + "_knot_vals_log"])
    env, p = R.envelope_of(g["an_x"], 44100)
    np.testing.assert_allclose(env, g["an_env"], rtol=1e-12)


def test_warps():
    g = golden("warps")
    env = g["env"]
    for r in (0.75, 1.25, 1.5, 0.5):
        out = R.shift_formants(env, r, 44100)
        assert out.dtype == np.float32
        assert np.array_equal(out, g["shift_%g" % r])
    F = g["formants"]
    for i in range(3):
        out = R.warp_env_by_formants(env, F, F * g[f"ratios_{i}"][:, None], 44100)
        assert np.array_equal(out, g[f"warp_{i}"])


def _synth_case(g, name):
    sr, n_fft, hop, seed = (int(v) for v in g[f"{name}_geo"])
    kw = {k: float(v) for k, v in zip(g[f"{name}_kw_keys"], g[f"{name}_kw_vals"]) if k != "_"}
    if "apply_brightness" in kw:
        kw["apply_brightness"] = bool(kw["apply_brightness"])
    env = g[f"{name}_env"]
    n = len(g[f"{name}_f0"])
    T = 1 + n // hop
    phi = np.random.default_rng(seed).uniform(0.0, 2.0 * np.pi, size=(env.shape[0], T)).astype(np.float32)
    F = g[f"{name}_formants"]
    return dict(env=env, f0=g[f"{name}_f0"], mask=g[f"{name}_mask"], n=n, sr=sr, n_fft=n_fft, hop=hop, phi=phi,
                formants={i + 1: F[i] for i in range(4)}, kw=kw)


def test_synthesize_against_reference():
    g = golden("synthesize")
    for name in g["names"]:
        c = _synth_case(g, name)
        rec, harm, uv, bre = R.synthesize(c["env"], c["f0"], c["mask"], np.empty(c["n"], bool), c["sr"],
                                          n_fft=c["n_fft"], hop_length=c["hop"], formants=c["formants"],
                                          phi=c["phi"], **c["kw"])
        for got, key in ((rec, "rec"), (harm, "harm"), (uv, "uv"), (bre, "bre")):
            ref = g[f"{name}_{key}"]
            assert got.dtype == np.float32 and got.shape == ref.shape
            # tolerance: fp32 shape arithmetic in the stub-imported pulse train (oracle header)
            assert rms_err(got, ref) < 2e-6 * max(1.0, float(np.max(np.abs(ref)))), (name, key, rms_err(got, ref))


ROUGH_KW = {
    "default": dict(roughness_on=True),
    "custom": dict(roughness_on=True, rough_k_list=(2, 5), rough_h_list=[0.5, 0.2], rough_alpha=0.8, rough_hp_fc=180.0,
                   rough_noise_amp=0.3, rough_noise_smooth_ms=60.0, rough_alpha_slew_ms=40.0, normalize=0.5, pitch_shift=1.2),
    "three_defaults_more_k": dict(roughness_on=True, rough_k_list=(2, 3, 4, 6)),
}


def rough_case(g):
    sr, n_fft, hop, seed = (int(v) for v in g["geo"])
    env = g["env"]
    T = 1 + len(g["f0"]) // hop
    phi = np.random.default_rng(seed).uniform(0.0, 2.0 * np.pi, size=(env.shape[0], T)).astype(np.float32)
    return dict(env=env, f0=g["f0"], mask=g["mask"], n=len(g["f0"]), sr=sr, n_fft=n_fft, hop=hop, phi=phi,
                formants={i + 1: g["formants"][i] for i in range(4)})


def test_vocal_roughness_against_reference():
    """gf.synthesize(roughness_on=True): only `reconstruct` hears the roughness layer, the stems share its peak gain."""
    g = golden("synthesize_rough")
    c = rough_case(g)
    for name in g["names"]:
        out = R.synthesize(c["env"], c["f0"], c["mask"], np.empty(c["n"], bool), c["sr"], n_fft=c["n_fft"], hop_length=c["hop"],
                           formants=c["formants"], phi=c["phi"], **ROUGH_KW[str(name)])
        for got, key in zip(out, ("rec", "harm", "uv", "bre")):
            ref = g[f"{name}_{key}"]
            assert got.dtype == np.float32 and got.shape == ref.shape
            assert rms_err(got, ref) < 2e-6 * max(1.0, float(np.max(np.abs(ref)))), (name, key, rms_err(got, ref))
    plain = R.synthesize(c["env"], c["f0"], c["mask"], np.empty(c["n"], bool), c["sr"], n_fft=c["n_fft"], hop_length=c["hop"],
                         formants=c["formants"], phi=c["phi"])
    assert rms_err(plain[0], g["default_rec"]) > 1e-3          # the layer is audible


def test_goofy_files(tmp_path):
    import os
    from conftest import GOLDEN
    g = golden("goofy_roundtrip")
    env, f0, mask, forms, sr, ylen = R.load_features(os.path.join(GOLDEN, "sample_features.goofy"))
    assert np.array_equal(env["knot_vals_log"], g["knot_vals_log"]) and env["knot_vals_log"].dtype == np.float16
    assert np.array_equal(env["hz_knots"], g["hz_knots"])
    assert np.array_equal(f0, g["f0"]) and np.array_equal(mask, g["mask"])
    assert [env["n_bins"], env["n_fft"], env["sr"], sr, ylen] == [int(v) for v in g["meta"]]
    for i in range(4):
        assert np.array_equal(forms[i + 1], g["formants"][i])
    # write with the oracle, read back: identical arrays, and the member list matches the reference file
    p = tmp_path / "x_features.goofy"
    R.save_features(p, env, f0, mask, forms, sr, ylen)
    a = np.load(p, allow_pickle=True)
    b = np.load(os.path.join(GOLDEN, "sample_features.goofy"), allow_pickle=True)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k == "formants":
            continue
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
    g2 = golden("goofy_roundtrip_full")
    env2, f02, m2, forms2, sr2, ylen2 = R.load_features(os.path.join(GOLDEN, "sample_full_features.goofy"))
    assert np.array_equal(env2, g2["env"]) and np.array_equal(f02, g2["f0"])
    for k in (1, 2, 3, 4):
        assert np.array_equal(forms2[k], g2["formant_%d" % k])


def test_product_interp1d_matches_oracle_interpolant():
    """goofer_amd.core.interp1d (host helper of the reference's module surface) against the oracle's LinInterp."""
    from goofer_amd import core
    rng = np.random.default_rng(4)
    x = np.sort(rng.random(12)) * 10
    y = rng.standard_normal(12)
    q = np.linspace(-3, 14, 101)
    assert np.array_equal(core.interp1d(x, y)(q), R.LinInterp(x, y)(q))
    assert np.array_equal(core.interp1d(x, y, fill_value=0.5)(q), R.LinInterp(x, y, fill=0.5)(q))
    assert np.array_equal(core.interp1d([2.0], [7.0])(q), R.LinInterp([2.0], np.array([7.0]))(q))
    with pytest.raises(ValueError):
        core.interp1d([], [])
    with pytest.raises(ValueError):
        core.interp1d(x, y, kind="cubic")


def test_stretched_f0_is_the_float64_array_the_reference_holds():
    """gf.synthesize's time stretch turns f0_interp into a float64 array (np.interp inside the reference's interp1d,
    GOOFER.py:173-239, 1053); the f0 jitter and the sub-harmonic phase trackers then work on THAT array.  core._stretch64 makes it
    for goofer_batch.f0_64: against the reference's own stretch fixture and the oracle, bit for bit, float32 input included."""
    from goofer_amd import core
    g = golden("mask_interp")
    got = core._stretch64(g["feat"][0], 0, None, 2.3)
    assert got.dtype == np.float64 and np.array_equal(got, g["stretch_1d_2.3"])
    rng = np.random.default_rng(0)
    for n, fac in ((1000, 1.23), (777, 0.61), (2, 1.5), (5000, 1.499)):
        x = (rng.random(n) * 400 + 80).astype(np.float32)
        a = core._stretch64(x, 0, None, fac)
        assert a.dtype == np.float64 and np.array_equal(a, R.stretch_feature(x, fac))
    x = (rng.random(3000) * 400 + 80).astype(np.float32)
    want = np.concatenate([x[:500], R.stretch_feature(x[500:2000], 1.3), x[2000:]])
    assert want.dtype == np.float64 and np.array_equal(core._stretch64(x, 500, 2000, 1.3), want)
    assert core._stretch64(x[:1], 0, None, 2.0) is None            # (one sample: the reference's result keeps float32)
