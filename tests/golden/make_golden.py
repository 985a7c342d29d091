#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference CPU path.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
The reference's third-party imports that are absent here (numba, soundfile, parselmouth,
tkinter, sounddevice) are replaced by inert stubs *before* import; ``numba.njit`` becomes the
identity decorator, so the ``@njit`` loops run as plain Python with the same arithmetic.

Harness corrections (SURVEY.md §8 c), so that the stubbed run equals the real (numba) reference:
  * ``pulse_train_numba`` is called with ``sr`` as ``np.float64`` — numba types ``f0[i]/sr`` as
    fp64; numpy-2 scalar promotion would otherwise accumulate the phase in fp32.
  * ``np.random.default_rng`` is replaced by a seeded factory (seed recorded in each fixture) and
    ``np.random.seed`` is called before the legacy-RNG flags (sh / sr / sd).
  * ``soundfile.read/write`` are an in-memory dict; ``gf.load_features`` is fed from memory for
    the index-plan cases (fp16 storage would destroy the index encoding).

Only *data* is written: seeded inputs and the reference's outputs.  No reference source text.

Usage:  python tests/golden/make_golden.py            (rewrites every fixture)
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from goofer_amd import synthetic as syn  # noqa: E402

# --------------------------------------------------------------------------------------------
# stubs + import
# --------------------------------------------------------------------------------------------
_WAVS = {}
_WRITTEN = {}


def _njit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def _sf_read(path, *a, **k):
    y, sr = _WAVS[str(path)]
    return y.copy(), sr


def _sf_write(path, data, sr, *a, **k):
    _WRITTEN[str(path)] = (np.array(data), sr)


def _install_stubs():
    nb = types.ModuleType("numba")
    nb.njit = _njit
    sf = types.ModuleType("soundfile")
    sf.read = _sf_read
    sf.write = _sf_write
    mods = {"numba": nb, "soundfile": sf, "parselmouth": types.ModuleType("parselmouth"),
            "tkinter": types.ModuleType("tkinter"), "tkinter.ttk": types.ModuleType("tkinter.ttk"),
            "sounddevice": types.ModuleType("sounddevice")}
    for k, v in mods.items():
        sys.modules[k] = v


_install_stubs()
sys.path.insert(0, "/root/reference")
import GOOFER as gf  # noqa: E402
import SillySampler as ss  # noqa: E402

_orig_pulse = gf.pulse_train_numba
gf.pulse_train_numba = lambda f0, sr, **k: _orig_pulse(f0, np.float64(sr), **k)

_orig_default_rng = np.random.default_rng
_RNG_SEED = [None]


def _seeded_default_rng(*a, **k):
    if a or k:
        return _orig_default_rng(*a, **k)
    assert _RNG_SEED[0] is not None, "unseeded default_rng() in golden run"
    return _orig_default_rng(_RNG_SEED[0])


np.random.default_rng = _seeded_default_rng


# the un-JITted _overlap_add is a 0.2 s python loop; this slice form adds the same fp32 terms in
# the same order (frame-major, one add per sample per frame) and was checked bit-identical below.
def _fast_overlap_add(frames, window, hop_length, expected_len):
    n_fft, n_frames = frames.shape
    y = np.zeros(expected_len, dtype=np.float32)
    ws = np.zeros(expected_len, dtype=np.float32)
    w = window.astype(np.float32)
    w2 = w * w
    for i in range(n_frames):
        s = i * hop_length
        y[s:s + n_fft] += frames[:, i] * w
        ws[s:s + n_fft] += w2
    nz = ws > 1e-9
    y[nz] /= ws[nz]
    return y


def _check_fast_ola():
    r = _orig_default_rng(3)
    fr = r.standard_normal((64, 9)).astype(np.float32)
    w = np.hanning(64).astype(np.float32) ** 0.5
    a = gf._overlap_add(fr, w, 16, 64 + 16 * 8)
    b = _fast_overlap_add(fr, w, 16, 64 + 16 * 8)
    assert np.array_equal(a, b), "fast OLA is not bit-identical"


_check_fast_ola()
_slow_ola = gf._overlap_add
gf._overlap_add = _fast_overlap_add


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------------------------
# function-level vectors
# --------------------------------------------------------------------------------------------
def gen_stft_istft():
    out = {}
    r = _orig_default_rng(11)
    cases = [("a", 4000, 1024, 256), ("b", 700, 1024, 256), ("c", 1, 1024, 256),
             ("d", 3000, 2048, 96), ("e", 256, 1024, 256), ("f", 5000, 512, 128)]
    for tag, n, n_fft, hop in cases:
        x = r.standard_normal(n).astype(np.float32)
        win = gf.get_cached_window(44100, n_fft)
        S = gf.stft(x, n_fft=n_fft, hop_length=hop, window=win)
        y = gf.istft(S, hop_length=hop, window=win, length=n)
        y2 = gf.istft(S, hop_length=hop, window=win, length=n + 300)
        out.update({f"x_{tag}": x, f"S_{tag}": S, f"y_{tag}": y, f"ylong_{tag}": y2,
                    f"geo_{tag}": np.array([n_fft, hop])})
    # slow reference OLA on one case (pins the fast form used for every other fixture)
    fr = r.standard_normal((1024, 6)).astype(np.float32)
    out["ola_frames"] = fr
    out["ola_y"] = _slow_ola(fr, gf.get_cached_window(44100, 1024), 256, 1024 + 256 * 5)
    out["cases"] = np.array([c[0] for c in cases])
    save("stft_istft", **out)


def gen_tables():
    out = {}
    for sr, n_fft in ((44100, 1024), (96000, 2048), (48000, 512)):
        t = f"{sr}_{n_fft}"
        out["win_" + t] = gf.get_cached_window(sr, n_fft)
        out["freqs_" + t] = gf.get_cached_freqs(sr, n_fft)
        out["boost_" + t] = gf.get_cached_boost(sr, n_fft)
        h, b = gf.get_cached_brightness(sr, n_fft)
        out["bright_harm_" + t] = h
        out["bright_breath_" + t] = b
    save("tables", **out)


def gen_pulse():
    out = {}
    r = _orig_default_rng(21)
    sr = 44100
    n = 6000
    t = np.arange(n) / sr
    f0s = {
        "glide": 180.0 + 120.0 * t / t[-1] + 7.0 * np.sin(2 * np.pi * 5.3 * t),
        "gaps": np.where((t > 0.03) & (t < 0.09), 0.0, 233.3 + 20 * np.sin(40 * t)),
        "low": np.full(n, 41.7),          # long pulses (T0 = 1058) overlapping note end
        "high": 900.0 + 300 * np.sin(9 * t),  # short pulses, T0 ~ 37..73
        "jump": np.where(t < 0.06, 55.0, 610.0),  # long pulse under many short ones
        "many": 150.0 + 400.0 * (np.arange(n) % 700) / 700.0,  # > 5 distinct T0 (cache eviction)
        "silent": np.zeros(n),
        "knife": np.full(n, 441.0),       # f0/sr exactly 0.01: onset every 100 samples, knife-edge
    }
    for k, f in f0s.items():
        f = f.astype(np.float32)
        out["f0_" + k] = f
        out["pulse_" + k] = gf.pulse_train_numba(f, sr, Ra=0.02, Rg=1.7, Rk=0.8)
    f = (300 + 50 * r.standard_normal(2500)).astype(np.float32)
    out["f0_sr96"] = f
    out["pulse_sr96"] = gf.pulse_train_numba(f, 96000, Ra=0.02, Rg=1.7, Rk=0.8)
    out["names"] = np.array(list(f0s.keys()))
    # LF pulse (numpy form, used by the sub-harmonic layer)
    for i, (T, Rk) in enumerate(((1 / 440.0, 1.0), (1 / 97.3, 1.0), (1 / 3000.0, 0.34))):
        out[f"lf_{i}"] = gf.lf_model_pulse(T, Ra=0.02, Rg=1.7, Rk=Rk, sr=sr)
        out[f"lf_args_{i}"] = np.array([T, Rk])
    save("pulse_train", **out)


def gen_pulse_lf():
    """pulse_train_numba with Ra / Rg / Rk other than the constants gf.synthesize passes (GOOFER.py:474, 508-519)."""
    out = {}
    sr = 44100
    n = 5000
    t = np.arange(n) / sr
    f0s = {
        "glide": 140.0 + 200.0 * t / t[-1] + 5.0 * np.sin(2 * np.pi * 4.1 * t),
        "low": np.full(n, 18.3),          # T0 = 2410: longer than the shape table, evaluated on the fly
        "gaps": np.where((t > 0.02) & (t < 0.05), 0.0, 311.0 + 15 * np.sin(33 * t)),
    }
    models = [(0.01, 1.47, 0.34), (0.05, 3.0, 1.0), (0.2, 0.5, 0.6), (0.02, 1.7, 0.8)]
    for k, f in f0s.items():
        f = f.astype(np.float32)
        out["f0_" + k] = f
        for m, (Ra, Rg, Rk) in enumerate(models):
            out["pulse_%s_%d" % (k, m)] = gf.pulse_train_numba(f, sr, Ra=Ra, Rg=Rg, Rk=Rk)
    out["names"] = np.array(list(f0s.keys()))
    out["models"] = np.array(models, dtype=np.float64)
    out["sr"] = np.array(sr)
    save("pulse_train_lf", **out)


def gen_gauss():
    out = {}
    r = _orig_default_rng(31)
    env = np.abs(r.standard_normal((513, 12))).astype(np.float32) + 0.01
    out["env"] = env
    for s in (0.5, 1.75, 2.0, 3.4, 7.0):
        out["ax0_s%g" % s] = gf.gaussian_filter1d(env, sigma=s, axis=0)
    cx = (r.standard_normal((513, 5)) + 1j * r.standard_normal((513, 5))).astype(np.complex64)
    out["cx"] = cx
    out["cx_s0.5"] = gf.gaussian_filter(cx, sigma=(0.5, 0))
    v = r.standard_normal(900)
    out["vec"] = v
    out["vec_s25"] = gf.gaussian_filter1d(v.astype(np.float32), sigma=25.0)
    out["vec_s4"] = gf.gaussian_filter1d(v, sigma=4)
    small = np.abs(r.standard_normal((9, 3))).astype(np.float32)
    out["small"] = small
    out["small_s2"] = gf.gaussian_filter1d(small, sigma=2.0, axis=0)  # radius 8 on 9 bins
    save("gauss", **out)


def gen_mask_interp():
    out = {}
    r = _orig_default_rng(41)
    n = 9001
    m = (r.uniform(size=n) > 0.5).astype(np.float32)
    m[:2000] = 0
    m[2000:6000] = 1
    out["mask"] = m
    out["smooth_100"] = gf.smooth_mask_ds(m, sigma=100, ds=4)
    out["smooth_1"] = gf.smooth_mask_ds(m, sigma=1, ds=4)
    x = np.sort(r.uniform(0, 10, 17))
    y = r.standard_normal(17)
    q = np.linspace(-2, 12, 101)
    out["ix"], out["iy"], out["iq"] = x, y, q
    out["interp_extrap"] = gf.interp1d(x, y, fill_value="extrapolate")(q)
    out["interp_fill0"] = gf.interp1d(x, y, fill_value=0)(q)
    feat = r.standard_normal((7, 23))
    out["feat"] = feat
    out["stretch_2d_1.7"] = gf.stretch_feature(feat, 1.7)
    out["stretch_2d_0.6"] = gf.stretch_feature(feat, 0.6)
    out["stretch_1d_2.3"] = gf.stretch_feature(feat[0], 2.3)
    save("mask_interp", **out)


def gen_knots():
    out = {}
    src = syn.make_source(77, seconds=0.25)
    pack = src["env_pack"]
    env = gf.decode_env_from_knots(pack)
    out["knot_vals_log"] = pack["knot_vals_log"]
    out["hz_knots"] = pack["hz_knots"]
    out["decoded"] = env
    for sr, n_fft, K in ((44100, 1024, 32), (44100, 1024, 192), (96000, 2048, 64)):
        fr, hz = gf.make_mel_knots(sr, n_fft, K)
        out[f"mel_hz_{sr}_{K}"] = hz
        out[f"W_{sr}_{K}"] = gf.precompute_interp_matrix(fr, hz)
    # encode: envelope from a real analysis (stft of noise-excited resonances)
    r = _orig_default_rng(51)
    x = r.standard_normal(6000).astype(np.float32)
    x = np.convolve(x, np.exp(-np.arange(40) / 6.0) * np.cos(np.arange(40) * 0.4), mode="same")
    S = gf.stft(x, 1024, 256, gf.get_cached_window(44100, 1024))
    mag = np.abs(S) + 1e-8
    env_an = gf.gaussian_filter1d(mag, sigma=2.0, axis=0)
    packed = gf.compress_env_to_knots(env_an, 44100, 1024)
    out["an_x"] = x.astype(np.float32)
    out["an_env"] = env_an
    out["an_knot_vals_log"] = packed["knot_vals_log"]
    out["an_hz_knots"] = packed["hz_knots"]
    smooth = gf.decode_env_from_knots(pack).astype(np.float64)
    packed2 = gf.compress_env_to_knots(smooth, 44100, 1024)
    out["sm_env"] = smooth
    out["sm_knot_vals_log"] = packed2["knot_vals_log"]
    out["sm_hz_knots"] = packed2["hz_knots"]
    save("knots", **out)


def gen_warps():
    out = {}
    src = syn.make_source(78, seconds=0.2)
    env = gf.decode_env_from_knots(src["env_pack"])
    T = env.shape[1]
    out["env"] = env
    for ratio in (0.75, 1.25, 1.5, 0.5):
        out["shift_%g" % ratio] = gf.shift_formants(env, ratio, 44100)
    F = np.stack([src["formants"][i][:T] for i in (1, 2, 3, 4)], 0)
    F[0, 3] = 0.0       # dropped anchor
    F[2, 5] = 30000.0   # above nyquist -> dropped
    F[1, 7] = 40.0      # below 50 Hz -> dropped
    out["formants"] = F
    for i, ratios in enumerate(([1.3, 0.8, 1.1, 0.9], [0.7, 1.2, 0.9, 1.1], [1.0, 1.0, 1.5, 1.0])):
        sh = gf.transpose_formants_array(F, ratios)
        out[f"ratios_{i}"] = np.array(ratios)
        out[f"warp_{i}"] = gf.warp_env_by_formants(env, F, sh, 44100)
    save("warps", **out)


def gen_synthesize():
    """gf.synthesize with pinned phi on small notes (a10)."""
    cases = [
        ("plain", dict(), 44100, 1024, 256, 0.30),
        ("g_up", dict(formant_shift=1.25, pitch_shift=1.0), 44100, 1024, 256, 0.25),
        ("g_dn_fshift", dict(formant_shift=0.75, F1_shift=1.3, F2_shift=0.8, F3_shift=1.1, F4_shift=0.9), 44100, 1024, 256, 0.25),
        ("norm_half", dict(normalize=0.5, uv_strength=0.4, breath_strength=0.05), 44100, 1024, 256, 0.2),
        ("nobright", dict(apply_brightness=False, normalize=0.0), 44100, 1024, 256, 0.2),
        ("sr96", dict(), 96000, 2048, 96, 0.08),
        ("unvoiced", dict(), 44100, 1024, 256, 0.15),
        ("sa_like", dict(uv_strength=1.0, breath_strength=1.0, noise_transition_smoothness=1), 44100, 1024, 256, 0.2),
    ]
    out = {"names": np.array([c[0] for c in cases])}
    for idx, (name, kw, sr, n_fft, hop, secs) in enumerate(cases):
        src = syn.make_source(300 + idx, sr, n_fft, hop, seconds=secs)
        env = gf.decode_env_from_knots(src["env_pack"])
        n = src["y_len"]
        t = np.arange(n) / sr
        mask = src["mask"].copy()
        if name == "unvoiced":
            mask[:] = 0.0
        if name == "sa_like":
            mask[:] = 1.0
        f0 = (196.0 * 2 ** (0.3 * np.sin(2 * np.pi * 3.1 * t))) * mask
        seed = 7000 + idx
        _RNG_SEED[0] = seed
        rec, harm, uv, bre = gf.synthesize(env, f0.astype(np.float64), mask, np.empty(n, bool), sr,
                                           n_fft=n_fft, hop_length=hop, formants=src["formants"], **kw)
        _RNG_SEED[0] = None
        out[f"{name}_env"] = env
        out[f"{name}_f0"] = f0
        out[f"{name}_mask"] = mask
        out[f"{name}_formants"] = np.stack([src["formants"][i] for i in (1, 2, 3, 4)], 0)
        out[f"{name}_geo"] = np.array([sr, n_fft, hop, seed])
        out[f"{name}_kw_keys"] = np.array(list(kw.keys()) or ["_"])
        out[f"{name}_kw_vals"] = np.array([float(v) for v in kw.values()] or [0.0])
        out[f"{name}_rec"], out[f"{name}_harm"], out[f"{name}_uv"], out[f"{name}_bre"] = rec, harm, uv, bre
    save("synthesize", **out)


def gen_synthesize_rough():
    """gf.synthesize(roughness_on=True) (GOOFER.py:901-940, 1195-1206): defaults, and every rough_* argument moved."""
    cases = [
        ("default", dict(roughness_on=True)),
        ("custom", dict(roughness_on=True, rough_k_list=(2, 5), rough_h_list=[0.5, 0.2], rough_alpha=0.8, rough_hp_fc=180.0,
                        rough_noise_amp=0.3, rough_noise_smooth_ms=60.0, rough_alpha_slew_ms=40.0, normalize=0.5,
                        pitch_shift=1.2)),
        ("three_defaults_more_k", dict(roughness_on=True, rough_k_list=(2, 3, 4, 6))),
    ]
    src = syn.make_source(300, 44100, 1024, 256, seconds=0.30)
    env = gf.decode_env_from_knots(src["env_pack"])
    n = src["y_len"]
    t = np.arange(n) / 44100
    mask = src["mask"].copy()
    f0 = (196.0 * 2 ** (0.3 * np.sin(2 * np.pi * 3.1 * t))) * mask
    out = {"names": np.array([c[0] for c in cases]), "env": env, "f0": f0, "mask": mask,
           "formants": np.stack([src["formants"][i] for i in (1, 2, 3, 4)], 0), "geo": np.array([44100, 1024, 256, 7000])}
    for name, kw in cases:
        _RNG_SEED[0] = 7000
        rec, harm, uv, bre = gf.synthesize(env, f0.astype(np.float64), mask, np.empty(n, bool), 44100, n_fft=1024, hop_length=256,
                                           formants=src["formants"], **kw)
        _RNG_SEED[0] = None
        out[f"{name}_rec"], out[f"{name}_harm"], out[f"{name}_uv"], out[f"{name}_bre"] = rec, harm, uv, bre
    save("synthesize_rough", **out)


# --------------------------------------------------------------------------------------------
# sampler-level vectors (a13-a20)
# --------------------------------------------------------------------------------------------
def gen_flags_pitch():
    flag_strings = ["", "t0g0", "t+12g-50", "fa30fb-20fc10fd-10fw50fst40fsta20fstb-20fstc10fstd-10V80B20U-30",
                    "L1", "l2", "L7", "br40es-50", "BR40ES60", "R1FV1P50", "sh50sr50sg50", "su50sj30sa30sd30",
                    "st-50vf40vh60vl30pd50", "g", "B/U-20", "Mt50", "se1SE1", "P150", "V150", "fstA20FST-130"]
    out = {"flag_strings": np.array(flag_strings)}
    import json
    out["parsed"] = np.array([json.dumps(ss.parse_flags(f)) for f in flag_strings])
    pitch_strings = ["AA", "AA#5#", "AB#3#AC/+//", "", "4e4f4g#10#AAABAC", "//#2#gA", syn.encode_cents(range(-2048, 2048, 37))]
    out["pitch_strings"] = np.array(pitch_strings)
    for i, p in enumerate(pitch_strings):
        out[f"cents_{i}"] = ss.pitch_string_to_cents(p)
    notes = ["C4", "A4", "C#5", "G#2", "B-1", "A#7", "F3"]
    out["notes"] = np.array(notes)
    out["midi"] = np.array([ss.note_to_midi(n) for n in notes])
    out["hz"] = np.array([ss.midi_to_hz(m) for m in (0, 57, 69, 69.5, 127)])
    # flag -> parameter scaling (GooferResampler.__init__ without the render)
    attrs = ["formant_shift", "brightness_env", "F1_shift", "F2_shift", "F3_shift", "F4_shift", "f0_jitter",
             "f0_jitter_strength", "volume_jitter", "volume_jitter_strength", "sd_strength", "breathiness_mix",
             "unvoiced_mix", "harmonic_mix", "loop_mode", "tension", "subharm_weight", "add_subharm", "reverse",
             "growl_mix", "aperiodic_mix", "subharm_gain", "normalize", "env_shape", "force_voiced", "pitch_dyn",
             "formant_width", "formant_strength_f1", "formant_strength_f2", "formant_strength_f3",
             "formant_strength_f4", "use_editor", "offset", "length", "consonant", "cutoff", "volume", "tempo",
             "velocity", "pitch_m"]
    saved_render = ss.GooferResampler.render
    ss.GooferResampler.render = lambda self: None
    rows = []
    for f in flag_strings:
        try:
            o = ss.GooferResampler("a.wav", "b.wav", "C4", "100", f, "50", "1000", "100", "-250", "80", "0", "!125", "AA")
            rows.append(json.dumps({a: (getattr(o, a) if isinstance(getattr(o, a), (str, bool)) else float(getattr(o, a))) for a in attrs}))
        except Exception as e:  # bare flag letter -> TypeError in arithmetic
            rows.append(json.dumps({"error": type(e).__name__}))
    ss.GooferResampler.render = saved_render
    out["params"] = np.array(rows)
    out["split_in"] = np.array(["C:/my voice/a b.wav C:/out dir/o.wav C4 100 g0 0 1000 0 700 100 0 !120 AA"])
    out["split_out"] = np.array(ss.split_arguments(str(out["split_in"][0])))
    save("flags_pitch", **out)


class _Capture:
    """Record every gf.synthesize call made by one GooferResampler render."""

    def __init__(self):
        self.calls = []
        self._orig = gf.synthesize

    def __enter__(self):
        cap = self

        def wrapped(env, f0, mask, y, sr, **kw):
            ins = dict(env=np.array(env), f0=np.array(f0), mask=np.array(mask), n=len(y), sr=sr,
                       kw={k: v for k, v in kw.items() if k != "formants"},
                       formants={k: np.array(v) for k, v in kw.get("formants", {}).items()})
            res = cap._orig(env, f0, mask, y, sr, **kw)
            cap.calls.append((ins, [np.array(r) for r in res]))
            return res

        gf.synthesize = wrapped
        return self

    def __exit__(self, *a):
        gf.synthesize = self._orig


def _run_sampler(src, req, seed, tmp, mem_features=None, legacy_seed=None):
    """One reference render.  Returns (out, sr, capture, locals-of-resample)."""
    wav = os.path.join(tmp, "src.wav")
    outp = os.path.join(tmp, "out.wav")
    feat = os.path.join(tmp, "src_features.goofy")
    _WAVS[wav] = (np.zeros(src["y_len"]), src["sr"])
    gf.save_features(feat, src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    saved_load = gf.load_features
    if mem_features is not None:
        gf.load_features = lambda p: mem_features
    loc = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name == "resample":
            def local_trace(fr, ev, a):
                if ev == "return":
                    for k in ("start_sample", "consonant_sample", "end_sample", "start_frame", "consonant_frame",
                              "end_frame", "desired_tail_frames", "desired_tail_samples", "tail_frames", "tail_len",
                              "pre_frames", "pre_samples", "vel_factor"):
                        if k in fr.f_locals:
                            loc[k] = fr.f_locals[k]
                return local_trace
            return local_trace
        return None

    _RNG_SEED[0] = seed
    if legacy_seed is not None:
        np.random.seed(legacy_seed)
    sys.settrace(tracer)
    try:
        with _Capture() as cap:
            ss.GooferResampler(wav, outp, *syn.request_args(req))
    finally:
        sys.settrace(None)
        _RNG_SEED[0] = None
        gf.load_features = saved_load
    out, sr = _WRITTEN.pop(outp)
    return out, sr, cap, loc


def _pack_formants(d):
    return np.stack([np.asarray(d[k], dtype=np.float64) for k in sorted(d.keys(), key=str)], 0)


SAMPLER_CASES = [
    # name, flags, request overrides, store_synth_io
    ("default", "t0g0", dict(length_ms=300), True),
    ("t12g50", "t12g50", dict(length_ms=250), True),
    ("tm12gm50", "t-12g-50", dict(length_ms=250), True),
    ("formants", syn.FULL_FORMANT_FLAGS, dict(length_ms=250), True),
    ("formants_flip", syn._flip(syn.FULL_FORMANT_FLAGS, 1), dict(length_ms=250), True),
    ("L0", "L0", dict(length_ms=1000), True),
    ("L1", "L1", dict(length_ms=1000), True),
    ("L2", "L2", dict(length_ms=1000), True),
    ("L0_short", "L0", dict(length_ms=150), False),
    ("br_es_neg", "br40es-50", dict(length_ms=250), True),
    ("br_es_pos", "br-40es60", dict(length_ms=250), True),
    ("vel60", "g10", dict(length_ms=250, velocity=60), True),
    ("vel150", "", dict(length_ms=250, velocity=150), True),
    ("R1", "R1", dict(length_ms=300), True),
    ("FV1_P50", "FV1P50", dict(length_ms=250), True),
    ("negcut", "t30", dict(length_ms=400, cutoff_ms=-300), True),
    ("vol_mix", "V60B-40U35", dict(length_ms=250, volume=70), False),
    # "next" rows (§8 f): full output only
    ("su50", "su50", dict(length_ms=250), False),
    ("sj30", "sj30", dict(length_ms=250), False),
    ("sa30", "sa30", dict(length_ms=250), False),
    ("st50", "st50", dict(length_ms=250), False),
    ("stm50", "st-50", dict(length_ms=250), False),
    ("vf40", "vf40", dict(length_ms=250), False),
    ("vfm40", "vf-40vh70vl40", dict(length_ms=250), False),
    ("pd50", "pd50", dict(length_ms=250), False),
    ("pdm50", "pd-50", dict(length_ms=250), False),
    ("sd30", "sd30", dict(length_ms=250), False),
    ("sh50sr50", "sh50sr50", dict(length_ms=250), False),
    ("sg50", "sg50", dict(length_ms=250), False),
]


def gen_sampler():
    import json
    index = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, (name, flags, over, store_io) in enumerate(SAMPLER_CASES):
            src = syn.make_source(2000 + i, seconds=0.45)
            req = syn.make_request(2000 + i, flags, **over)
            seed = 6000 + i
            out, sr, cap, loc = _run_sampler(src, req, seed, tmp, legacy_seed=4000 + i)
            d = {"out": out, "seed": np.array([seed, 4000 + i, 2000 + i]),
                 "args": np.array(syn.request_args(req)),
                 "locals": np.array(json.dumps({k: (float(v) if isinstance(v, float) else int(v)) for k, v in loc.items()})),
                 "n_calls": np.array([len(cap.calls)])}
            ins, res = cap.calls[0]
            d["kw"] = np.array(json.dumps({k: (v if isinstance(v, (bool, int, str)) else float(v)) for k, v in ins["kw"].items()}))
            if store_io:
                d["env_new"] = ins["env"]
                d["f0_new"] = ins["f0"]
                d["mask_new"] = ins["mask"]
                d["formants_new"] = _pack_formants(ins["formants"])
                d["harm"], d["uv"], d["bre"] = res[1], res[2], res[3]
            save("sampler_" + name, **d)
            index.append(name)
    save("sampler_index", names=np.array(index))


N_COMBOS = 16


def gen_sampler_combos():
    """Flag interactions pinned by the reference itself: 16 random subsets of the whole flag vocabulary (assembly edits,
    jitter / sub-harmonic layers and the post chain together), rendered by the real GooferResampler."""
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(N_COMBOS):
            rng = np.random.default_rng(7700 + i)
            flags = syn.random_flags(rng)
            src = syn.make_source(5000 + i, seconds=float(rng.uniform(0.3, 0.55)))
            req = syn.make_request(5000 + i, flags, length_ms=float(rng.integers(200, 700)), offset_ms=float(rng.integers(0, 60)),
                                   consonant_ms=float(rng.integers(0, 120)), cutoff_ms=float(rng.choice([-200, 30, 80])),
                                   velocity=float(rng.choice([60, 100, 140])), volume=float(rng.integers(50, 121)),
                                   tempo=float(rng.choice([90, 120, 150])))
            seed = 6500 + i
            out, sr, cap, _ = _run_sampler(src, req, seed, tmp, legacy_seed=4500 + i)
            name = "combo_%02d" % i
            save(name, out=out, seed=np.array([seed, 4500 + i, 5000 + i]), args=np.array(syn.request_args(req)),
                 seconds=np.array([src["y_len"] / src["sr"]]), n_calls=np.array([len(cap.calls)]))
            names.append(name)
            print(name, flags, "synth calls:", len(cap.calls), flush=True)
    save("combo_index", names=np.array(names))


N_HARD = 12


def gen_sampler_hard():
    """Hard sources (synthetic.make_hard_source: interior V/UV transitions, fractional fp16 mask values, formant frames that are
    0 / NaN / out of range / crossing, 40 dB envelope jumps, near-zero bins) through the real GooferResampler — what real .goofy
    content does to the path, pinned by the reference itself."""
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(N_HARD):
            src, req = syn.hard_case(i)
            seed = 6800 + i
            out, sr, cap, _ = _run_sampler(src, req, seed, tmp, legacy_seed=4800 + i)
            ins, res = cap.calls[0]
            name = "sampler_hard_%02d" % i
            assert np.isfinite(out).all(), name
            save(name, out=out, seed=np.array([seed, 4800 + i, 3000 + i]), args=np.array(syn.request_args(req)),
                 seconds=np.array([src["y_len"] / src["sr"]]), n_calls=np.array([len(cap.calls)]),
                 mask_new=ins["mask"], f0_new=ins["f0"].astype(np.float32), harm=res[1].astype(np.float32),
                 uv=res[2].astype(np.float32), bre=res[3].astype(np.float32))
            names.append(name)
            print(name, req["flags"], "voiced share %.2f" % float((ins["mask"] > 0).mean()), "rms %.3f" % float(np.sqrt(np.mean(out ** 2))), flush=True)
    save("sampler_hard_index", names=np.array(names))


def gen_index_plans():
    """Loop-mode / slicing index plans, bit-exact: feed env[b,t] = t and mask[n] = n so the
    assembled arrays spell out which source frame / sample every output position came from."""
    import json
    out = {}
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        combos = []
        for mode in ("L0", "L1", "L2"):
            for length in (120, 700, 1900):
                for (off, cons, cut) in ((50, 100, 100), (0, 0, 0), (30, 60, -200), (120, 0, 50), (10, 250, 100)):
                    combos.append((mode, length, off, cons, cut, 100))
        combos += [("L0", 500, 50, 100, 100, 60), ("L1", 500, 50, 100, 100, 140), ("L2", 900, 50, 100, 100, 70),
                   ("R1", 500, 50, 100, 100, 100), ("R1L1", 900, 20, 80, -220, 100), ("L0", 1500, 50, 100, 330, 100),
                   ("L0", 800, 50, 100, 361, 100), ("L1", 800, 50, 100, 349, 100)]
        src = syn.make_source(3000, seconds=0.5)
        n = src["y_len"]
        T = 1 + n // 256
        env = np.tile(np.arange(T, dtype=np.float64)[None, :], (513, 1))
        mask = np.arange(n, dtype=np.float64)
        forms = {k: 1000.0 * k + np.arange(T, dtype=np.float64) for k in (1, 2, 3, 4)}
        for j, (fl, length, off, cons, cut, vel) in enumerate(combos):
            req = syn.make_request(3000 + j, fl, length_ms=length, offset_ms=off, consonant_ms=cons,
                                   cutoff_ms=cut, velocity=vel)
            mem = (env.copy(), np.full(n, 100.0), mask.copy(), {k: v.copy() for k, v in forms.items()}, 44100, n)
            tag = "p%02d" % j
            try:
                _, _, cap, loc = _run_sampler(src, req, 1, tmp, mem_features=mem)
            except Exception as e:
                out[tag + "_error"] = np.array(type(e).__name__)
                out[tag + "_args"] = np.array(syn.request_args(req))
                names.append(tag)
                continue
            ins, _ = cap.calls[0]
            out[tag + "_args"] = np.array(syn.request_args(req))
            out[tag + "_env_row"] = ins["env"][0].astype(np.float64)
            out[tag + "_mask"] = ins["mask"].astype(np.float64)
            out[tag + "_formants"] = _pack_formants(ins["formants"])
            out[tag + "_locals"] = np.array(json.dumps({k: (float(v) if isinstance(v, float) else int(v)) for k, v in loc.items()}))
            names.append(tag)
    out["names"] = np.array(names)
    save("index_plans", **out)


def gen_goofy_file():
    """A real .goofy written by the reference's save_features (data file, for loader parity)."""
    src = syn.make_source(88, seconds=0.12)
    path = os.path.join(HERE, "sample_features.goofy")
    gf.save_features(path, src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    env, f0, mask, forms, sr, ylen = gf.load_features(path)
    save("goofy_roundtrip", knot_vals_log=env["knot_vals_log"], hz_knots=env["hz_knots"], f0=f0, mask=mask,
         formants=_pack_formants(forms), meta=np.array([env["n_bins"], env["n_fft"], env["sr"], sr, ylen]))
    full = os.path.join(HERE, "sample_full_features.goofy")
    dense = gf.decode_env_from_knots(src["env_pack"])[:, :6]
    gf.save_features(full, dense, src["f0"][:1280], src["mask"][:1280], {"F1": [500.0] * 6, 2: [1500.0] * 6}, src["sr"], 1280)
    env2, f02, mask2, forms2, sr2, ylen2 = gf.load_features(full)
    save("goofy_roundtrip_full", env=env2, f0=f02, mask=mask2, meta=np.array([sr2, ylen2]),
         **{"formant_%d" % k: np.asarray(v, dtype=np.float64) for k, v in forms2.items()})


def gen_post_chain():
    """Time-varying one-pole cascades + jitter generators (§8 f rows 1-2)."""
    out = {}
    r = _orig_default_rng(61)
    n = 5000
    x = r.standard_normal(n).astype(np.float32)
    f0 = (200 + 50 * np.sin(np.arange(n) / 300.0)).astype(np.float32)
    f0[1000:1300] = 0
    out["x"], out["f0"] = x, f0
    for i, (cf, order, bt) in enumerate(((1.0, 6, "highpass"), (1.4, 3, "lowpass"), (200.0, 6, "highpass"), (0.5, 4, "highpass"))):
        out[f"dyn_{i}"] = ss.dynamic_butter_filter(x, f0, 44100, cf, order=order, btype=bt)
        out[f"dyn_args_{i}"] = np.array([cf, order, 1.0 if bt == "highpass" else 0.0])
    out["dyn_short_f0"] = ss.dynamic_butter_filter(x, f0[::7], 44100, 1.0, order=2, btype="lowpass")
    np.random.seed(123)
    out["f0_jitter"] = gf.apply_f0_jitter(f0, 44100, speed=100, strength=1.0)
    np.random.seed(124)
    out["vol_jitter"] = gf.create_volume_jitter(n, 44100, speed=150, strength=0.8)
    out["vol_vibrato"] = gf.create_volume_jitter(n, 44100, speed=150.0, strength=0.15, vibrato=True)
    out["sub_vibrato"] = gf.apply_subharm_vibrato(f0.astype(np.float64), 44100, vibrato_rate=75, vibrato_depth=3, vibrato_delay=0.01)
    m = (f0 > 0).astype(np.float64)
    out["subharm"] = gf.add_subharms(out["sub_vibrato"], 44100, voicing_mask=m, subharm_weight=0.75, subharm_semitones=12)
    a = r.standard_normal(200)
    out["sp_in"] = a
    out["sp_1d"] = ss.stretch_prefix_1d(a, 60, 1.3195)
    M = r.standard_normal((5, 40))
    out["sp_M"] = M
    out["sp_2d"] = ss.stretch_prefix_2d_frames(M, 11, 0.7071)
    tr = np.array([500, 510, 0, 0, 530, 90000, 540, np.nan, 520, 500, 480, 100], dtype=np.float64)
    out["san_in"] = tr
    out["san_out"] = ss.sanitize_smooth_formant(tr, 14, 44100, min_hz=120.0, sigma_frames=4)
    out["san_allbad"] = ss.sanitize_smooth_formant(np.zeros(5), 8, 44100, min_hz=300.0, sigma_frames=4)
    save("post_chain", **out)


def gen_cold_cache():
    """gf.extract_features end to end (GOOFER.py:940-969) with a FAKE parselmouth: the Praat half of the analysis is
    third-party arithmetic (parity unpinned), but everything the reference does around it is its own — which calls it makes
    with which arguments, nan_to_num + fix_f0_gaps, the per-sample interpolation / clip / voicing threshold, the formant
    dict padded / trimmed to the STFT frame count, the envelope and its knots.  The fake returns deterministic tracks (with
    unvoiced gaps of 1-3 frames, a NaN, leading / trailing zeros, formant values that are None or raise) and records the
    arguments it was called with; the product's cold-cache path is tested with a stub tracker returning the same tracks."""
    calls = {}

    class _Pitch:
        def __init__(self, freq):
            self.selected_array = {"frequency": freq}

    class _Formant:
        def __init__(self, n, step, t0):
            self.n, self.step, self.t0 = n, step, t0

        def get_number_of_frames(self):
            return self.n

        def get_time_from_frame_number(self, k):                  # 1-based, like Praat
            return self.t0 + (k - 1) * self.step

        def get_value_at_time(self, num, t):
            k = int(round((t - self.t0) / self.step))
            if num == 5 and k % 7 == 3:
                raise RuntimeError("undefined")
            if num == 4 and k % 5 == 0:
                return None
            return float(num * 650.0 + 40.0 * np.sin(0.3 * k + num))

    class _Sound:
        class ToPitchMethod:
            AC = "AC"

        def __init__(self, y, sr):
            self.y, self.sr = np.asarray(y), sr

        def to_pitch(self, **kw):
            calls["to_pitch"] = dict(kw)
            n = int(len(self.y) / self.sr / kw["time_step"]) - 2          # Praat's frame count differs from the STFT's
            k = np.arange(n)
            f = 220.0 + 30.0 * np.sin(0.21 * k)
            f[:2] = 0.0                                                    # leading zeros: no left neighbour, stay zero
            if n > 40:
                f[8] = 0.0                                                 # gaps of 1, 2 (bridged at max_gap 2) and 3 (kept)
                f[14:16] = 0.0
                f[22:25] = 0.0
                f[30] = np.nan                                             # nan_to_num -> 0 -> a one-frame gap
                f[-3:] = 0.0                                               # trailing zeros: no right neighbour
            else:
                f[4] = 0.0
            return _Pitch(f)

        def to_formant_burg(self, **kw):
            calls["to_formant_burg"] = dict(kw)
            n = int(len(self.y) / self.sr / kw["time_step"]) + 3           # more frames than the STFT: trimmed
            return _Formant(n, kw["time_step"], 0.5 * kw["time_step"])

    fake = types.ModuleType("parselmouth")
    fake.Sound = _Sound
    real = gf.parselmouth
    gf.parselmouth = fake
    try:
        r = _orig_default_rng(2024)
        sr, n = 44100, 11700
        t = np.arange(n) / sr
        y = (0.4 * np.sin(2 * np.pi * 220.0 * t) * (1 + 0.3 * np.sin(2 * np.pi * 3.0 * t)) + 0.05 * r.standard_normal(n))
        env, f0i, vmask, forms, knots = gf.extract_features(y, sr)
        out = {"y": y, "sr": np.array([sr]), "env_spec": env, "f0_interp": f0i, "voicing_mask": vmask,
               "knot_vals_log": knots["knot_vals_log"], "hz_knots": knots["hz_knots"],
               "knots_meta": np.array([knots["n_bins"], knots["n_fft"], knots["sr"]]),
               "pitch_track": _Sound(y, sr).to_pitch(**calls["to_pitch"]).selected_array["frequency"],
               "pitch_kw_names": np.array(sorted(k for k in calls["to_pitch"] if k != "method")),
               "pitch_kw_vals": np.array([float(calls["to_pitch"][k]) for k in sorted(calls["to_pitch"]) if k != "method"]),
               "pitch_method": np.array([str(calls["to_pitch"]["method"])]),
               "formant_kw_names": np.array(sorted(calls["to_formant_burg"])),
               "formant_kw_vals": np.array([float(calls["to_formant_burg"][k]) for k in sorted(calls["to_formant_burg"])])}
        for k, v in forms.items():
            out["formant_%d" % k] = np.asarray(v, dtype=np.float64)
        # a short second case: fewer pitch frames than three, a track that is all zeros after the gap repair
        y2 = y[:3000]
        env2, f0i2, vmask2, forms2, _ = gf.extract_features(y2, sr)
        out.update(y2_f0_interp=f0i2, y2_voicing_mask=vmask2, y2_formant_1=np.asarray(forms2[1], dtype=np.float64),
                   y2_pitch_track=_Sound(y2, sr).to_pitch(**calls["to_pitch"]).selected_array["frequency"])
        # fix_f0_gaps on its own, other max_gap values
        tr = np.array([0, 0, 200, 0, 210, 0, 0, 230, 0, 0, 0, 260, 0, 0, 0, 0, 300, 0], dtype=np.float64)
        out["gaps_in"] = tr
        for g in (0, 1, 2, 4):
            out["gaps_out_%d" % g] = gf.fix_f0_gaps(tr, g)
        save("cold_cache", **out)
    finally:
        gf.parselmouth = real


if __name__ == "__main__" and len(sys.argv) > 1:
    for which in sys.argv[1:]:                         # regenerate only the named groups, e.g. `make_golden.py sampler_combos`
        globals()["gen_" + which]()
elif __name__ == "__main__":
    gen_tables()
    gen_stft_istft()
    gen_pulse()
    gen_pulse_lf()
    gen_gauss()
    gen_mask_interp()
    gen_knots()
    gen_warps()
    gen_synthesize()
    gen_synthesize_rough()
    gen_flags_pitch()
    gen_goofy_file()
    gen_post_chain()
    gen_cold_cache()
    gen_index_plans()
    gen_sampler()
    gen_sampler_combos()
    gen_sampler_hard()
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("total fixture bytes: %.1f MB" % (total / 1e6))
