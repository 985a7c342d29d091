"""CPU: the oracle's sampler restatement against the reference's golden vectors."""
import json

import numpy as np
import pytest

from conftest import golden, rms_err
from goofer_amd import synthetic as syn
from oracle import goofer_ref as G
from oracle import sampler_ref as S

CASES = [str(n) for n in golden("sampler_index")["names"]]


def test_flag_and_pitch_decode_bit_exact():
    g = golden("flags_pitch")
    for fs, want in zip(g["flag_strings"], g["parsed"]):
        assert S.parse_flags(str(fs)) == json.loads(str(want))
    for i, ps in enumerate(g["pitch_strings"]):
        got = S.pitch_string_to_cents(str(ps))
        assert got.dtype == np.float32 and np.array_equal(got, g[f"cents_{i}"])
    assert [S.note_to_midi(str(n)) for n in g["notes"]] == list(g["midi"])
    assert np.array_equal([S.midi_to_hz(m) for m in (0, 57, 69, 69.5, 127)], g["hz"])
    assert S.split_arguments(str(g["split_in"][0])) == [str(v) for v in g["split_out"]]
    assert np.array_equal(S.pitch_string_to_cents(syn.encode_cents(range(-2048, 2048, 37))), np.arange(-2048, 2048, 37))


def test_flag_scaling_matches_reference():
    g = golden("flags_pitch")
    for fs, want in zip(g["flag_strings"], g["params"]):
        want = json.loads(str(want))
        if "error" in want:
            with pytest.raises(Exception) as ei:
                S.decode_request("C4", "100", str(fs), "50", "1000", "100", "-250", "80", "0", "!125", "AA")
            assert type(ei.value).__name__ == want["error"]
            continue
        p = S.decode_request("C4", "100", str(fs), "50", "1000", "100", "-250", "80", "0", "!125", "AA")
        got = {
            "formant_shift": p.formant_shift, "brightness_env": p.brightness_env,
            "F1_shift": p.F_shift[0], "F2_shift": p.F_shift[1], "F3_shift": p.F_shift[2], "F4_shift": p.F_shift[3],
            "f0_jitter": p.f0_jitter, "f0_jitter_strength": p.f0_jitter_strength, "volume_jitter": p.volume_jitter,
            "volume_jitter_strength": p.volume_jitter_strength, "sd_strength": p.sd_strength,
            "breathiness_mix": p.breathiness_mix, "unvoiced_mix": p.unvoiced_mix, "harmonic_mix": p.harmonic_mix,
            "loop_mode": p.loop_mode, "tension": p.tension, "subharm_weight": p.subharm_weight,
            "add_subharm": p.add_subharm, "reverse": p.reverse, "growl_mix": p.growl_mix,
            "aperiodic_mix": p.aperiodic_mix, "subharm_gain": p.subharm_gain, "normalize": p.normalize,
            "env_shape": p.env_shape, "force_voiced": p.force_voiced, "pitch_dyn": p.pitch_dyn,
            "formant_width": p.formant_width, "formant_strength_f1": p.formant_strength[0],
            "formant_strength_f2": p.formant_strength[1], "formant_strength_f3": p.formant_strength[2],
            "formant_strength_f4": p.formant_strength[3], "use_editor": p.use_editor, "offset": p.offset,
            "length": p.length, "consonant": p.consonant, "cutoff": p.cutoff, "volume": p.volume,
            "tempo": p.tempo, "velocity": p.velocity, "pitch_m": p.pitch_m,
        }
        for k, v in want.items():
            gv = got[k]
            assert (gv == v) if isinstance(v, (str, bool)) else (float(gv) == v), (str(fs), k, gv, v)


def test_post_chain_primitives():
    g = golden("post_chain")
    x, f0 = g["x"], g["f0"]
    for i in range(4):
        cf, order, hp = g[f"dyn_args_{i}"]
        got = S.dynamic_filter(x, f0, 44100, float(cf), order=int(order), btype="highpass" if hp else "lowpass")
        assert got.dtype == np.float32
        # the stub-imported reference forms alpha in fp32 (numba: fp64) -> last-bit differences
        assert rms_err(got, g[f"dyn_{i}"]) < 2e-6 * max(1.0, np.abs(g[f"dyn_{i}"]).max()), i
    assert rms_err(S.dynamic_filter(x, f0[::7], 44100, 1.0, order=2, btype="lowpass"), g["dyn_short_f0"]) < 2e-6
    np.random.seed(123)
    np.testing.assert_allclose(G.f0_jitter_curve(len(f0), 44100, 100, 1.0), g["f0_jitter"], rtol=1e-11)
    np.random.seed(124)
    np.testing.assert_allclose(G.volume_jitter_curve(len(x), 44100, 150, 0.8), g["vol_jitter"], rtol=1e-11)
    assert np.array_equal(G.volume_jitter_curve(len(x), 44100, 150.0, 0.15, vibrato=True), g["vol_vibrato"])
    sv = G.subharm_vibrato(f0.astype(np.float64), 44100, 75, 3, 0.01)
    assert np.array_equal(sv, g["sub_vibrato"])
    sub = G.subharm_layer(sv, 44100, weight=0.75, semitones=12, vmask=(f0 > 0).astype(np.float64))
    np.testing.assert_allclose(sub, g["subharm"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(S.stretch_prefix_1d(g["sp_in"], 60, 1.3195), g["sp_1d"])
    assert np.array_equal(S.stretch_prefix_2d(g["sp_M"], 11, 0.7071), g["sp_2d"])
    np.testing.assert_allclose(S.sanitize_formant(g["san_in"], 14, 44100, min_hz=120.0, sigma_frames=4), g["san_out"], rtol=1e-6)
    assert np.array_equal(S.sanitize_formant(np.zeros(5), 8, 44100, min_hz=300.0, sigma_frames=4), g["san_allbad"])


def _features(src):
    return (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
            src["sr"], src["y_len"])


def test_index_plans_bit_exact():
    """Loop modes / slicing / reverse / velocity: env[b,t]=t and mask[n]=n reveal the source index of
    every assembled frame and sample; must equal the reference exactly (integer path)."""
    g = golden("index_plans")
    src = syn.make_source(3000, seconds=0.5)
    n = src["y_len"]
    T = 1 + n // 256
    for tag in g["names"]:
        args = [str(a) for a in g[f"{tag}_args"]]
        p = S.decode_request(*args)
        env = np.tile(np.arange(T, dtype=np.float64)[None, :], (513, 1))
        forms = {k: 1000.0 * k + np.arange(T, dtype=np.float64) for k in (1, 2, 3, 4)}
        feats = (env, np.full(n, 100.0), np.arange(n, dtype=np.float64), forms, 44100, n)
        if f"{tag}_error" in g.files:
            with pytest.raises(Exception) as ei:
                S.assemble(feats, p)
            assert type(ei.value).__name__ == str(g[f"{tag}_error"])
            continue
        a = S.assemble(feats, p)
        loc = json.loads(str(g[f"{tag}_locals"]))
        for k, v in a["seg"].items():
            assert loc[k] == v, (tag, k)
        assert loc["desired_tail_frames"] == a["want_frames"] and loc["desired_tail_samples"] == a["want_samples"]
        assert np.array_equal(a["env"][0].astype(np.float64), g[f"{tag}_env_row"]), tag
        assert np.array_equal(np.asarray(a["mask"], dtype=np.float64), g[f"{tag}_mask"]), tag
        F = np.stack([np.asarray(a["formants"][k], dtype=np.float64) for k in sorted(a["formants"], key=str)], 0)
        assert np.array_equal(F, g[f"{tag}_formants"]), tag


def test_concat_loop_of_short_tails_matches_numpy_semantics():
    """The L0 (concat) loop for every short tail, including the one-frame tail whose fades are empty: under numpy's slicing /
    broadcasting rules the reference then simply repeats the frame (found by the extreme-request soak).  The planner's
    index-level plan must realise the oracle's array-level loop for tails of 1..20 frames."""
    from goofer_amd import sampler as P
    from oracle import sampler_ref as SR
    rng = np.random.default_rng(4)
    for n in range(1, 21):
        tail = rng.random((5, n))
        for want in (n, n + 1, 2 * n, 2 * n + 1, 3 * n + n // 2, 7 * n + 3, 61):
            if want < n:
                continue
            ref = SR._loop_env_concat(tail, want) if want > n else tail[:, :want]
            taps = P._loop_frames(P.Taps.copy(np.arange(n)), want, "concat")
            got = taps.w[:, 0][None, :] * tail[:, taps.idx[:, 0]] + taps.w[:, 1][None, :] * tail[:, taps.idx[:, 1]]
            assert got.shape == ref.shape, (n, want)
            assert np.allclose(got, ref, rtol=0, atol=1e-15), (n, want)


@pytest.mark.parametrize("name", CASES)
def test_render_against_reference(name):
    g = golden("sampler_" + name)
    i = CASES.index(name)
    seed, legacy, src_seed = (int(v) for v in g["seed"])
    assert src_seed == 2000 + i
    src = syn.make_source(src_seed, seconds=0.45)
    args = [str(a) for a in g["args"]]
    p = S.decode_request(*args)
    np.random.seed(legacy)
    out, a, stems = S.render(_features(src), p, seed=seed, return_parts=True)
    ref = g["out"]
    assert out.shape == ref.shape
    if "env_new" in g.files:
        assert a["env"].shape == g["env_new"].shape and a["env"].dtype == g["env_new"].dtype
        np.testing.assert_allclose(a["env"], g["env_new"], rtol=1e-6, atol=1e-12)
        assert np.array_equal(a["f0"], g["f0_new"]) and np.array_equal(a["mask"], g["mask_new"])
        F = np.stack([np.asarray(a["formants"][k], dtype=np.float64) for k in sorted(a["formants"], key=str)], 0)
        np.testing.assert_allclose(F, g["formants_new"], rtol=1e-6)
        for got, key in zip(stems, ("harm", "uv", "bre")):
            assert rms_err(got, g[key]) < 3e-6, (name, key)
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert rms_err(out, ref) < 1e-5 * scale, (name, rms_err(out, ref), scale)


COMBOS = [str(n) for n in golden("combo_index")["names"]]


def _combo(name):
    """The source of a flag-interaction fixture, rebuilt with the generator's own draws (make_golden.gen_sampler_combos)."""
    g = golden(name)
    i = COMBOS.index(name)
    rng = np.random.default_rng(7700 + i)
    flags = syn.random_flags(rng)
    src = syn.make_source(5000 + i, seconds=float(rng.uniform(0.3, 0.55)))
    args = [str(a) for a in g["args"]]
    assert args[2] == flags and abs(src["y_len"] / src["sr"] - float(g["seconds"][0])) < 1e-9
    return g, src, args


@pytest.mark.parametrize("name", COMBOS)
def test_flag_combinations_against_reference(name):
    """Random subsets of the whole flag vocabulary rendered by the reference itself: the oracle must follow it through
    every interaction (assembly edits + jitter / sub-harmonic layers + post chain in one note)."""
    g, src, args = _combo(name)
    seed, legacy, _ = (int(v) for v in g["seed"])
    np.random.seed(legacy)
    out = S.render(_features(src), S.decode_request(*args), seed=seed)
    ref = g["out"]
    assert out.shape == ref.shape
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert rms_err(out, ref) < 1e-5 * scale, (name, args[2], rms_err(out, ref), scale)


HARD = [str(n) for n in golden("sampler_hard_index")["names"]]


@pytest.mark.parametrize("name", HARD)
def test_hard_sources_against_reference(name):
    """Hard sources (synthetic.make_hard_source: 3-6 interior V/UV transitions with fp16 ramps and fractional plateaus, formant
    frames that are 0 / NaN / negative / above Nyquist, crossing and all-invalid tracks, 40 dB envelope jumps, near-zero bins)
    rendered by the reference itself: the oracle follows it through the full sampler path — assembled mask and f0 exactly, the
    stems and the note to 1e-5 (SillySampler.py:242-283, GOOFER.py:556-569, 849-873, 1131-1144)."""
    g = golden(name)
    src, req = syn.hard_case(HARD.index(name))
    args = [str(a) for a in g["args"]]
    assert args == [str(a) for a in syn.request_args(req)] and abs(src["y_len"] / src["sr"] - float(g["seconds"][0])) < 1e-9
    seed, legacy, _ = (int(v) for v in g["seed"])
    np.random.seed(legacy)
    out = S.render(_features(src), S.decode_request(*args), seed=seed)
    ref = g["out"]
    assert out.shape == ref.shape and np.isfinite(out).all()
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert rms_err(out, ref) < 1e-5 * scale, (name, args[2], rms_err(out, ref), scale)
