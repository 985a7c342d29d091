"""CPU: the PRODUCT's host code (goofer_amd.sampler / goofer_amd.core — not the oracle) against the reference's golden
vectors: flag / pitch-string decode, flag scaling, segment indices and the loop-mode index plans (bit-exact integer path,
SillySampler.py:50-93, 286-411, 453-500, 625-763), and the .goofy reader / writer (GOOFER.py:287-339).
The device half of the index-plan check (the assembled probe arrays) is tests/test_gpu_sampler.py::test_index_plans_on_device."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from goofer_amd import core
from goofer_amd import sampler as S
from goofer_amd import synthetic as syn


def test_product_flag_and_pitch_decode_bit_exact():
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
    with pytest.raises(ValueError):
        S.note_to_midi("H4")
    with pytest.raises(ValueError):
        S.split_arguments("only one.wav 1 2 3 4 5 6 7 8 9 10 11")


def test_product_flag_scaling_matches_reference():
    g = golden("flags_pitch")
    for fs, want in zip(g["flag_strings"], g["params"]):
        want = json.loads(str(want))
        args = ("C4", "100", str(fs), "50", "1000", "100", "-250", "80", "0", "!125", "AA")
        if "error" in want:
            with pytest.raises(Exception) as ei:
                S.decode_request(*args)
            assert type(ei.value).__name__ == want["error"], str(fs)
            continue
        p = S.decode_request(*args)
        got = {
            "formant_shift": p.formant_shift, "brightness_env": p.brightness_env,
            "F1_shift": p.f_shift[0], "F2_shift": p.f_shift[1], "F3_shift": p.f_shift[2], "F4_shift": p.f_shift[3],
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


def probe_source():
    """The probe of tests/golden/make_golden.py:gen_index_plans: env[b, t] = t, mask[n] = n, formant k = 1000 k + t."""
    src = syn.make_source(3000, seconds=0.5)
    n = src["y_len"]
    T = 1 + n // 256
    env = np.tile(np.arange(T, dtype=np.float64)[None, :], (513, 1))
    forms = {k: 1000.0 * k + np.arange(T, dtype=np.float64) for k in (1, 2, 3, 4)}
    return env, np.full(n, 100.0), np.arange(n, dtype=np.float64), forms, 44100, n, T


def test_product_index_plans_match_reference():
    """All 53 reference index plans through the product's planner: cut points and tail lengths exactly, the error cases by
    exception type, and the frame plan evaluated on the probe (env[b, t] = t): every output frame is the weighted sum of
    source frame numbers the reference produced (copies exactly; lerps / cross-fades to 1e-9 of the frame number)."""
    g = golden("index_plans")
    env, f0, mask, forms, sr, n, T = probe_source()
    assert len(g["names"]) == 53
    for tag in g["names"]:
        args = [str(a) for a in g[f"{tag}_args"]]
        req = S.decode_request(*args)
        if f"{tag}_error" in g.files:
            with pytest.raises(Exception) as ei:
                S.plan_note(req, sr, n, T, forms)
            assert type(ei.value).__name__ == str(g[f"{tag}_error"]), tag
            continue
        p = S.plan_note(req, sr, n, T, forms)
        loc = json.loads(str(g[f"{tag}_locals"]))
        for k, v in p.seg.items():
            assert loc[k] == v, (tag, k)
        assert loc["desired_tail_frames"] == int(np.ceil(req.length * sr / S.HOP)), tag
        assert loc["desired_tail_samples"] == p.want_samples, tag
        want_row = g[f"{tag}_env_row"]
        assert p.tap_idx.shape[0] == len(want_row), tag
        src_rows = np.arange(T, dtype=np.float64)[::-1] if req.reverse else np.arange(T, dtype=np.float64)
        live = p.tap_w != 0.0
        idx = np.where(live, p.tap_idx, 0)
        assert idx.min() >= 0 and idx.max() < T, tag
        got_row = np.sum(np.where(live, p.tap_w * src_rows[idx], 0.0), axis=1)
        copies = np.sum(live, axis=1) == 1
        assert np.array_equal(got_row[copies], want_row[copies]), tag
        assert np.max(np.abs(got_row - want_row), initial=0.0) <= 1e-9 * max(1.0, float(np.max(want_row, initial=0.0))), tag
        assert p.n_out == len(g[f"{tag}_mask"]), tag
        # formant tracks as gf.synthesize uses them: the planner already applies its pad_trim_to_len (edge-pad / truncate to
        # the envelope's frame count, GOOFER.py:64-70, 999-1000) to what the reference hands over
        want_f = g[f"{tag}_formants"]
        T_env = p.tap_idx.shape[0]
        assert p.formants.shape == (T_env, 4), tag
        Lf = want_f.shape[1]
        want_fit = want_f[:, :T_env] if Lf >= T_env else np.pad(want_f, ((0, 0), (0, T_env - Lf)), mode="edge")
        assert np.array_equal(p.formants.T, want_fit), tag


def test_product_goofy_io_matches_reference_files(tmp_path):
    """goofer_amd.core.load_features / save_features on the files the reference itself wrote (GOOFER.py:287-339): same arrays,
    same dtypes, same member list — knots mode and dense ('full') mode."""
    g = golden("goofy_roundtrip")
    ref_path = os.path.join(GOLDEN, "sample_features.goofy")
    env, f0, mask, forms, sr, ylen = core.load_features(ref_path)
    assert env["mode"] == "knots"
    assert np.array_equal(env["knot_vals_log"], g["knot_vals_log"]) and env["knot_vals_log"].dtype == np.float16
    assert np.array_equal(env["hz_knots"], g["hz_knots"]) and env["hz_knots"].dtype == np.float32
    assert np.array_equal(f0, g["f0"]) and f0.dtype == g["f0"].dtype
    assert np.array_equal(mask, g["mask"]) and mask.dtype == g["mask"].dtype
    assert [env["n_bins"], env["n_fft"], env["sr"], sr, ylen] == [int(v) for v in g["meta"]]
    assert sorted(forms) == [1, 2, 3, 4]
    for i in range(4):
        assert np.array_equal(forms[i + 1], g["formants"][i])
    out = tmp_path / "x_features.goofy"
    core.save_features(out, env, f0, mask, forms, sr, ylen)
    a = np.load(out, allow_pickle=True)
    b = np.load(ref_path, allow_pickle=True)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k == "formants":
            fa, fb_ = a[k].item(), b[k].item()
            assert sorted(fa) == sorted(fb_)
            for kk in fa:
                assert np.array_equal(np.asarray(fa[kk]), np.asarray(fb_[kk])), kk
            continue
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    # dense mode
    g2 = golden("goofy_roundtrip_full")
    full = os.path.join(GOLDEN, "sample_full_features.goofy")
    env2, f02, m2, forms2, sr2, ylen2 = core.load_features(full)
    assert np.array_equal(env2, g2["env"]) and env2.dtype == g2["env"].dtype
    assert np.array_equal(f02, g2["f0"]) and np.array_equal(m2, g2["mask"])
    assert [sr2, ylen2] == [int(v) for v in g2["meta"]]
    for k in (1, 2, 3, 4):
        assert np.array_equal(forms2[k], g2["formant_%d" % k])
    out2 = tmp_path / "y_features.goofy"
    core.save_features(out2, env2, f02, m2, forms2, sr2, ylen2)
    a2, b2 = np.load(out2, allow_pickle=True), np.load(full, allow_pickle=True)
    assert sorted(a2.files) == sorted(b2.files)
    for k in a2.files:
        if k != "formants":
            assert a2[k].dtype == b2[k].dtype and np.array_equal(a2[k], b2[k]), k
