"""GPU: the 13-argument resampler path (plan on host, assemble + synthesize on the device) against the
reference's renders of the same (features, flags, pitch string) with the same injected phases."""
import os

import numpy as np
import pytest

from conftest import golden, rms_err
from goofer_amd import synthetic as syn

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CASES = [str(n) for n in golden("sampler_index")["names"]]
SUPPORTED = ["default", "t12g50", "tm12gm50", "formants", "formants_flip", "L0", "L1", "L2", "L0_short", "br_es_neg",
             "br_es_pos", "vel60", "vel150", "R1", "FV1_P50", "negcut", "vol_mix"]
TOL = 1e-4


@pytest.fixture(scope="module")
def renderer():
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer
    c = Context(0)
    yield Renderer(c)
    c.close()


def _job(name):
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    g = golden("sampler_" + name)
    i = CASES.index(name)
    src = syn.make_source(2000 + i, seconds=0.45)
    source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    req = S.decode_request(*[str(a) for a in g["args"]])
    return g, source, req


@pytest.mark.parametrize("name", SUPPORTED)
def test_note_matches_reference(renderer, name):
    g, source, req = _job(name)
    (out,), parts = renderer.render([(source, req)], phi_seeds=[int(g["seed"][0])], return_parts=True)
    ref = g["out"]
    assert out.shape == ref.shape
    if "env_new" in g.files:
        env = parts["env"].cpu().numpy().T
        ref_env = np.asarray(g["env_new"], dtype=np.float32)
        assert env.shape == ref_env.shape
        assert np.max(np.abs(env - ref_env)) <= 3e-6 * np.abs(ref_env).max(), name
        assert np.array_equal(parts["mask"].cpu().numpy(), np.asarray(g["mask_new"], dtype=np.float32))
        f0 = parts["f0"].cpu().numpy()
        np.testing.assert_allclose(f0, np.asarray(g["f0_new"], dtype=np.float32), rtol=3e-7, atol=1e-6)
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, (name, e)
    assert e < 2e-5, (name, e)


def test_all_supported_notes_as_one_batch(renderer):
    jobs, refs, seeds = [], [], []
    for name in SUPPORTED:
        g, source, req = _job(name)
        jobs.append((source, req))
        refs.append(g["out"])
        seeds.append(int(g["seed"][0]))
    outs = renderer.render(jobs, phi_seeds=seeds)
    worst = 0.0
    for name, o, r in zip(SUPPORTED, outs, refs):
        assert o.shape == r.shape, name
        e = rms_err(o, r) / max(1.0, float(np.max(np.abs(r))))
        worst = max(worst, e)
        assert e < TOL, (name, e)
    print("worst sample-RMS over the 17-note batch vs reference:", worst)


POST = ["su50", "sj30", "sa30", "st50", "stm50", "sd30", "vf40", "vfm40", "pd50", "pdm50"]


@pytest.mark.parametrize("name", POST)
def test_post_chain_flags_match_reference(renderer, name):
    """su / sj / sa layers, st tension, sd dryness, vf fry, pd pitch dynamics (SillySampler.py:857-997, 1037-1182)."""
    g, source, req = _job(name)
    np.random.seed(int(g["seed"][1]))
    (out,) = renderer.render([(source, req)], phi_seeds=[int(g["seed"][0])])
    ref = g["out"]
    assert out.shape == ref.shape
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, e


def test_post_chain_mixed_batch(renderer):
    """Every post flag in one ragged batch next to plain notes: the extra synth calls run on sub-batches."""
    names = ["default", "su50", "sj30", "L1", "sa30", "stm50", "vfm40", "pd50", "st50", "sd30"]
    jobs, seeds, refs = [], [], []
    for nm in names:
        g, source, req = _job(nm)
        jobs.append((source, req))
        seeds.append(int(g["seed"][0]))
        refs.append(g["out"])
    outs = renderer.render(jobs, phi_seeds=seeds)
    for nm, o, r in zip(names, outs, refs):
        assert o.shape == r.shape, nm
        assert rms_err(o, r) / max(1.0, float(np.max(np.abs(r)))) < TOL, nm


def test_resampler_call_surface(renderer, tmp_path):
    """GooferResampler(in.wav, out.wav, ...13 args): reads <stem>_features.goofy, writes a PCM16 wav."""
    import wave
    from goofer_amd import core
    from goofer_amd.render import GooferResampler
    src = syn.make_source(2000, seconds=0.45)
    wav = tmp_path / "a b" / "src.wav"
    wav.parent.mkdir()
    core.save_features(wav.with_name("src_features.goofy"), src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"],
                       src["y_len"])
    req = syn.make_request(2000, "t0g0", length_ms=300)
    out = tmp_path / "out.wav"
    r = GooferResampler(str(wav), str(out), *syn.request_args(req), renderer=renderer, seed=1)
    with wave.open(str(out), "rb") as w:
        assert w.getframerate() == 44100 and w.getnchannels() == 1 and w.getsampwidth() == 2
        n = w.getnframes()
    assert n == len(r.out) == len(golden("sampler_default")["out"])
    with pytest.raises(FileNotFoundError):
        GooferResampler(str(tmp_path / "none.wav"), str(out), *syn.request_args(req), renderer=renderer)


def _config4_ids():
    """Notes of the 10 000-note job that cover every loop mode (the flag cycles L0 / L1 / L2 with the note id) in every decile
    of the job's length distribution (log-uniform 100 ms .. 3 s): the first id of each of the 30 cells, plus the first five."""
    frames = np.array([syn.config_note_frames(4, i) for i in range(10000)])
    edges = np.quantile(frames, np.linspace(0.0, 1.0, 11))
    ids = [0, 1, 2, 5, 7]
    for mode in range(3):
        for d in range(10):
            cell = np.nonzero((np.arange(10000) % 3 == mode) & (frames >= edges[d]) & (frames <= edges[d + 1]))[0]
            ids.append(int(cell[0]))
    return sorted(set(ids))


@pytest.mark.parametrize("config,ids", [(1, [0]), (4, _config4_ids()), (5, [0, 1, 2, 3, 257, 512, 777, 1023]), (2, [0, 1, 2, 3, 100, 255]),
                                        (3, sorted(set(list(range(0, 1024, 33)) + [257, 640, 1023])))])
def test_baseline_config_notes_vs_oracle(config, ids):
    """Notes of the BASELINE configs (config 1: the single 1 s note with default flags; 96 kHz / n_fft 2048 / hop 96 with
    br+es at 8 ids over the batch; config 4's L0/L1/L2 loops at 30+ ids covering every loop mode in every length decile of
    the 10 000-note job; config 3 at 34 ids spread over the 1024-note batch) rendered as one batch vs the CPU oracle's full
    render, same injected phases."""
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    geo = syn.config_geometry(config)
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=geo["hop"])
        jobs, refs, seeds = [], [], []
        for i in ids:
            src, req, phi_seed = syn.config_note(config, i)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
            feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                     src["sr"], src["y_len"])
            refs.append(SR.render(feats, SR.decode_request(*syn.request_args(req)), seed=phi_seed, n_fft=geo["n_fft"], hop=geo["hop"]))
            seeds.append(phi_seed)
        outs = r.render(jobs, phi_seeds=seeds)
        for i, o, ref in zip(ids, outs, refs):
            assert o.shape == ref.shape
            e = rms_err(o, ref) / max(1.0, float(np.max(np.abs(ref))))
            assert e < 2e-5, (config, i, e)
    finally:
        ctx.close()


@pytest.mark.parametrize("config,ids", [(3, range(1024)), (4, range(0, 10000, 10)), (5, range(0, 1024, 16))])
def test_whole_batches_of_the_baseline_configs_vs_oracle(config, ids):
    """All 1024 notes of BASELINE config 3 — the batch the headline number is quoted on —, every tenth note of config 4's 10 000-note
    job and every sixteenth of config 5's batch, each set rendered as ONE device batch against the CPU oracle's render of each
    note, same injected phases: every note within the 2e-5 bound (worst and mean printed).  ~20 s of oracle time per set on one core."""
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    geo = syn.config_geometry(config)
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=geo["hop"])
        jobs, refs, seeds = [], [], []
        for i in ids:
            src, req, phi_seed = syn.config_note(config, i)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
            feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                     src["sr"], src["y_len"])
            refs.append(SR.render(feats, SR.decode_request(*syn.request_args(req)), seed=phi_seed, n_fft=geo["n_fft"], hop=geo["hop"]))
            seeds.append(phi_seed)
        outs = r.render(jobs, phi_seeds=seeds)
        errs = []
        for i, (o, ref) in enumerate(zip(outs, refs)):
            assert o.shape == ref.shape, i
            errs.append(rms_err(o, ref) / max(1.0, float(np.max(np.abs(ref)))))
        errs = np.asarray(errs)
        assert errs.max() < 2e-5, (config, int(errs.argmax()), float(errs.max()), float(errs.mean()))
        print("config %d, %d notes vs oracle: worst %.3g (position %d), mean %.3g" % (config, errs.size, errs.max(), int(errs.argmax()), errs.mean()))
    finally:
        ctx.close()


@pytest.mark.parametrize("config", [3, 4])
def test_render_batch_equals_separate_calls(config):
    """goofer_render_batch (pulse chain forked as soon as the assembled f0 exists) against goofer_assemble_batch followed by
    goofer_synth_batch, bit for bit, twice in a row (the second run re-uses the scratch arena the first one left)."""
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    ctx = Context(0)
    try:
        wl = SamplerWorkload(ctx, config, list(range(24)))
        r = wl.renderer
        ref = r.run(wl.prep, seed=3, split=True)["mix"].clone()
        torch.cuda.synchronize()
        for _ in range(2):
            got = r.run(wl.prep, seed=3)["mix"]
            torch.cuda.synchronize()
            assert torch.equal(got, ref)
        ctx.set_option("overlap", 0)
        got = r.run(wl.prep, seed=3)["mix"]
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        assert float(ref.abs().max()) > 0
    finally:
        ctx.close()


def test_render_batch_alternating_batches_stay_exact():
    """Two resident batches of different shapes rendered alternately through goofer_render_batch on one handle (the scratch
    arena, the side stream and its events are shared): every render equals the batch's own split-call result, bit for bit."""
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    ctx = Context(0)
    try:
        a = SamplerWorkload(ctx, 3, list(range(40)))
        b = SamplerWorkload(ctx, 4, list(range(100, 117)))
        ref_a = a.renderer.run(a.prep, seed=1, split=True)["mix"].clone()
        ref_b = b.renderer.run(b.prep, seed=2, split=True)["mix"].clone()
        torch.cuda.synchronize()
        outs = []
        for k in range(12):                                   # no synchronisation in between: stream order alone
            wl, sd = (a, 1) if k % 3 != 1 else (b, 2)
            outs.append((wl is a, wl.renderer.run(wl.prep, seed=sd)["mix"]))
        torch.cuda.synchronize()
        for is_a, o in outs:
            assert torch.equal(o, ref_a if is_a else ref_b)
    finally:
        ctx.close()


def test_jitter_flags_sh_sr_match_reference(renderer):
    """sh (f0 jitter) + sr (volume jitter): legacy-RNG draws seeded like the reference run."""
    g, source, req = _job("sh50sr50")
    seed, legacy = int(g["seed"][0]), int(g["seed"][1])
    np.random.seed(legacy)
    (out,) = renderer.render([(source, req)], phi_seeds=[seed])
    ref = g["out"]
    assert out.shape == ref.shape
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, e
    assert e < 2e-5, e


def test_subharm_flag_sg_matches_reference(renderer):
    """sg: vibrato'd sub-harmonic LF pulse layer added to the pulse train (SillySampler.py:364-366, GOOFER.py:1076-1097)."""
    g, source, req = _job("sg50")
    assert req.add_subharm and abs(req.subharm_weight - 0.75) < 1e-12
    (out,) = renderer.render([(source, req)], phi_seeds=[int(g["seed"][0])])
    ref = g["out"]
    assert out.shape == ref.shape
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, e


def test_subharm_mixed_batch(renderer):
    """A note with sg next to notes without it: the layer touches only its own note."""
    g, source, req = _job("sg50")
    g0, source0, req0 = _job("default")
    seeds = [int(g0["seed"][0]), int(g["seed"][0]), int(g0["seed"][0])]
    outs = renderer.render([(source0, req0), (source, req), (source0, req0)], phi_seeds=seeds)
    for o, r in zip(outs, (g0["out"], g["out"], g0["out"])):
        assert rms_err(o, r) / max(1.0, float(np.max(np.abs(r)))) < TOL


def test_dense_feature_source_vs_oracle(renderer):
    """'full' mode .goofy (dense fp16 envelope, GOOFER.py:306-333) next to a knots-mode source in one batch."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from oracle import goofer_ref as R, sampler_ref as SR
    jobs, refs, seeds = [], [], []
    for i, flags in enumerate(("br30es40fw20", "g-30L1")):
        src = syn.make_source(3100 + i, seconds=0.4)
        env16 = R.decode_env_from_knots(src["env_pack"]).astype(np.float16)
        args = ("D4", "100", flags, "20", "450", "60", "40", "100", "0", "!120", "AA#3#AF")
        jobs.append((Source.from_pack(env16, src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*args)))
        feats = (env16.astype(np.float32), src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                 src["sr"], src["y_len"])
        refs.append(SR.render(feats, SR.decode_request(*args), seed=700 + i))
        seeds.append(700 + i)
    src = syn.make_source(3102, seconds=0.4)
    args = ("C4", "100", "t0g0", "20", "450", "60", "40", "100", "0", "!120", "AA")
    jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                 S.decode_request(*args)))
    refs.append(SR.render((src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                           src["sr"], src["y_len"]), SR.decode_request(*args), seed=702))
    seeds.append(702)
    outs = renderer.render(jobs, phi_seeds=seeds)
    for o, ref in zip(outs, refs):
        assert o.shape == ref.shape
        assert rms_err(o, ref) / max(1.0, float(np.max(np.abs(ref)))) < 2e-5


_random_flags = syn.random_flags


_FUZZ_DEFAULT = "2000"  # the driver-run suite: 2000 random flag strings, 166 mixed batches, 125 / 250 at other geometries / extreme requests
_FUZZ_FIRST = int(os.environ.get("GOOFER_FUZZ_FIRST", "0"))          # a soak run: GOOFER_FUZZ_FIRST=3000 GOOFER_FUZZ_CASES=9000


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("GOOFER_FUZZ_CASES", _FUZZ_DEFAULT))))
def test_random_flag_combinations_vs_oracle(renderer, case):
    """Flag interactions: random subsets of the whole vocabulary (assembly edits, jitter / sub-harmonic layers, post chain
    together), one note at a time so the legacy-RNG draw order matches, against the oracle's full render."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    rng = np.random.default_rng(9000 + case)
    # every fourth case on a hard source (interior V/UV transitions, fractional mask, bad / crossing formant frames, 40 dB jumps)
    src = (syn.make_hard_source if case % 4 == 3 else syn.make_source)(4000 + case, seconds=float(rng.uniform(0.3, 0.6)))
    flags = _random_flags(rng)
    pitch = ["A3", "C4", "E4", "G#4", "D5"][int(rng.integers(0, 5))]
    args = (pitch, str(int(rng.choice([60, 100, 140]))), flags, str(int(rng.integers(0, 60))), str(int(rng.integers(200, 700))),
            str(int(rng.integers(0, 120))), str(int(rng.choice([-200, 30, 80]))), str(int(rng.integers(50, 121))), "0",
            "!" + str(int(rng.choice([90, 120, 150]))), ["AA", "AA#5#AF#3#/+", "B7CPCV#2#Cb"][int(rng.integers(0, 3))])
    # (drawn behind everything above, so a case number keeps the request it had) the two per-formant strengths random_flags does
    # not hold — 'fstb', 'fstd': with 'fsta' / 'fstc' every column of the strength tracks is switched on and off on its own
    if rng.random() < 0.25:
        for name in ("fstb", "fstd"):
            if rng.random() < 0.6 and name not in flags:
                flags += "%s%d" % (name, int(rng.integers(-40, 41)))
        args = args[:2] + (flags,) + args[3:]
    feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
             src["sr"], src["y_len"])
    seed = 800 + case
    np.random.seed(77 + case)
    ref = SR.render(feats, SR.decode_request(*args), seed=seed)
    source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    np.random.seed(77 + case)
    (out,) = renderer.render([(source, S.decode_request(*args))], phi_seeds=[seed])
    assert out.shape == ref.shape, (flags, out.shape, ref.shape)
    assert np.all(np.isfinite(out)), flags
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, (flags, args, e)


COMBOS = [str(n) for n in golden("combo_index")["names"]]


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + max(4, int(os.environ.get("GOOFER_FUZZ_CASES", _FUZZ_DEFAULT)) // 12)))
def test_random_mixed_batch_equals_single_notes(renderer, case):
    """Six notes with unrelated random flag strings (assembly edits, jitter / sub-harmonic layers, post chain) rendered as ONE
    ragged batch against the same notes rendered one at a time: bit for bit.  The legacy-RNG draws are made note by note in
    both modes, so one seed in front of either run gives every note the same noise."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    rng = np.random.default_rng(50000 + case)
    jobs, seeds = [], []
    for k in range(6):
        src = syn.make_source(60000 + 10 * case + k, seconds=float(rng.uniform(0.25, 0.5)))
        flags = _random_flags(rng)
        pitch = ["A3", "C4", "E4", "G#4", "D5"][int(rng.integers(0, 5))]
        args = (pitch, str(int(rng.choice([60, 100, 140]))), flags, str(int(rng.integers(0, 60))), str(int(rng.integers(150, 500))),
                str(int(rng.integers(0, 120))), str(int(rng.choice([-200, 30, 80]))), str(int(rng.integers(50, 121))), "0",
                "!" + str(int(rng.choice([90, 120, 150]))), ["AA", "AA#5#AF#3#/+", "B7CPCV#2#Cb"][int(rng.integers(0, 3))])
        jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*args)))
        seeds.append(7000 + 6 * case + k)
    np.random.seed(case)
    try:
        batch = renderer.render(jobs, phi_seeds=seeds)
    except (ValueError, ZeroDivisionError):
        pytest.skip("a request the reference rejects as well (empty region): the planner raises its error")
    np.random.seed(case)
    for k, job in enumerate(jobs):
        (one,) = renderer.render([job], phi_seeds=[seeds[k]])
        assert one.shape == batch[k].shape
        assert np.array_equal(one, batch[k]), (case, k, float(np.max(np.abs(one - batch[k]))))


GEOMS = [(96000, 2048, 96), (48000, 2048, 512), (22050, 512, 128)]


@pytest.fixture(scope="module", params=GEOMS, ids=lambda g: "sr%d_nfft%d_hop%d" % g)
def geo_renderer(request):
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer
    sr, n_fft, hop = request.param
    c = Context(0)
    yield request.param, Renderer(c, hop=hop)
    c.close()


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + max(3, int(os.environ.get("GOOFER_FUZZ_CASES", _FUZZ_DEFAULT)) // 16)))
def test_random_flags_other_geometries_vs_oracle(geo_renderer, case):
    """The random flag vocabulary at other sample rates / transform sizes / hops (the 2048-point and 512-point kernels, a hop
    that is not a multiple of 64, eight output slots per hop) against the oracle's full render."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    (sr, n_fft, hop), renderer = geo_renderer
    rng = np.random.default_rng(70000 + case)
    src = syn.make_source(80000 + case, sr, n_fft, hop, seconds=float(rng.uniform(0.25, 0.4)))
    flags = _random_flags(rng)
    pitch = ["A3", "C4", "E4", "G#4", "D5"][int(rng.integers(0, 5))]
    args = (pitch, str(int(rng.choice([60, 100, 140]))), flags, str(int(rng.integers(0, 40))), str(int(rng.integers(150, 400))),
            str(int(rng.integers(0, 100))), str(int(rng.choice([-150, 30, 60]))), str(int(rng.integers(50, 121))), "0",
            "!" + str(int(rng.choice([90, 120, 150]))), ["AA", "AA#5#AF#3#/+", "B7CPCV#2#Cb"][int(rng.integers(0, 3))])
    feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
             src["sr"], src["y_len"])
    seed = 900 + case
    np.random.seed(177 + case)
    try:
        ref = SR.render(feats, SR.decode_request(*args), seed=seed, n_fft=n_fft, hop=hop)
    except (ValueError, ZeroDivisionError):
        pytest.skip("a request the reference rejects (empty region)")
    source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    np.random.seed(177 + case)
    (out,) = renderer.render([(source, S.decode_request(*args))], phi_seeds=[seed])
    assert out.shape == ref.shape, (flags, out.shape, ref.shape)
    assert np.all(np.isfinite(out)), flags
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, (sr, n_fft, hop, flags, args, e)


@pytest.mark.parametrize("case", range(_FUZZ_FIRST, _FUZZ_FIRST + max(6, int(os.environ.get("GOOFER_FUZZ_CASES", _FUZZ_DEFAULT)) // 8)))
def test_random_extreme_requests_vs_oracle(renderer, case):
    """The request arguments at their edges: notes of 5-120 ms (shorter than a window, a hop, a pitch-bend tick), no
    consonant, negative and large cutoffs, velocities 0 and 200, extreme tempi, long pitch-bend strings with runs, notes far
    above and below the source pitch — with random flags on top, against the oracle's full render."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    rng = np.random.default_rng(90000 + case)
    # (every third case on a hard source since the end of round 6: interior V/UV transitions, bad formant frames, 40 dB jumps)
    src = (syn.make_hard_source if case % 3 == 2 else syn.make_source)(95000 + case, seconds=float(rng.uniform(0.15, 0.7)))
    flags = _random_flags(rng) if rng.random() < 0.7 else ""
    pitch = ["C2", "A2", "C4", "B5", "C7"][int(rng.integers(0, 5))]
    bend = ["AA", "AA#50#", "/+/+/+#9#AAAA#3#gA", "B7CPCV#2#Cb" * 6, "AAABACADAEAFAGAH" * 4][int(rng.integers(0, 5))]
    args = (pitch, str(int(rng.choice([0, 1, 100, 199, 200]))), flags, str(int(rng.choice([0, 1, 5, 30, 200]))),
            str(int(rng.choice([5, 12, 40, 120, 2500]))), str(int(rng.choice([0, 1, 40, 300]))),
            str(int(rng.choice([-400, -50, 0, 1, 50, 350]))), str(int(rng.choice([0, 1, 100, 200]))), "0",
            "!" + str(int(rng.choice([20, 60, 120, 480]))), bend)
    feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
             src["sr"], src["y_len"])
    seed = 1900 + case
    np.random.seed(277 + case)
    try:
        ref = SR.render(feats, SR.decode_request(*args), seed=seed)
    except (ValueError, ZeroDivisionError, IndexError) as e:
        source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
        with pytest.raises((ValueError, ZeroDivisionError, IndexError)):       # the same request fails here as well
            renderer.render([(source, S.decode_request(*args))], phi_seeds=[seed])
        return
    source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    np.random.seed(277 + case)
    (out,) = renderer.render([(source, S.decode_request(*args))], phi_seeds=[seed])
    assert out.shape == ref.shape, (flags, args, out.shape, ref.shape)
    assert np.all(np.isfinite(out)) == np.all(np.isfinite(ref)), (flags, args)
    if np.all(np.isfinite(ref)):
        e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
        assert e < TOL, (flags, args, e)


def test_subharmonic_layer_above_nyquist_fires_on_every_sample(renderer):
    """Soak case 259412 (round 6): a forced-voiced C7 note with 'sh' and 'sg'.  The resampler runs the sub-harmonic layer at
    2 x f0 with a vibrato of depth 3 (SillySampler.py:1013-1020): up to 8 x f0, here above the sample rate.  The reference's tracker
    (`if phase >= 1`, GOOFER.py:693-696) then fires on every sample — more events than the n / 2 + 16 slots the layer shared with the
    pulse train, and the library refused the note.  The layer has a slot per sample now."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from oracle import sampler_ref as SR
    case = 259412
    src = syn.make_source(95000 + case, seconds=0.6625507038454174)
    bend = "AAABACADAEAFAGAH" * 4
    for args in (("C7", "100", "g61es74B45FV1sh72sr73sg67vl23", "1", "5", "0", "-400", "100", "0", "!60", bend),
                 ("C7", "100", "FV1sh72sg67", "60", "300", "40", "30", "100", "0", "!120", bend)):     # ... and where it is audible
        feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                 src["sr"], src["y_len"])
        np.random.seed(277 + case)
        ref = SR.render(feats, SR.decode_request(*args), seed=1900 + case)
        source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
        np.random.seed(277 + case)
        (out,) = renderer.render([(source, S.decode_request(*args))], phi_seeds=[1900 + case])
        assert out.shape == ref.shape
        assert rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref)))) < TOL, args
    assert float(np.max(np.abs(ref))) > 0.05


@pytest.mark.parametrize("name", COMBOS)
def test_flag_combinations_match_reference(renderer, name):
    """The same 16 random flag subsets the reference rendered (tests/golden/combo_*.npz), on the device."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    g = golden(name)
    i = COMBOS.index(name)
    rng = np.random.default_rng(7700 + i)
    flags = syn.random_flags(rng)
    src = syn.make_source(5000 + i, seconds=float(rng.uniform(0.3, 0.55)))
    args = [str(a) for a in g["args"]]
    assert args[2] == flags
    seed, legacy, _ = (int(v) for v in g["seed"])
    source = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    np.random.seed(legacy)
    (out,) = renderer.render([(source, S.decode_request(*args))], phi_seeds=[seed])
    ref = g["out"]
    assert out.shape == ref.shape
    e = rms_err(out, ref) / max(1.0, float(np.max(np.abs(ref))))
    assert e < TOL, (flags, e)


def test_index_plans_on_device(renderer):
    """The integer path end to end on the device: the reference's 53 index-plan cases (loop modes L0/L1/L2, every slicing
    variant, reverse, velocity) with the probe features env[b, t] = t, mask[n] = n.  The assembled envelope spells out which
    source frames every output frame was built from and the assembled mask which source sample every output sample is —
    they must be the reference's (SillySampler.py:453-500, 625-788): the mask bit for bit, frames that are copies exactly,
    cross-faded / resampled frames to fp32 rounding of the frame number."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    from test_product_sampler import probe_source
    g = golden("index_plans")
    env, f0, mask, forms, sr, n, T = probe_source()
    jobs, tags = [], []
    for tag in g["names"]:
        if f"{tag}_error" in g.files:
            continue
        req = S.decode_request(*[str(a) for a in g[f"{tag}_args"]])
        jobs.append((Source.from_pack(env, f0, mask, forms, sr, n), req))
        tags.append(str(tag))
    assert len(jobs) >= 45
    # assembly only: the probe mask counts samples, so the "f0" it gates is absurd and a synthesis of it overflows the pulse
    # onset slots (which Context.check() now reports, as it should)
    prep = renderer.prepare(jobs, phi_seeds=list(range(len(jobs))), trim_rows=False)
    # The tap blend runs in fp32: copied frames must come out exactly, blended ones to fp32 rounding of the frame
    # number (round 5 also ran round 4's fp64 gather kernel here, which pinned whole frame numbers exactly; that kernel is gone).
    renderer.assemble(prep)
    torch.cuda.synchronize()
    env_dev = prep["env"].cpu().numpy()
    mask_dev = prep["mask"].cpu().numpy()
    e_off, s_off = prep["env_off"], prep["sample_off"]
    for j, tag in enumerate(tags):
        want_row = g[f"{tag}_env_row"]
        got_row = env_dev[e_off[j]:e_off[j + 1], 0].astype(np.float64)
        assert got_row.shape == want_row.shape, tag
        assert np.max(np.abs(got_row - want_row), initial=0.0) <= 2e-7 * max(1.0, float(want_row.max(initial=0.0))), tag
        # every bin of a frame is the same combination of source frames
        assert np.array_equal(env_dev[e_off[j]:e_off[j + 1], 0], env_dev[e_off[j]:e_off[j + 1], 512]), tag
        want_mask = np.asarray(g[f"{tag}_mask"], dtype=np.float32)
        assert np.array_equal(mask_dev[s_off[j]:s_off[j + 1]], want_mask), tag


def test_trimmed_envelope_rows_change_nothing(renderer):
    """Renderer.prepare(trim_rows=True) leaves out the envelope rows gf.synthesize never reaches (the L0 loop hands over more
    frames than the note has STFT frames, GOOFER.py:1115-1119 cuts them): the audio is the same bits, with fewer rows assembled."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    jobs, seeds = [], []
    for k, (flags, length, vel) in enumerate([("L0fst30", 900, 100), ("L1g-10", 700, 100), ("L2br20", 800, 60), ("L0R1es-40", 1000, 140),
                                              ("fa20fw30vf30", 650, 100), ("L0sa40", 500, 100)]):
        src = syn.make_source(91000 + k, seconds=0.45)
        args = ("C4", str(vel), flags, "30", str(length), "80", "40", "100", "0", "!120", "AA#5#AF#3#/+")
        jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*args)))
        seeds.append(8100 + k)
    full = renderer.prepare(jobs, phi_seeds=seeds, trim_rows=False)
    trim = renderer.prepare(jobs, phi_seeds=seeds, trim_rows=True)
    assert sum(trim["env_lens"]) < sum(full["env_lens"]) and trim["lens"] == full["lens"]
    a = renderer.run(full, seed=3)["mix"].cpu().numpy()
    b = renderer.run(trim, seed=3)["mix"].cpu().numpy()
    renderer.ctx.check()
    assert a.shape == b.shape and np.array_equal(a, b)
    assert np.isfinite(a).all() and np.abs(a).max() > 0


def test_notes_sharing_a_source_render_like_single_notes(renderer):
    """Several notes cut from ONE voicebank sample (the usual case in a song): the batch uploads the sample once and every note
    renders to the bits it has alone."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    srcs = [syn.make_source(93000 + k, seconds=0.5) for k in range(2)]
    sources = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s in srcs]
    reqs = [("C4", "100", "g10", "30", "400", "80", "40"), ("E4", "120", "L1fa20", "10", "700", "60", "-200"), ("G3", "80", "L2br30", "50", "300", "90", "20"),
            ("A3", "100", "R1", "20", "500", "70", "30"), ("D4", "100", "fst30es-30", "0", "350", "50", "60")]
    jobs, seeds = [], []
    for k, r in enumerate(reqs):
        args = r + ("100", "0", "!120", "AA#5#AF#3#/+")
        jobs.append((sources[k % 2], S.decode_request(*args)))
        seeds.append(9400 + k)
    batch = renderer.render(jobs, phi_seeds=seeds)
    for k, job in enumerate(jobs):
        (one,) = renderer.render([job], phi_seeds=[seeds[k]])
        assert np.array_equal(one, batch[k]), k


def test_source_arena_keeps_samples_resident(renderer):
    """Renderer.sources (render.SourceArena): a Source is uploaded the first time a batch names it and found resident afterwards;
    the audio does not depend on where in the arena it sits; a Source nobody holds any more leaves the arena's table (its id may
    be recycled); past the budget the arena starts over in fresh arrays and batches prepared before keep theirs."""
    import gc
    from goofer_amd.render import Renderer, Source, SourceArena
    from goofer_amd import sampler as S
    r = Renderer(renderer.ctx)
    def make(k):
        src = syn.make_source(93000 + k, seconds=0.4)
        return Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
    srcs = [make(k) for k in range(4)]
    req = S.decode_request("C4", "100", "L1g-5", "20", "600", "60", "30", "100", "0", "!120", "AA#8#AK")
    jobs = [(s, req) for s in srcs]
    a = r.render(jobs, seed=11)
    used = (r.sources.k_used, r.sources.m_used)
    assert used[0] == sum(s.knots.size for s in srcs) and used[1] == sum(s.ylen for s in srcs) and len(r.sources.where) == 4
    b = r.render(list(reversed(jobs)), seed=11)              # resident: nothing uploaded, other batch order
    assert (r.sources.k_used, r.sources.m_used) == used
    # (the Philox stream of a note is keyed by its position here, so compare position by position through a second render)
    c = r.render(jobs, seed=11)
    assert all(np.array_equal(x, y) for x, y in zip(a, c)) and len(b) == 4
    # a fresh arena places the same sources elsewhere (two fillers in front): same audio
    r2 = Renderer(renderer.ctx)
    fill = [make(10), make(11)]
    r2.render([(f, req) for f in fill], seed=1)
    d = r2.render(jobs, seed=11)
    assert all(np.array_equal(x, y) for x, y in zip(a, d))
    del fill
    gc.collect()
    assert len(r2.sources.where) == 4                        # the fillers' entries went with them
    # budget: the third source does not fit beside the first two -> fresh arrays, earlier batches keep the old ones
    small = Renderer(renderer.ctx)
    per = 2 * srcs[0].knots.size + 4 * srcs[0].ylen
    small.sources = SourceArena(renderer.ctx, budget_bytes=int(2.5 * per))
    prep01 = small.prepare(jobs[:2], note_ids=[0, 1])
    old_knots = small.sources.knots
    e = small.render(jobs[2:], seed=11)                      # resets the arena
    assert small.sources.knots is not old_knots and len(small.sources.where) == 2
    out01 = small.run(prep01, seed=11)["mix"].cpu().numpy()  # prepared against the old arrays: still valid
    small.ctx.check()
    so = prep01["sample_off"]
    assert np.array_equal(out01[so[0]:so[1]], a[0]) and np.array_equal(out01[so[1]:so[2]], a[1])
    f = r.render(jobs[2:], seed=11)
    assert all(np.array_equal(x, y) for x, y in zip(e, f))


def test_sample_assemble_fast_path_is_bit_identical(renderer):
    """k_sample_assemble's branch-free path (all of a thread's loads in flight together; round 5) against the per-sample
    function it restates: f0 and voicing mask of every reference flag set that takes it (and of those that do not) bit for bit,
    plus 96 notes of BASELINE configs 3 and 4 (tiled tails, reverse, every loop mode)."""
    from goofer_amd.render import Source
    from goofer_amd import sampler as S
    jobs = [_job(name)[1:] for name in SUPPORTED]
    for config, ids in ((3, range(0, 1024, 16)), (4, range(32))):
        for i in ids:
            src, req, _ = syn.config_note(config, int(i))
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
    prep = renderer.prepare(jobs, phi_seeds=list(range(len(jobs))))
    got = {}
    for v in (0, 1):
        renderer.ctx.set_option("sa_fast", v)
        try:
            prep["f0"].zero_()
            prep["mask"].zero_()
            renderer.assemble(prep)
            torch.cuda.synchronize()
            got[v] = (prep["f0"].cpu().numpy().copy(), prep["mask"].cpu().numpy().copy())
        finally:
            renderer.ctx.set_option("sa_fast", 1)
    assert np.array_equal(got[0][0], got[1][0])
    assert np.array_equal(got[0][1], got[1][1])
    assert float(np.abs(got[1][0]).max()) > 50.0
