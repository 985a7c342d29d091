"""GPU: hard sources (synthetic.make_hard_source — 3-6 interior V/UV transitions with fp16 ramps and fractional plateaus, formant
frames that are 0 / NaN / negative / above Nyquist, crossing and all-invalid tracks, 40 dB envelope jumps, bands of near-zero bins)
through the full sampler path: against the reference's own renders (tests/golden/sampler_hard_*.npz, written by
make_golden.gen_sampler_hard from the imported reference) and, on BASELINE config 3 / 4 requests, against the oracle.
Matches SillySampler.py:242-283, GOOFER.py:556-569, 849-873, 1131-1144."""
import numpy as np
import pytest

from conftest import golden, rms_err
from goofer_amd import synthetic as syn

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HARD = [str(n) for n in golden("sampler_hard_index")["names"]]
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
    g = golden(name)
    src, req = syn.hard_case(HARD.index(name))
    args = [str(a) for a in g["args"]]
    assert args == [str(a) for a in syn.request_args(req)]
    return g, Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]), S.decode_request(*args)


@pytest.mark.parametrize("name", HARD)
def test_hard_source_matches_reference(renderer, name):
    """One hard note against the reference's render of it: the assembled voicing mask bit for bit, f0 to fp32 rounding, the three
    stems and the note inside the 1e-4 bound (measured: printed)."""
    g, source, req = _job(name)
    (out,), parts = renderer.render([(source, req)], phi_seeds=[int(g["seed"][0])], return_parts=True)
    ref = g["out"]
    assert out.shape == ref.shape and np.isfinite(out).all()
    assert np.array_equal(parts["mask"].cpu().numpy(), np.asarray(g["mask_new"], dtype=np.float32))
    f0 = parts["f0"].cpu().numpy()
    assert np.max(np.abs(f0 - g["f0_new"]) / np.maximum(1.0, np.abs(g["f0_new"]))) < 3e-7
    scale = max(1.0, float(np.max(np.abs(ref))))
    errs = {"out": rms_err(out, ref) / scale}
    if int(g["n_calls"][0]) == 1:                             # (the stems of the main synthesize call are the note's stems)
        for k in ("harm", "uv", "bre"):
            errs[k] = rms_err(parts["stems"][k].cpu().numpy(), g[k])
    print(name, {k: "%.2e" % v for k, v in errs.items()})
    assert max(errs.values()) < TOL, (name, errs)


def test_hard_sources_as_one_batch(renderer):
    """All twelve as ONE ragged batch, against the same renders."""
    jobs, seeds, refs = [], [], []
    for name in HARD:
        g, source, req = _job(name)
        jobs.append((source, req))
        seeds.append(int(g["seed"][0]))
        refs.append(g["out"])
    outs = renderer.render(jobs, phi_seeds=seeds)
    worst = 0.0
    for name, o, r in zip(HARD, outs, refs):
        assert o.shape == r.shape, name
        e = rms_err(o, r) / max(1.0, float(np.max(np.abs(r))))
        worst = max(worst, e)
        assert e < TOL, (name, e)
    print("hard fixtures as one batch: worst %.3g" % worst)


@pytest.mark.parametrize("config,ids", [(3, list(range(0, 1024, 16))), (4, list(range(5, 10000, 157))), (5, list(range(3, 1024, 64)))])
def test_baseline_requests_on_hard_sources_vs_oracle(config, ids):
    """64 requests each of BASELINE configs 3 (full formant set, V/B/U mix) and 4 (L0 / L1 / L2 over log-uniform lengths), 16 of
    config 5 (96 kHz, n_fft 2048, hop 96, br + es: the spectra pipeline with per-frame skipping) on the hard version of their
    samples, one device batch per config, against the CPU oracle's render of every note."""
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
            src, req, phi_seed = syn.config_note(config, i, hard=True)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
            feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                     src["sr"], src["y_len"])
            refs.append(SR.render(feats, SR.decode_request(*syn.request_args(req)), seed=phi_seed, n_fft=geo["n_fft"], hop=geo["hop"]))
            seeds.append(phi_seed)
        outs = r.render(jobs, phi_seeds=seeds)
        errs = []
        for i, (o, ref) in enumerate(zip(outs, refs)):
            assert o.shape == ref.shape and np.isfinite(o).all(), i
            errs.append(rms_err(o, ref) / max(1.0, float(np.max(np.abs(ref)))))
        errs = np.asarray(errs)
        print("config %d on hard sources, %d notes vs oracle: worst %.3g (position %d), mean %.3g" % (config, errs.size, errs.max(), int(errs.argmax()), errs.mean()))
        assert errs.max() < TOL, (config, int(errs.argmax()), float(errs.max()))
    finally:
        ctx.close()
