"""GPU: the BASELINE.json headline configuration (config 3, 1024 notes, 194 560 frames, 49.7 M samples) through
size-independent properties — the oracle cannot render this much in seconds, so parity here is structural:
perfect reconstruction of the framewise FFT pair, independence of the notes of a batch, run-to-run determinism,
and bit-invariance of the normalised render under a power-of-two scaling of every envelope."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

NOTES = 1024


@pytest.fixture(scope="module")
def workload():
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    c = Context(0)
    wl = SamplerWorkload(c, 3, list(range(NOTES)))
    yield c, wl
    c.close()


def test_stft_istft_round_trip_full_size(workload):
    """istft(stft(x)) == x on [0, hop*(T-1)) for every note (sqrt-Hann, 75 % overlap), zero-filled tail after it."""
    ctx, wl = workload
    o = wl.prep["offsets"]
    s_off, f_off = o["s_off"], o["f_off"]
    assert int(f_off[-1]) == wl.frames == 194560 and int(s_off[-1]) == wl.samples
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(wl.samples, device="cuda", generator=g)
    S = ctx.rfft_frames(x, o["d_s"], o["d_f"], wl.frames)
    y = ctx.irfft_ola(S, o["d_s"], o["d_f"], wl.samples)
    torch.cuda.synchronize()
    hop = wl.geo["hop"]
    keep = torch.zeros(wl.samples, dtype=torch.bool, device="cuda")
    for i in range(NOTES):
        n = int(s_off[i + 1] - s_off[i])
        keep[int(s_off[i]):int(s_off[i]) + hop * (n // hop)] = True
    err = (y - x)[keep].abs().max().item()
    assert err < 2e-5, err
    assert y[~keep].abs().max().item() == 0.0


def test_notes_are_independent_and_deterministic_full_size(workload):
    """A note renders to the same bits inside the 1024-note batch, in a 7-note batch, and on a second run."""
    from goofer_amd.workload import SamplerWorkload
    ctx, wl = workload
    a = wl.step()
    torch.cuda.synchronize()
    mix_a = a["mix"].clone()
    b = wl.step()
    torch.cuda.synchronize()
    assert torch.equal(mix_a, b["mix"])
    assert torch.isfinite(mix_a).all() and mix_a.abs().max().item() <= 1.5
    ids = [0, 1, 257, 511, 640, 1000, 1023]
    small = SamplerWorkload(ctx, 3, ids)
    c = small.step()
    torch.cuda.synchronize()
    so_big, so_small = wl.prep["sample_off"], small.prep["sample_off"]
    for k, i in enumerate(ids):
        got = c["mix"][int(so_small[k]):int(so_small[k + 1])]
        ref = mix_a[int(so_big[i]):int(so_big[i + 1])]
        assert torch.equal(got, ref), i
    ctx.plan(wl.geo["sr"], wl.geo["n_fft"], wl.geo["hop"])


def test_envelope_scale_invariance_full_size(workload):
    """Every stem is linear in the envelope and the render is peak-normalised (normalize = 1): doubling all
    assembled envelope rows — exact in fp32, through every product of the chain — must give the same bits."""
    ctx, wl = workload
    r, prep = wl.renderer, wl.prep

    def synth():
        out = ctx.synth_batch(prep["env"], prep["env_lens"], prep["f0"], prep["mask"], prep["lens"], prep["params"],
                              formants=prep["formants"], seed=0, want_rec=False, want_mix=True, offsets=prep["offsets"])
        torch.cuda.synchronize()
        return {k: out[k].clone() for k in ("harm", "uv", "bre", "mix")}

    assert float(np.min(prep["params"]["normalize"])) == 1.0
    r.assemble(prep)
    a = synth()
    prep["env"].mul_(2.0)
    b = synth()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    r.assemble(prep)                                           # restore the envelopes for other tests
    c = synth()
    for k in a:
        assert torch.equal(a[k], c[k]), k
