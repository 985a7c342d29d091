"""GPU: the batch shapes `bench.py --job-notes` times for BASELINE configs 4 and 5, at their full size.

Config 4 (10 000 notes of 0.1-3 s, L0/L1/L2): one rank's job is rendered in sub-batches of 4096 notes ordered by length, so
a sub-batch holds four rounds of pulse-walk waves and notes of very different lengths — here the FIRST such sub-batch (the
4096 longest notes of the job, incl. every 2.5-3 s note).  Config 5 (96 kHz, n_fft 2048, hop 96, br / es): one 1024-note
batch through the one-stem-per-wave overlap-add (hop != n_fft / 4).  The oracle cannot render this much in seconds, so the
full-size checks are structural — run-to-run determinism, a 32-note sample bit-equal to the same notes rendered singly,
finite and non-silent audio — and three notes of the long tail go through the oracle (same injected phases)."""
import numpy as np
import pytest

from conftest import rms_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _job_ids(config, job_notes, sub_batch):
    """The first sub-batch of rank 0 of a one-GPU job, as bench.py builds it."""
    from goofer_amd import synthetic as syn
    from goofer_amd.shard import assign_lpt
    est = [syn.config_note_frames(config, i) for i in range(job_notes)]
    ids = sorted(assign_lpt(est, 1)[0], key=lambda i: (-est[i], i))
    return ids[:sub_batch], est


def _structural_checks(ctx, wl, ids, config, sample):
    from goofer_amd.workload import SamplerWorkload
    a = wl.step()
    torch.cuda.synchronize()
    mix = a["mix"].clone()
    b = wl.step()
    torch.cuda.synchronize()
    assert torch.equal(mix, b["mix"]), "two runs of the same batch differ"
    assert bool(torch.isfinite(mix).all()) and float(mix.abs().max()) <= 1.5
    so = wl.prep["sample_off"]
    for k in sample:                                            # no note is silent
        assert float(mix[int(so[k]):int(so[k + 1])].abs().max()) > 1e-3, ids[k]
    for k in sample:                                            # a note's bits do not depend on the batch it sits in
        one = SamplerWorkload(ctx, config, [ids[k]])
        got = one.step()["mix"]
        torch.cuda.synchronize()
        assert torch.equal(got, mix[int(so[k]):int(so[k + 1])]), ids[k]
    ctx.plan(wl.geo["sr"], wl.geo["n_fft"], wl.geo["hop"])
    return mix


def _oracle_notes(ctx, config, note_ids, tol):
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.render import Renderer, Source
    from oracle import sampler_ref as SR
    geo = syn.config_geometry(config)
    r = Renderer(ctx, hop=geo["hop"])
    jobs, refs, seeds = [], [], []
    for i in note_ids:
        src, req, phi_seed = syn.config_note(config, i)
        jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                     S.decode_request(*syn.request_args(req))))
        feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                 src["sr"], src["y_len"])
        refs.append(SR.render(feats, SR.decode_request(*syn.request_args(req)), seed=phi_seed, n_fft=geo["n_fft"], hop=geo["hop"]))
        seeds.append(phi_seed)
    outs = r.render(jobs, phi_seeds=seeds)
    for i, o, ref in zip(note_ids, outs, refs):
        assert o.shape == ref.shape
        e = rms_err(o, ref) / max(1.0, float(np.max(np.abs(ref))))
        assert e < tol, (config, i, e)


def test_config4_first_sub_batch_full_size():
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    ids, est = _job_ids(4, 10000, 4096)
    hop, sr = syn.config_geometry(4)["hop"], syn.config_geometry(4)["sr"]
    assert len(ids) == 4096 and est[ids[0]] * hop / sr > 2.9 and est[ids[-1]] < est[ids[0]]
    flags = {syn.config_flags(4, i)[-2:] for i in ids[:64]}
    assert {"L0", "L1", "L2"} <= flags                        # the three loop modes ride in one batch
    ctx = Context(0)
    try:
        wl = SamplerWorkload(ctx, 4, ids)
        assert wl.frames == sum(est[i] for i in ids)           # the planner produced what the assignment counted on
        sample = [0, 1, 2, 3, 17, 100, 511, 1024, 2047, 2048, 3000, 4095] + list(range(200, 4000, 190))
        _structural_checks(ctx, wl, ids, 4, sample[:32])
        del wl
        # three notes of the long tail (>= 2.5 s) against the oracle, as one ragged batch with two short neighbours
        long_ids = [i for i in ids if est[i] * hop / sr >= 2.5][:3]
        assert len(long_ids) == 3
        _oracle_notes(ctx, 4, long_ids + [ids[-1], ids[-2]], 2e-5)
    finally:
        ctx.close()


def test_config5_batch_full_size():
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    ids, est = _job_ids(5, 1024, 4096)
    assert len(ids) == 1024
    ctx = Context(0)
    try:
        wl = SamplerWorkload(ctx, 5, ids)
        assert wl.geo["n_fft"] == 2048 and wl.geo["hop"] == 96 and wl.frames == sum(est[i] for i in ids) > 1_000_000
        sample = list(range(0, 1024, 32))
        _structural_checks(ctx, wl, ids, 5, sample)
        del wl
        _oracle_notes(ctx, 5, [ids[0], ids[1], ids[-1]], 2e-5)
    finally:
        ctx.close()


def test_config5_frame_skipping_changes_no_bit():
    """hop != n_fft / 4 (BASELINE config 5: n_fft 2048, hop 96): the noise stems' transforms are skipped where the smoothed
    voicing mask makes the stem gain exactly zero over every hop a frame reaches (k_frame_skip; neither written by
    k_noise_spectra nor transformed by k_irfft_ola1).  With the skipping off every frame is transformed: the stems and the
    mix must be the same bits — on sources with unvoiced stretches, so that both stems have skipped and live frames."""
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    geo = syn.config_geometry(5)
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=geo["hop"])
        jobs = []
        for k, i in enumerate([3, 11, 40, 77, 130, 500]):
            src, req, _ = syn.config_note(5, i)
            if k % 2 == 0:
                src = syn.with_unvoiced_gaps(src, 0.35, 4000 + i)
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
        prep = r.prepare(jobs, note_ids=list(range(len(jobs))))
        outs = {}
        for skip in (1, 0):
            ctx.set_option("skip_zero", skip)
            o = r.run(prep, seed=5, keep_stems=True)
            ctx.check()
            outs[skip] = {k: o[k].cpu().numpy().copy() for k in ("harm", "uv", "bre", "mix")}
        ctx.set_option("skip_zero", 1)
        for k in ("harm", "uv", "bre", "mix"):
            assert np.array_equal(outs[1][k], outs[0][k]), k
        ctx.set_option("overlap", 0)                          # ... and everything on one stream (n_fft 2048: the spectra pipeline)
        try:
            o = r.run(prep, seed=5, keep_stems=True)
            ctx.check()
            for k in ("harm", "uv", "bre", "mix"):
                assert np.array_equal(outs[1][k], o[k].cpu().numpy()), k
        finally:
            ctx.set_option("overlap", 1)
        uv, bre = outs[1]["uv"], outs[1]["bre"]
        assert np.isfinite(outs[1]["mix"]).all() and np.abs(uv).max() > 0 and np.abs(bre).max() > 0
        assert (uv == 0).mean() > 0.2 and (bre == 0).mean() > 0.02        # both kinds of silence occur
    finally:
        ctx.close()
