#!/usr/bin/env python3
"""bench.py — resynth frames/s of the GOOFER hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--notes 1024] [--config 3]

One process per GPU (the driver launches N>1 through torch.distributed.run; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the environment).  A *step* is one pass of the hot path
(goofer_render_batch: note assembly, pulse train, then the stem walkers — noise stems and the harmonic stem each from
their inputs to finished samples — and the per-note finish: 1 / max|S|, peak normalise, V/B/U mix) over one ragged batch
of synthetic notes already resident in HBM.  Notes are independent, so ranks shard by note id with no data-path collective (weak scaling:
`--notes` per GPU); the only collectives are the barriers and the MAX of the elapsed time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      HIP-event duration of the longest kernel of the timed steps against its SURVEY 8(d) algorithmic bytes: `frac` is
                that kernel's fraction of the 8 TB/s HBM peak; `traffic` (HBM bytes from the committed counter passes) only when
                those passes were measured on this very source tree, else "stale": true; `valu`: its vector instructions per
                second against 1024 SIMDs x one wave64 instruction per 2 cycles (SQ_INSTS_VALU of the same tree)
  config4 / config5   BASELINE configs 4 (10 000 notes, mixed loop modes) and 5 (96 kHz, n_fft 2048, hop 96) as fixed jobs on
                this GPU, after the timed region: value, ms per pass, sub-batches, their own roofline + roofline_step
  value_skip_zero_off / value_unvoiced_30pct   the same step with no transform skipped / on 30 %-unvoiced sources
  roofline_fft  the same for the framewise rFFT kernel (the kernel BASELINE's 40 % target names)
  roofline_step the whole step against its end-to-end algorithmic bytes and the HBM traffic the counter passes measured
  cpu_baseline  the numpy/C oracle (a port of the reference's CPU path) timed on this box's host cores: one thread, and a
                pool of worker processes over notes
  host_inclusive what a host that starts from argument strings pays around the step (decode, plan, upload, download)

`--job-notes N --config 4` (or 5) renders ONE fixed job of N notes instead: every rank derives the same
longest-processing-time assignment from the notes' frame counts, renders its share in sub-batches, and the line reports
`"scaling": "strong"` with the per-rank frames and the imbalance.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured ceiling
VALU_PEAK = 1024 * 2.4e9 / 2.0   # wave64 vector instructions/s: 1024 SIMDs, one per 2 cycles on the SIMD-32 (same guide)


def stage_alg_bytes(stage: str, F: int, N: int, B: int, hop: int, n_fft: int, executed: float = 1.0, E: float = 0.0, K: int = 0) -> float:
    """Algorithmic HBM bytes of one launch of each stage: unique bytes that must
    cross HBM if nothing were re-read.  F frames, N samples, B bins.  ``executed``: share of the 3 F stem transforms the
    overlap-add kernel of the spectra-in-HBM pipeline really ran (per-frame exact-zero skipping): a skipped frame's spectrum is
    neither written nor read, and is not credited."""
    fft = (4 * hop + 8 * B) * F                      # SURVEY.md §8(d) ALG_BYTES_FFT per frame-transform
    table = {
        "rfft_frames": fft, "rfft_frames_standalone": fft, "irfft_harm": fft, "irfft_breath": fft, "irfft_unvoiced": fft,
        "ola_harm": 4 * n_fft * F + 4 * N, "ola_breath": 4 * n_fft * F + 4 * N, "ola_unvoiced": 4 * n_fft * F + 4 * N,
        "ola3_gains": 3 * 4 * n_fft * F + 12 * N,     # three windowed-frame buffers in, three stems out
        "irfft_ola3": 3 * 8 * B * F * executed + 12 * N,   # three spectra in (the executed ones), three stems out (frames stay in LDS)
        "harm_shape": (16 * B + 4 * B) * F,          # S in+out, env in
        "noise_spectra": (16 * B + 4 * B) * F,       # two spectra out, env in (+4B when phi is injected)
        "gauss_env": 8 * B * F, "warp_env": 8 * B * F, "assemble": 12 * B * F + 12 * N,
        "phase_inc": 12 * N, "pulse_onsets": 4 * N, "pulse_place": 4 * N,     # the walk reads f0 and divides by sr itself
        "mask_short": 4 * N / 4 + 8 * N / 4, "stem_gains": 24 * N,
        "apply_gain": 16 * N,                         # three stems in, the mix out (mix_only)
        "setup_maps": 8 * N + 12 * F,
        # stem-split walkers (stems.hip): envelope row (+ unique pulse samples) in, finished samples out
        "noise_stems": 4 * B * F + 8 * N, "harm_stem": (4 * hop + 4 * B) * F + 4 * N,
        "note_finish": 16 * N,                        # three stems in, the mix out (mix_only)
        # the assembly's three large kernels (round 5: each its own profile stage).  E edited source rows of K fp16 knots
        "env_edit": (2 * K + 4 * B) * E,              # knots in, edited rows out
        "env_rows": 4 * B * (E + F),                  # edited rows in, assembled rows out (the warped copy is an intermediate: not credited)
        "sample_assemble": 8 * N,                     # f0 and voicing mask out (the source masks / pitch curves are re-read from L2)
    }
    return float(table[stage])


# stages that bracket several launches (the whole assembly call, the map kernels) or the phase scan: not roofline candidates.
# The assembly's three large kernels are single launches with their own stages (env_edit, env_rows, sample_assemble) and ARE.
MULTI_STAGES = {"assemble", "setup_maps", "phase_inc", "pulse_onsets"}
STAGE_KERNEL = {"rfft_frames": "void k_rfft_frames<512>", "rfft_frames_standalone": "void k_rfft_frames<512>",
                "irfft_harm": "void k_irfft_frames<512>", "harm_shape": "void k_harm_shape<9>",
                "noise_spectra": "void k_noise_spectra<9>", "ola3_gains": "k_ola3_gains", "irfft_ola3": "void k_irfft_ola3<512>",
                "apply_gain": "k_apply_gain", "pulse_onsets": "k_pulse_onsets_par", "pulse_place": "k_pulse_place",
                "mask_short": "k_mask_short", "phase_inc": "k_phase_inc", "setup_maps": "k_scale_f0",
                "noise_stems": "void k_noise_stems<512, false>", "harm_stem": "void k_harm_stem<512>", "note_finish": "k_note_finish",
                "env_edit": "void k_env_edit<false, 9>", "env_rows": "void k_env_rows<true, 9>", "sample_assemble": "void k_sample_assemble<4>"}


# kernel names of the fixed-job legs where they differ from the default step's (instantiations per row width / transform size)
LEG_KERNEL = {4: {"env_rows": "void k_env_rows<false, 9>"},
              5: {"rfft_frames": "void k_rfft_frames<1024, true>", "harm_shape": "void k_harm_shape<17, true>",
                  "noise_spectra": "void k_noise_spectra<17, true, false, true>", "irfft_ola3": "void k_irfft_ola1<1024, 8>",
                  "apply_gain": "k_note_finish", "env_edit": "void k_env_edit<false, 17>", "env_rows": "void k_env_rows<false, 17>"}}
SHARED_WITH = {"noise_stems": "the tail of the pulse chain on the side stream (its time alone is a few percent lower)",
               "noise_spectra": "the pulse chain on the side stream", "mask_short": "the pulse chain on the side stream",
               "pulse_place": "the envelope gather / noise walker on the caller's stream",
               "env_edit": "the f0 / mask kernel on the side stream", "env_rows": "the phase scan / pulse placement on the side stream",
               "sample_assemble": "the envelope edit on the caller's stream"}


def make_roof(per, Fb, Nb, B, hop, n_fft, executed, Eb, Kmax, pmc_frames, leg=None, counters=True):
    """roof(stage) -> the `roofline` object of one kernel: SURVEY 8(d) algorithmic bytes of one launch / its HIP-event time
    against the 8 TB/s HBM peak (`frac` is always this fraction), the counter-measured HBM bytes per launch (`traffic`) and the
    vector-pipe figure (`valu`) from the committed passes of this very tree."""
    def roof(stage, frames=None, samples=None):
        frames = Fb if frames is None else frames
        samples = Nb if samples is None else samples
        ms = per[stage]
        alg = stage_alg_bytes(stage, frames, samples, B, hop, n_fft, executed, E=Eb, K=Kmax)
        a = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        kern = LEG_KERNEL.get(leg, {}).get(stage)
        tr = pmc_traffic(stage, pmc_frames, leg, kern) if counters else {"bytes": None, "source": None, "stale": None}
        valu = sq_valu_issue(stage, ms, pmc_frames, leg, kern) if counters else None
        return {"kernel": stage, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a / HBM_PEAK_GBS, "traffic": tr["bytes"], "traffic_source": tr["source"], "stale": tr["stale"],
                "ms_per_launch": ms, "alg_bytes_per_launch": alg, "transforms_executed": executed if stage == "irfft_ola3" else None,
                "shared_with": SHARED_WITH.get(stage), "valu": valu}
    return roof


def executed_share(ctx):
    """share of the stem transforms the overlap-add kernel executed (configs with per-frame skipping: the last sub-batch's bits)"""
    try:
        fsk = ctx.debug_fetch("frame_skip")
        if fsk.size:
            return 1.0 - float(((fsk & 1) != 0).sum() + ((fsk & 2) != 0).sum()) / (3.0 * fsk.size)
    except Exception:
        pass
    return 1.0


def job_leg(local, config, job_notes, sub_batch, steps, warmup):
    """One BASELINE fixed job (config 4: 10 000 notes with mixed loop modes; config 5: 96 kHz / n_fft 2048 / hop 96) on this GPU,
    on a handle of its own: notes ordered by length into sub-batches, `warmup` untimed passes, `steps` timed passes (HIP events
    around them + a device synchronise), then one bracketed pass for the stage times.  Returns the leg's object of the line."""
    import torch
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    ctx = Context(local)
    try:
        est = [syn.config_note_frames(config, i) for i in range(job_notes)]
        ids = sorted(range(job_notes), key=lambda i: (-est[i], i))
        subs = [SamplerWorkload(ctx, config, ids[k:k + sub_batch]) for k in range(0, len(ids), sub_batch)]
        geo = subs[0].geo
        B, hop, n_fft, sr = geo["n_fft"] // 2 + 1, geo["hop"], geo["n_fft"], geo["sr"]
        frames, samples = sum(w.frames for w in subs), sum(w.samples for w in subs)

        def one_pass():
            out = None
            for w in subs:
                out = w.step()
            return out

        for _ in range(warmup):
            one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = one_pass()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        mix = last["mix"]
        assert bool(torch.isfinite(mix).all()) and float(mix.abs().max()) > 0.0, "the leg produced no valid audio"
        ctx.profile_begin(len(subs))
        one_pass()
        prof = ctx.profile_end()
        n = max(1, prof["steps"])
        per = {k: v / n for k, v in prof["ms"].items()}
        single = {k: v for k, v in per.items() if k not in MULTI_STAGES and v > 0} or per
        dom = max(single, key=single.get)
        Eb = sum(int(w.prep["assembly"].total_edit_rows) for w in subs) / len(subs)
        Kmax = max(int(w.prep["assembly"].max_K) for w in subs)
        roof = make_roof(per, frames / len(subs), samples / len(subs), B, hop, n_fft, executed_share(ctx), Eb, Kmax, frames, leg=config)
        ms = elapsed / steps * 1e3
        alg = (4 * B + 20 * hop) * frames
        return {"workload": f"BASELINE config {config} as ONE fixed job of {job_notes} notes on 1 GPU, sub-batches of {sub_batch} notes "
                            f"(longest first), sr {sr}, n_fft {n_fft}, hop {hop}; per-note flags {wl_flags(config)}",
                "value": frames * steps / elapsed, "unit": "frames/s", "realtime_factor": frames * steps / elapsed * hop / sr,
                "ms_per_step": ms, "steps": steps, "warmup": warmup, "sub_batches": len(subs), "notes": job_notes,
                "frames": frames, "samples": samples, "stage_ms": per,
                "stage_ms_from": "one bracketed pass after the timed ones: mean launch of every stage over the sub-batches",
                "roofline": roof(dom),
                "roofline_step": {"bound": "hbm", "alg_bytes_per_step": alg, "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_step_traffic(frames, config)}}
    finally:
        ctx.close()


def _tree_hash():
    from goofer_amd.build import source_hash
    return source_hash()


def _profile_file(kind, leg=None):
    """Newest committed counter file of the default workload (profiles/<tag>_<kind>) or of a fixed-job leg (<tag>_c4_<kind>,
    <tag>_c5_<kind>); tags sort in time order."""
    import glob
    import re
    pat = re.compile(r"^r\d+[a-z]*_" + ("c%d_" % leg if leg else "") + re.escape(kind) + "$")
    files = sorted(f for f in glob.glob(os.path.join(HERE, "profiles", "r*_" + kind)) if pat.match(os.path.basename(f)))
    return files[-1] if files else None


def pmc_traffic(stage, frames, leg=None, kernel=None):
    """{bytes, source, stale} from the newest committed rocprofv3 PMC passes of this same command (scripts/collect_profiles.sh),
    with the gfx950 FETCH_SIZE correction.  The counters cannot be read inside this process, so the figure is the committed
    measurement of a code tree, and only that tree's: the file carries the sha256 of goofer_amd/csrc it was measured on, and when
    it differs from the tree being timed (or the workload size does) the bytes are withheld and `stale` is true.  Per launch:
    a fixed job's kernels are launched once per sub-batch, and the file holds their mean launch."""
    try:
        fn = _profile_file("pmc_traffic.json", leg)
        d = json.load(open(fn))
        k = d["kernels"][kernel or STAGE_KERNEL[stage]]
        src = os.path.basename(fn)
        if d["_meta"].get("csrc_sha256") != _tree_hash() or d["_meta"]["frames"] != frames:
            return {"bytes": None, "source": src, "stale": True}
        return {"bytes": (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0, "source": src, "stale": False}
    except Exception:
        return {"bytes": None, "source": None, "stale": None}


def pmc_step_traffic(frames, leg=None):
    try:
        fn = _profile_file("pmc_traffic.json", leg)
        d = json.load(open(fn))
        src = os.path.basename(fn)
        if d["_meta"].get("csrc_sha256") != _tree_hash() or d["_meta"]["frames"] != frames:
            return {"bytes": None, "source": src, "stale": True}
        return {"bytes": d["_meta"]["step_hbm_bytes"], "source": src, "stale": False}
    except Exception:
        return {"bytes": None, "source": None, "stale": None}


def sq_valu_issue(stage, ms, frames, leg=None, kernel=None):
    """Vector-pipe fraction of a kernel from the newest committed SQ counter pass (profiles/r*_sq_counters.txt, collected with
    scripts/pmc_pass.sh on the 1024-note default workload; r*_c4_ / r*_c5_ for the fixed jobs): SQ_INSTS_VALU per launch / this
    run's kernel time, against what 1024 SIMDs can issue — one wave64 vector instruction per 2 cycles at 2.4 GHz
    (MI355X_MICROARCH.md; scripts/micro/valu_rates.hip measures one per 2.2 cycles at four waves per SIMD).  Vector
    instructions only: scalar, LDS and memory instructions do not take VALU issue turns.  The file's first line carries the
    sha256 of the kernel sources it was measured on; a different tree gets {"stale": true} and no number."""
    try:
        if leg is None and frames != 194560:
            return None
        fn = _profile_file("sq_counters.txt", leg)
        name = (kernel or STAGE_KERNEL[stage]).replace("void ", "").split("<")[0]
        lines = open(fn).read().splitlines()
        src = os.path.basename(fn)
        if not (lines and lines[0].startswith("# csrc_sha256=") and lines[0].split("=", 1)[1].strip() == _tree_hash()):
            return {"stale": True, "source": src}
        for ln in lines:
            if ln.startswith(name + " ") and "SQ_INSTS_VALU=" in ln:
                cnt = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in ln.split()[1:] if "=" in kv and kv.split("=")[0].startswith("SQ_INSTS_")}
                insts = cnt["SQ_INSTS_VALU"]
                return {"insts_valu_per_launch": insts, "achieved": insts / (ms * 1e-3) / 1e9, "unit": "G wave-instructions/s",
                        "peak_wave_insts_per_s": VALU_PEAK, "frac": insts / (ms * 1e-3) / VALU_PEAK, "source": src, "stale": False}
    except Exception:
        pass
    return None


def pcie_leg(wl, step_s):
    """What a host that hands over numpy buffers pays on top of `value` (never part of it): H2D of everything
    Renderer.prepare made resident (knots, masks, pitch curves, plans, taps, formant tracks) and D2H of the mix,
    through pinned staging buffers, timed with events around the copies."""
    import torch
    dev = [t for t in wl.prep["keep"].values() if torch.is_tensor(t)] + [wl.prep["formants"]]
    host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t) for t in dev]
    out = wl.step()
    mix_host = torch.empty(out["mix"].shape, dtype=out["mix"].dtype, pin_memory=True)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for h, d in zip(host, dev):
        d.copy_(h, non_blocking=True)
    e[1].record()
    mix_host.copy_(out["mix"], non_blocking=True)
    e[2].record()
    torch.cuda.synchronize()
    h2d_ms, d2h_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    h2d_b, d2h_b = sum(h.numel() * h.element_size() for h in host), mix_host.numel() * mix_host.element_size()
    tot = step_s + (h2d_ms + d2h_ms) * 1e-3
    return {"h2d_ms": h2d_ms, "d2h_ms": d2h_ms, "h2d_bytes": h2d_b, "d2h_bytes": d2h_b,
            "frames_per_s": wl.frames / tot, "note": "serial copies + compute, pinned host buffers; not the headline value"}


def _oracle_notes(config, ids, n_fft, hop, budget_s, min_notes):
    """Render notes `ids` of a BASELINE config with the oracle until the budget is spent: (frames, seconds, notes).  The
    synthetic notes are built before the clock starts; the first note warms tables / the native library and is not counted."""
    from oracle import goofer_ref as R
    from oracle import sampler_ref as SR
    from goofer_amd import synthetic as syn
    R._native()
    notes = [syn.config_note(config, int(i)) for i in ids]
    frames = done = 0
    t0 = time.perf_counter()
    for j, (src, req, phi_seed) in enumerate(notes):
        feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()},
                 src["sr"], src["y_len"])
        params = SR.decode_request(*syn.request_args(req))
        if j == 1:
            t0 = time.perf_counter()
        out = SR.render(feats, params, seed=phi_seed, n_fft=n_fft, hop=hop)
        if j >= 1:
            frames += 1 + len(out) // hop
            done += 1
        if done >= min_notes and time.perf_counter() - t0 > budget_s:
            break
    return frames, time.perf_counter() - t0, done


def _pool_worker(job):
    return _oracle_notes(*job)


def cpu_baseline(config, ids, n_fft, hop, budget_s=10.0, min_notes=4):
    """Oracle (CPU port of the reference path, oracle/) on the same notes — the full render the reference does per note:
    decode features, assemble, synthesize, mix (wav I/O excluded).  Leg (i): one worker process, numpy / BLAS threads = 1.
    Leg (ii): a pool of worker processes over notes, one per host core of this GPU's share (16 of the
    box's cores).  Workers are spawned, with single-threaded numpy, before this process touches the GPU."""
    import multiprocessing as mp
    ids = list(ids)
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"                    # inherited by the spawned workers (set before they import numpy)
    try:
        ctx = mp.get_context("spawn")
        per1 = min(len(ids), 1 + max(min_notes, int(budget_s * 70)))
        with ctx.Pool(1) as pool:
            frames, dt, done = pool.map(_pool_worker, [(config, ids[:per1], n_fft, hop, budget_s, min_notes)])[0]
        res = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{done} notes of the same workload ({frames} frames) through oracle/sampler_ref.render (numpy + "
                         f"gcc -O2 loops standing in for numba): knot decode, assembly, synthesize, mix; {dt:.1f} s, one process, "
                         "numpy threads = 1", "host_cpus": os.cpu_count()}
        workers = max(1, min(16, os.cpu_count() or 1))
        per = 1 + max(min_notes, int(np.ceil(budget_s * done / dt)))          # about budget_s of work per worker
        jobs = [(config, [ids[(w * per + k) % len(ids)] for k in range(per)], n_fft, hop, budget_s, min_notes) for w in range(workers)]
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            out = pool.map(_pool_worker, jobs)
        wall = time.perf_counter() - t0
        res["pool"] = {"value": sum(o[0] / o[1] for o in out), "unit": "frames/s", "cores": workers, "kind": "port",
                       "sample": f"{sum(o[2] for o in out)} notes ({sum(o[0] for o in out)} frames) over {workers} spawned worker "
                                 f"processes rendering side by side for {max(o[1] for o in out):.1f} s (sum of the workers' rates; "
                                 f"{wall:.1f} s wall with start-up and note synthesis)"}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def host_inclusive(wl, ctx, step_s):
    """What one batch costs a host that starts from the 13 argument strings and ends with the audio in host memory: decode,
    plan + tables + upload (Renderer.prepare), one device step, download of the mix into a pinned buffer.  `serial`: one batch
    after the other on one host thread.  `pipelined`: a second host thread prepares batch k + 1 (its planner and its copies
    run outside the interpreter lock) while batch k is on the device — the steady-state cost of a long job.  Never part of
    `value`."""
    import threading
    import torch
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.render import Source
    raw = wl.raw
    args = [syn.request_args(q) for _, q, _ in raw]
    srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
    ids = list(range(len(raw)))
    host_mix = torch.empty(wl.samples, dtype=torch.float32, pin_memory=True)

    def prepare():
        t0 = time.perf_counter()
        reqs = S.decode_request_batch(args)
        t1 = time.perf_counter()
        prep = wl.renderer.prepare((srcs, reqs), note_ids=ids)
        return prep, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)

    def run(prep):
        out = wl.renderer.run(prep, seed=0)
        host_mix.copy_(out["mix"], non_blocking=True)
        torch.cuda.synchronize()

    from goofer_amd.render import SourceArena
    _warm = prepare()                                          # (a free staging block exists from here on: the workload's own batch holds one)
    run(_warm[0])                                              # ... and the allocator holds the outputs' blocks: first_batch measures the sample uploads,
    del _warm                                                  # not four hipMallocs of 200 MB (25 ms on some boxes)
    # "first": none of the batch's voicebank samples resident (fresh Source objects — loading them is the frontend's file read,
    # outside the timer — and an empty arena); three trials, the fastest reported and all three listed: on some boxes the first
    # trial spends 60 ms in the allocator growing the arena's two device arrays, which later trials find cached.
    def one_pass():
        t0 = time.perf_counter()
        prep, dec_ms, prep_ms = prepare()
        t2 = time.perf_counter()
        run(prep)
        t3 = time.perf_counter()
        return ({"decode_ms": dec_ms, "plan_upload_ms": prep_ms, "step_and_download_ms": 1e3 * (t3 - t2),
                 "total_ms": 1e3 * (t3 - t0), "frames_per_s": prep["frames"] / (t3 - t0), "notes_per_s": len(raw) / (t3 - t0)}, prep["frames"])

    import gc
    gc.collect()
    gc.freeze()                                                # a render server's usual setting: the objects of the set-up (thousands of
                                                               # sources and plans in this process) are not traversed by later collections
                                                               # (an unfrozen full collection in the middle of a batch is 50-100 ms here)
    firsts = []
    for _ in range(3):
        srcs[:] = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
        wl.renderer.sources = SourceArena(ctx)
        firsts.append(one_pass()[0])
    first = min(firsts, key=lambda c: c["total_ms"])
    best = None
    for k in range(3):                                         # the samples resident from here on
        cur, frames = one_pass()
        if best is None or cur["total_ms"] < best["total_ms"]:
            best = cur
    best["first_batch"] = {"total_ms": first["total_ms"], "plan_upload_ms": first["plan_upload_ms"], "frames_per_s": first["frames_per_s"],
                           "trials_total_ms": [c["total_ms"] for c in firsts],
                           "note": "the same batch when none of its 1024 voicebank samples is resident in HBM yet (knot tables + voicing masks "
                                   "uploaded): the fastest of three trials, each with fresh Source objects and an empty arena"}
    # Long jobs: goofer_amd.render.PipelinedRenderer — two handles / streams, batches decoded and planned on worker threads while
    # the previous ones render, the mix of batch k - 1 crossing PCIe under step k.  The same 1024 argument lists and sources as
    # batch after batch (the sources resident, as in a job that renders a voicebank's samples thousands of times).
    from goofer_amd.render import PipelinedRenderer
    rounds, lead, total = 32, 24, 120                          # one job of 120 batches per setting; reported: the MEAN over everything
                                                               # behind the first 24 batches (a job's first dozens run 4-8 ms: allocator
                                                               # pools filling, threads falling into step); the best window of 32
                                                               # consecutive batches rides along as a secondary figure only

    def pipelined(coalesce):
        pipe = PipelinedRenderer(torch.cuda.current_device(), hop=wl.geo["hop"], depth=2, workers=4, coalesce=coalesce)
        try:
            def job(pcm16):
                stamps = []
                for mix, off in pipe.render_iter(((srcs, args) for _ in range(total)), seed=0, note_ids=lambda k, n: ids, pcm16=pcm16):
                    stamps.append(time.perf_counter())
                assert len(stamps) == total and float(np.abs(mix).max()) > 0.0
                s = np.asarray(stamps[lead - 1:])
                win = float(np.min(s[rounds:] - s[:-rounds])) / rounds
                return win, float(s[-1] - s[0]) / (s.size - 1)
            w32, m32 = job(False)
            w16, m16 = job(True)
        finally:
            pipe.close()
        return {"coalesce": coalesce, "ms_per_batch": 1e3 * m32, "frames_per_s": frames / m32, "best_window_ms_per_batch": 1e3 * w32,
                "pcm16": {"ms_per_batch": 1e3 * m16, "frames_per_s": frames / m16, "job_mean_ms_per_batch": 1e3 * m16,
                          "job_mean_frames_per_s": frames / m16, "best_window_ms_per_batch": 1e3 * w16}}

    try:
        p4 = pipelined(4)
        p1 = pipelined(1)
    finally:
        gc.unfreeze()
    p4["job_batches"] = p1["job_batches"] = total - lead
    p4["job_mean_ms_per_batch"], p4["job_mean_frames_per_s"] = p4["ms_per_batch"], p4["frames_per_s"]
    p4["uncoalesced"] = p1
    p4["note"] = ("goofer_amd.render.PipelinedRenderer(depth=2, workers=4): a job of 120 batches of 1024 notes, 13 argument strings -> audio in "
                  "pinned host memory; two handles / streams with three batches in flight, decode + planning of the next batches on four "
                  "worker threads, D2H of the previous mix on a copy stream under the running step.  ms_per_batch / frames_per_s = the MEAN "
                  "over the job behind its first 24 batches (best_window_ms_per_batch: the best 32 consecutive batches, for reference "
                  "only).  coalesce=4: four caller batches planned and rendered as one device batch and handed back one by one (first audio "
                  "four batches later); `uncoalesced`: the same job with coalesce=1.  pcm16: the mix converted to the wav's int16 samples "
                  "on the device (goofer_pcm16: what the reference's PCM_16 file holds), half the bytes over PCIe")
    best["pipelined"] = p4
    best["note"] = ("serial, one host thread: 13 argument strings -> request columns (decode_request_batch), plans written by the library's "
                    "host planner into a pinned staging block + tables, one H2D copy (Renderer.prepare; the voicebank samples are resident "
                    "in HBM, see first_batch), device step, D2H of the mix into pinned memory; the best of three passes; the device step "
                    "alone is ms_per_step")
    return best


def launch_ranks(n):
    """`bench.py --gpus N` (N > 1) started WITHOUT a launcher: run the same command under torch.distributed.run, one rank per
    GPU, as a CHILD process and exit with its status.  This process has not touched the GPU (nothing here imports torch), and it
    never replaces itself: the child is spawned, not exec'd."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: starting %s\n" % (n, " ".join(cmd)))
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--notes", type=int, default=1024, help="notes per GPU (weak scaling)")
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (1-based)")
    ap.add_argument("--job-notes", type=int, default=0, help="a FIXED job of this many notes (BASELINE configs 4 / 5): sharded over the "
                    "ranks by longest-processing-time assignment, rendered in sub-batches; reports strong scaling")
    ap.add_argument("--sub-batch", type=int, default=4096, help="notes per device batch of a fixed job (4096: the pulse walk runs one wave per note, four rounds of resident waves at most; "
                    "smaller batches fill the device worse, one 10 000-note batch makes the walk the critical path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the skip_zero-off / 30 %%-unvoiced variants of the step")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="goofer_set_option on the handle before anything runs (e.g. skip_zero=0); the line carries them as `options` and is not the headline configuration")
    ap.add_argument("--gather", action="store_true", help="also time the optional ragged gather of finished audio to rank 0 "
                    "(RCCL over xGMI; never part of `value`)")
    ap.add_argument("--rehearse", action="store_true", help="dress rehearsal of the multi-rank logic WITHOUT a GPU: gloo process "
                    "group, the same assignment / sub-batching / planning / reductions, a sleep standing in for the device step")
    ap.add_argument("--no-legs", action="store_true", help="skip the config4 / config5 legs of the default line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    if args.rehearse:
        return rehearse(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")
    # The CPU baseline runs first, before this process touches the GPU: its pool leg starts worker processes, and nothing
    # that holds a device context should be forked / exec'd from.
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline:
        from goofer_amd import synthetic as syn0
        g0_ = syn0.config_geometry(args.config)
        n_ids = args.job_notes if args.job_notes > 0 else args.notes
        cpu_line = cpu_baseline(args.config, list(range(n_ids)), g0_["n_fft"], g0_["hop"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload

    from goofer_amd.shard import assign_lpt, note_range, reduce_timing

    ctx = Context(local)
    options = {}
    for kv in args.opt:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))
        options[name] = int(val)
    job = args.job_notes > 0
    if job:
        # every rank derives the same assignment from the frame counts of the whole job: no communication
        est = [syn.config_note_frames(args.config, i) for i in range(args.job_notes)]
        ids = assign_lpt(est, world)[rank]
        # sub-batches of similar lengths: the pulse walk of a sub-batch takes as long as its longest note (one wave per note,
        # sequential in time), which then hides behind that sub-batch's own envelope assembly instead of a 3 s note stalling
        # a sub-batch of short ones
        ids = sorted(ids, key=lambda i: (-est[i], i))
        subs = [SamplerWorkload(ctx, args.config, ids[k:k + args.sub_batch]) for k in range(0, len(ids), args.sub_batch)]
    else:
        ids = list(note_range(rank, world, args.notes))
        subs = [SamplerWorkload(ctx, args.config, ids)]
    wl = subs[0]
    my_frames, my_samples = sum(w.frames for w in subs), sum(w.samples for w in subs)
    geo = wl.geo
    B, hop, n_fft, sr = geo["n_fft"] // 2 + 1, geo["hop"], geo["n_fft"], geo["sr"]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def step():
        out = None
        for w in subs:
            out = w.step()
        return out

    for _ in range(args.warmup):
        step()
    # Which kernel the roofline is about = the longest single-kernel stage of one more untimed step with every stage bracketed.
    # The timed steps then carry that stage's two HIP events only: the ~20 event records of the full breakdown are 0.06 ms of a
    # 2.3 ms step (scripts/prof_cost.py).  The breakdown itself (stage_ms) comes from a further untimed pass of the same steps.
    ctx.profile_begin(len(subs))
    step()
    pre = ctx.profile_end()
    dominant = max((k for k, v in pre["ms"].items() if k not in MULTI_STAGES and v > 0), key=lambda k: pre["ms"][k], default=None)
    if dominant is not None:
        ctx.profile_only(dominant)
    barrier()
    ctx.profile_begin(args.steps * len(subs))
    scan0 = (ctx.counter("pulse_scanned_notes"), ctx.counter("pulse_fallback_notes"))
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    timed = ctx.profile_end()
    scan1 = (ctx.counter("pulse_scanned_notes"), ctx.counter("pulse_fallback_notes"))
    ctx.profile_only(None)
    ctx.profile_begin(args.steps * len(subs))
    for k in range(args.steps):
        step()
    prof = ctx.profile_end()                                   # every stage, untimed pass
    if dominant is not None:
        prof["ms"][dominant] = timed["ms"][dominant] * prof["steps"] / max(1, timed["steps"])   # ... the dominant one from the timed steps
    # the timed path must have produced audio: finite, not silent (the last sub-batch of the last step)
    mix = last["mix"]
    assert bool(torch.isfinite(mix).all()) and float(mix.abs().max()) > 0.0, "the timed steps produced no valid audio"
    # the assembly call on its own (the synth stages overlap on two streams, so step - sum(stages) no longer isolates it)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(args.steps):
        wl.renderer.assemble(wl.prep)
    a1.record()
    torch.cuda.synchronize()
    prof["ms"]["assemble"] = a0.elapsed_time(a1) / args.steps * max(1, prof["steps"])
    elapsed, frames_total = reduce_timing(elapsed, my_frames, device="cuda")
    per_rank = [my_frames]
    if world > 1:
        t = torch.tensor([float(my_frames)], dtype=torch.float64, device="cuda")
        allf = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allf, t)
        per_rank = [int(round(v.item())) for v in allf]

    # the framewise rFFT kernel on its own (the kernel BASELINE's >= 40 % HBM target names): same frames and
    # CSR geometry as the batch, HIP events on the launch stream (k_frame_note, ~5 us, rides along)
    o = wl.prep["offsets"]
    xin = torch.randn(wl.samples, device="cuda")
    Sout = torch.empty((wl.frames, (B + 15) & ~15), dtype=torch.complex64, device="cuda")      # 128-byte aligned rows
    for _ in range(2):
        ctx.rfft_frames(xin, o["d_s"], o["d_f"], wl.frames, out=Sout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        ctx.rfft_frames(xin, o["d_s"], o["d_f"], wl.frames, out=Sout)
    e1.record()
    torch.cuda.synchronize()
    rfft_ms = e0.elapsed_time(e1) / args.steps
    del xin, Sout

    # The headline workload is fully voiced behind the notes' 50 ms offset, so the noise walker skips the unvoiced stem's
    # transform (exactly zero gain) on every frame.  Two variants of the same step say what that is worth: the skipping
    # switched off (both transforms on every frame: SURVEY 8d's "1 rFFT + 3 irFFT-OLA" literally), and 30 % of every source
    # unvoiced in 50 ms gaps.  Same notes, same flags, same step function; `value` stays the BASELINE workload.
    variants = None
    if world == 1 and not job and not args.no_variants:
        def timed(w, k):
            best = float("inf")
            for _ in range(2):                                 # the better of two passes (a first pass after a change of option /
                for _ in range(3):                             # workload has been seen to run a third slower once)
                    w.step()
                torch.cuda.synchronize()
                v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                v0.record()
                for _ in range(k):
                    w.step()
                v1.record()
                torch.cuda.synchronize()
                best = min(best, v0.elapsed_time(v1) / k)
            return best
        ctx.set_option("skip_zero", 0)
        ms_off = timed(wl, args.steps)
        ctx.set_option("skip_zero", 1)
        wl30 = SamplerWorkload(ctx, args.config, ids, unvoiced_share=0.3)
        ms_30 = timed(wl30, args.steps)
        voiced30 = float((wl30.prep["f0"] > 0).float().mean())
        variants = {"skip_zero_off": {"value": my_frames / (ms_off * 1e-3), "ms_per_step": ms_off,
                                      "what": "option skip_zero = 0: no transform is skipped (one rFFT + three irFFT-OLA on every frame)"},
                    "unvoiced_30pct": {"value": wl30.frames / (ms_30 * 1e-3), "ms_per_step": ms_30, "voiced_share_of_samples": voiced30,
                                       "what": "30 % of every source unvoiced in 50 ms gaps (synthetic.with_unvoiced_gaps), skipping on"}}
        del wl30

    # Two batches in flight: consecutive launches alternate between two handles (own scratch, own side streams) on two streams,
    # so the f0 / envelope assembly of one batch runs beside the walkers and the gain pass of the other — how a job of many
    # batches is driven for throughput (the latency of a batch and `value` stay those of one batch at a time).  For a fixed job
    # the sub-batches of a pass alternate; for the weak-scaling workload whole steps do.
    if world == 1 and not args.no_variants:
        ctx_b = Context(local)
        for name, val in options.items():
            ctx_b.set_option(name, val)
        if job:
            subs_b = [SamplerWorkload(ctx_b, args.config, ids[k:k + args.sub_batch]) for k in range(0, len(ids), args.sub_batch)]
        else:
            subs_b = [SamplerWorkload(ctx_b, args.config, ids)]
        both = [subs, subs_b]                                  # a handle stays on its stream: launch i -> handle i % 2, sub-batch i % per_pass
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def launches(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % 2]):
                    both[i % 2][i % len(subs)].step()

        per_pass = len(subs)                                   # launches of one step (one pass over this rank's notes)
        n_launch = per_pass * args.steps
        n_launch += n_launch % 2
        best = float("inf")
        for _ in range(2):
            launches(2 * len(subs))
            torch.cuda.synchronize()
            v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            v0.record()
            for st_ in streams:
                st_.wait_stream(torch.cuda.current_stream())
            launches(n_launch)
            for st_ in streams:
                torch.cuda.current_stream().wait_stream(st_)
            v1.record()
            torch.cuda.synchronize()
            best = min(best, v0.elapsed_time(v1) / n_launch * per_pass)
        variants = variants or {}
        variants["two_in_flight"] = {"value": my_frames / (best * 1e-3), "ms_per_step": best,
                                     "what": "the same launches alternating between two handles on two streams (two batches resident, twice the scratch): "
                                             "throughput of a long job; not the headline, which runs one batch at a time"}
        del subs_b, both
        ctx_b.close()

    gather_ms = None
    if args.gather:
        from goofer_amd.shard import gather_audio
        out = wl.step()
        barrier()
        g0 = time.perf_counter()
        got = gather_audio(out["mix"], wl.prep["lens"], dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert sum(int(a.numel()) for a, _ in got) >= wl.samples
        del got

    if rank == 0:
        value = frames_total * args.steps / elapsed
        steps = max(1, prof["steps"])
        per = {k: v / steps for k, v in prof["ms"].items()}              # ms per launch (per sub-batch), this rank
        # the dominant kernel = the longest single kernel of the step.  (Stages bracketing several launches — the assembly, the
        # map kernels — and the latency-bound phase scan, which moves 4 B per sample by design, are not roofline candidates.)
        # Some stages share the chip with the side stream: their event time includes that, see `shared_with`.
        single = {k: v for k, v in per.items() if k not in MULTI_STAGES and v > 0} or per
        dom = max(single, key=single.get)
        per["rfft_frames_standalone"] = rfft_ms
        Fb, Nb = my_frames / len(subs), my_samples / len(subs)           # frames / samples per launch (mean sub-batch)
        Eb = sum(int(w.prep["assembly"].total_edit_rows) for w in subs) / len(subs)   # edited source rows per launch
        Kmax = max(int(w.prep["assembly"].max_K) for w in subs)
        executed = executed_share(ctx)
        roof = make_roof(per, Fb, Nb, B, hop, n_fft, executed, Eb, Kmax, wl.frames, leg=None, counters=not job)

        step_ms = elapsed / args.steps * 1e3
        alg_step = (4 * B + 20 * hop) * frames_total / world            # SURVEY 8d ALG_BYTES_FRAME x frames of one rank's step
        if job:
            shape = (f"BASELINE config {args.config} as ONE fixed job of {args.job_notes} notes sharded over {world} GPU(s) by "
                     f"longest-processing-time assignment on frame counts, sub-batches of {args.sub_batch} notes")
        else:
            shape = f"BASELINE config {args.config}: {args.notes} notes/GPU x ~1.1 s"
        line = {
            "metric": "resynth_frames_per_sec", "value": value, "unit": "frames/s",
            "realtime_factor": value * hop / sr,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "strong" if job else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{shape}, sr {sr}, n_fft {n_fft}, "
                                   f"hop {hop} ({1e3 * hop / sr:.1f} ms); per-note flags {wl_flags(args.config)}; one step = "
                                   "goofer_render_batch = goofer_assemble_batch + goofer_synth_batch (SillySampler.resample + gf.synthesize + V/B/U mix) "
                                   "from .goofy features and host-made plans resident in HBM, on-device Philox phases",
                       "notes_per_gpu": len(ids), "frames_per_gpu": my_frames, "samples_per_gpu": my_samples,
                       "sub_batches_per_gpu": len(subs),
                       "sharding": "independent notes, no data-path collective"},
            "stage_ms": per,
            "stage_ms_from": ("HIP events on the launch streams: `%s` (the roofline's kernel) from the timed steps, which carry its two "
                              "events only; the other stages from an untimed pass of the same %d steps with every stage bracketed "
                              "(twenty event records per step: 0.06 ms of it)" % (dominant, args.steps)),
            # pulse onsets (GOOFER.py:487-493): notes of the timed steps settled by the parallel phase scan, and those it had to
            # walk sequentially (a phase within the scan's rounding band of an integer) — rank 0's share
            "pulse_scan": {"notes": scan1[0] - scan0[0], "fallback_notes": scan1[1] - scan0[1]},
            "roofline": roof(dom),                # the longest kernel of the step (the noise walker on the default workload)
            "roofline_noise": roof("noise_stems") if per.get("noise_stems", 0) > 0 else None,
            "roofline_harm": roof("harm_stem") if per.get("harm_stem", 0) > 0 else None,
            # in-pipeline launch when the active path has a standalone rFFT stage, else the entry-point timing
            "roofline_fft": roof("rfft_frames", wl.frames, wl.samples) if per.get("rfft_frames", 0) > 0
            else roof("rfft_frames_standalone", wl.frames, wl.samples),
            # the whole step against the end-to-end algorithmic bytes (4 B + 20 hop per frame) and, from the committed counter
            # passes of this command, the HBM bytes all its kernels really moved
            "roofline_step": {"bound": "hbm", "alg_bytes_per_step": alg_step, "achieved": alg_step / (step_ms * 1e-3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic": None if job else pmc_step_traffic(wl.frames),
                              "note": "the stem walkers are latency bound at two / three waves per SIMD (about 30 flop per algorithmic "
                                      "byte): a quarter of the HBM peak and a quarter of the vector pipe each, DESIGN.md section 3"},
        }
        if job or world > 1:
            line["per_rank_frames"] = per_rank
            line["imbalance"] = max(per_rank) / (sum(per_rank) / len(per_rank))
        if gather_ms is not None:
            line["gather_to_rank0"] = {"ms": gather_ms, "bytes": 4 * wl.samples * (world - 1), "note": "ragged gather of the finished "
                                       "notes (goofer_amd.shard.gather_audio), outside the timed steps"}
        if variants:
            if "skip_zero_off" in variants:
                # what `value` counts, in the config text itself: SURVEY 8(d) defines a frame as 1 rFFT + 3 irFFT-OLA; the timed
                # step skips the transforms whose stem gain is exactly 0 (bit-identical output)
                line["config"]["transforms_per_frame"] = (
                    "value: option skip_zero on — on this workload (every rendered sample voiced) the unvoiced stem's irFFT is skipped on "
                    "every frame, so 1 rFFT + 2 irFFT-OLA run per frame, output bit-identical; with every transform executed "
                    "(1 rFFT + 3 irFFT-OLA per frame, SURVEY 8d literally) the same step gives %.4g frames/s (value_skip_zero_off)"
                    % variants["skip_zero_off"]["value"])
                line["value_skip_zero_off"] = variants["skip_zero_off"]["value"]
                line["value_unvoiced_30pct"] = variants["unvoiced_30pct"]["value"]
            line["value_two_in_flight"] = variants["two_in_flight"]["value"]

            line["variants"] = variants
        if world == 1 and not job:
            line["pcie_inclusive"] = pcie_leg(wl, elapsed / args.steps)
            if not args.no_host_inclusive:
                line["host_inclusive"] = host_inclusive(wl, ctx, elapsed / args.steps)
        if options:
            line["options"] = options
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        # BASELINE configs 4 and 5 are 8-GPU jobs: their single-GPU passes ride on the default line, after the timed region,
        # each on a handle of its own (the default workload's buffers are released first)
        if world == 1 and not job and args.config == 3 and not args.no_legs and not options:
            del last, mix, subs, wl
            ctx.close()
            torch.cuda.empty_cache()
            line["config4"] = job_leg(local, 4, 10000, args.sub_batch, steps=5, warmup=2)
            torch.cuda.empty_cache()
            line["config5"] = job_leg(local, 5, 1024, args.sub_batch, steps=5, warmup=2)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def rehearse(args):
    """`bench.py --rehearse`: everything a rank does around the device work, on CPUs under gloo — the note assignment (weak:
    note_range, fixed job: LPT over planner frame counts, notes ordered by length, sub-batches), the host set-up of every
    sub-batch (synthetic sources, 13-argument decode, index plans: what SamplerWorkload does before its first upload), the
    barrier / MAX-elapsed / SUM-frames reductions, the per-rank gather and the JSON line.  The device step is a sleep of
    frames / 60e6 s.  Prints the set-up seconds of every rank, so that the first real 8-GPU run does not discover its Python
    costs in front of the barrier.  No GPU is touched; `value` of this line means nothing."""
    import torch
    import torch.distributed as dist
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.render import Source
    from goofer_amd.shard import assign_lpt, note_range, reduce_timing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    geo = syn.config_geometry(args.config)
    hop = geo["hop"]
    job = args.job_notes > 0
    t_setup = time.perf_counter()
    if job:
        est = [syn.config_note_frames(args.config, i) for i in range(args.job_notes)]
        ids = sorted(assign_lpt(est, world)[rank], key=lambda i: (-est[i], i))
        subs = [ids[k:k + args.sub_batch] for k in range(0, len(ids), args.sub_batch)]
    else:
        ids = list(note_range(rank, world, args.notes))
        subs = [ids]
    t_assign = time.perf_counter() - t_setup
    sub_frames = []
    for sub in subs:                                           # host half of SamplerWorkload.__init__ / Renderer.prepare
        jobs = []
        for i in sub:
            src, req, _ = syn.config_note(args.config, int(i))
            sobj = Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"])
            jobs.append((S.decode_request(*syn.request_args(req)), sobj.sr, sobj.ylen, sobj.knots.shape[1], sobj.formants))
        planned = S.plan_notes_arrays(jobs, hop, trim_rows=True)          # the library's host planner (pure CPU code)
        sub_frames.append(int((1 + planned.geo["n_out"].astype(np.int64) // hop).sum()))
    setup_s = time.perf_counter() - t_setup
    my_frames = sum(sub_frames)
    if job:
        assert my_frames == sum(est[i] for i in ids), "the planner and the assignment disagree on frame counts"

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for f in sub_frames:
            time.sleep(f / 60e6)
    t1 = time.perf_counter()
    barrier()
    elapsed, frames_total = reduce_timing(t1 - t0, my_frames)
    per_rank, setups = [my_frames], [setup_s]
    if world > 1:
        t = torch.tensor([float(my_frames), setup_s, t_assign], dtype=torch.float64)
        allf = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allf, t)
        per_rank = [int(round(v[0].item())) for v in allf]
        setups = [float(v[1].item()) for v in allf]
    if rank == 0:
        line = {"rehearsal": True, "metric": "resynth_frames_per_sec", "value": frames_total * args.steps / elapsed, "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong" if job else "weak", "backend": "gloo (no GPU: the device step is a sleep of frames / 60e6 s)",
                "config": {"workload": f"BASELINE config {args.config}" + (f" as ONE fixed job of {args.job_notes} notes" if job else f", {args.notes} notes per rank"),
                           "sub_batch": args.sub_batch, "sub_batches_rank0": len(subs)},
                "per_rank_frames": per_rank, "imbalance": max(per_rank) / (sum(per_rank) / len(per_rank)),
                "setup_seconds_per_rank": [round(v, 3) for v in setups],
                "setup_note": "assignment + synthetic sources + 13-argument decode + index plans of every sub-batch, one host thread per rank; "
                              "uploads and the first device step come on top on the GPU box"}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def wl_flags(config):
    from goofer_amd import synthetic as syn
    return syn.config_flags(config, 0) + " / " + syn.config_flags(config, 1)


if __name__ == "__main__":
    main()
