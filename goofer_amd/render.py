"""Batch renderer: UTAU note requests -> audio, entirely on the MI355X.

``Renderer.render(jobs)`` takes any number of (source features, 13-argument request) pairs, plans them on
the host (``sampler.plan_note``: scalars, index plans, few-hundred-value tracks), uploads the plans and
runs ``goofer_render_batch``: ``goofer_assemble_batch`` (SillySampler.resample up to the synthesize call) and
``goofer_synth_batch`` (gf.synthesize + V/B/U mix) as one call.  ``GooferResampler`` keeps the reference's
construct-to-render, one-note call surface (SillySampler.py:285-413) on top of it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import sampler as S
from .device import Context, default_context, default_params, row_stride


@dataclass
class Source:
    """Features of one voicebank sample, as stored in ``<stem>_features.goofy`` (knots mode, or the dense 'full'
    mode: then ``knots`` is the fp16 envelope itself, K = bins, and ``hz_knots`` is None)."""
    knots: np.ndarray        # fp16 [K, T] log-envelope knot values (reference layout)
    hz_knots: np.ndarray     # fp32 [K]
    mask: np.ndarray         # fp32 [ylen] voicing mask
    formants: dict           # {1..4: [T]} Hz
    sr: int
    ylen: int
    n_fft: int = S.N_FFT

    def knot_rows(self) -> np.ndarray:
        """The knot table frame-major ([T, K] contiguous, flattened), the layout the device kernels index; made once per source
        (a voicebank sample is rendered many times)."""
        kr = getattr(self, "_knot_rows", None)
        if kr is None:
            kr = np.ascontiguousarray(self.knots.T).reshape(-1)
            object.__setattr__(self, "_knot_rows", kr)
        return kr

    @staticmethod
    def from_pack(env_pack, f0, mask, formants, sr, ylen):
        """Features as load_features returns them.  Everything the device kernels index by these sizes is checked here —
        a short mask, a knot table that disagrees with its frequencies or an envelope with the wrong frame count would
        otherwise shift every later note's offsets in the batch and read out of bounds on the device."""
        sr, ylen = int(sr), int(ylen)
        mask = np.asarray(mask, dtype=np.float32)
        if sr <= 0 or ylen <= 0:
            raise ValueError(f"bad features: sr {sr}, y_len {ylen}")
        if mask.ndim != 1 or mask.size < ylen:
            raise ValueError(f"bad features: voicing mask has {mask.size} samples, y_len is {ylen}")
        if not (isinstance(env_pack, dict) and env_pack.get("mode") == "knots"):
            env = np.asarray(env_pack)                        # dense [bins, T]: stored fp16, computed fp32 (GOOFER.py:306-333)
            if env.ndim != 2 or env.shape[0] < 3 or env.shape[1] < 1:
                raise ValueError("features must be a knots dict or a [bins, T] envelope")
            src = Source(env.astype(np.float16), None, mask, formants, sr, ylen, 2 * env.shape[0] - 2)
        else:
            knots = np.asarray(env_pack["knot_vals_log"], dtype=np.float16)
            hz = np.asarray(env_pack["hz_knots"], dtype=np.float32)
            if knots.ndim != 2 or hz.ndim != 1 or knots.shape[0] != hz.size or hz.size < 2 or knots.shape[1] < 1:
                raise ValueError(f"bad features: knot table {knots.shape} against {hz.size} knot frequencies")
            if int(env_pack["n_bins"]) != int(env_pack["n_fft"]) // 2 + 1:
                raise ValueError("bad features: n_bins does not match n_fft")
            src = Source(knots, hz, mask, formants, sr, ylen, int(env_pack["n_fft"]))
        if not isinstance(formants, dict):
            raise ValueError("bad features: formants must be a dict of tracks")
        for k, v in formants.items():
            a = np.asarray(v)
            if a.ndim != 1 or not np.issubdtype(a.dtype, np.number):
                raise ValueError(f"bad features: formant track {k!r} is not a numeric vector")
        for k in (1, 2, 3, 4):                                # the planner takes F1..F4 by position in the sorted key list:
            if k not in formants:                             # a dict without one of them would silently shift the others
                raise KeyError(k)                             # (gf.synthesize raises the same KeyError, GOOFER.py:1000)
        return src


def _lerp_plan(sr, n_fft, hz):
    """2-tap lerp of precompute_interp_matrix (GOOFER.py:84-90) in fp32."""
    f = np.fft.rfftfreq(n_fft, 1.0 / sr).astype(np.float32)
    K = len(hz)
    idx = np.clip(np.searchsorted(hz, f, side="right") - 1, 0, K - 2)
    x0, x1 = hz[idx], hz[idx + 1]
    w1 = ((f - x0) / np.maximum(x1 - x0, 1e-12)).astype(np.float32)
    return idx.astype(np.int32), (1.0 - w1).astype(np.float32), w1


def _tilt(sr, n_bins, brightness_env):
    """br flag curve (SillySampler.py:506-510): fp64 power of an fp32 ramp, mean-normalised, cast to fp32."""
    fr = np.linspace(1e-6, sr * 0.5, n_bins, dtype=np.float32)
    nf = np.clip(fr / (sr * 0.5), 0.02, 1.0)
    t = nf ** np.clip(brightness_env - 1.0, -0.9, 1.0)
    t /= (t.mean() + 1e-12)
    return t.astype(np.float32)


def _fw_plan(n_bins, amount):
    """fw flag (SillySampler.py:555-564): bins stretched about the centre, clipped; (lo, hi, frac)."""
    bins = np.arange(n_bins, dtype=np.float64)
    c = n_bins / 2.0
    w = np.clip((bins - c) * (1.0 + amount) + c, 0, n_bins - 1)
    lo = np.floor(w).astype(int)
    return lo.astype(np.int32), np.minimum(lo + 1, n_bins - 1).astype(np.int32), (w - lo)


class Renderer:
    def __init__(self, ctx: Context | None = None, hop: int = S.HOP):
        self.ctx = ctx or default_context()
        self.hop = hop

    def render(self, jobs, seed: int = 0, phi_seeds=None, return_parts: bool = False):
        """jobs: list of (Source, Request).  Returns a list of fp32 arrays (the mix the reference writes to
        out.wav); with ``return_parts`` also a dict of device-side intermediates for tests.
        ``phi_seeds``: per-note seeds for INJECTED phases (parity with a seeded reference run); otherwise the
        device draws phases from Philox keyed by ``seed`` and the note index."""
        if not jobs:
            return []
        prep = self.prepare(jobs, phi_seeds=phi_seeds, trim_rows=not return_parts)   # tests look at the whole assembled envelope
        out = self.run(prep, seed=seed, keep_stems=return_parts)
        self.ctx.check()                                   # synchronises; raises if the device flagged a note
        mix = out["mix"].cpu().numpy()
        offs = prep["sample_off"]
        res = [mix[offs[i]:offs[i + 1]] for i in range(len(jobs))]
        if return_parts:
            parts = {"env": prep["env"], "f0": prep["f0"], "mask": prep["mask"], "env_off": prep["env_off"], "sample_off": offs,
                     "plans": prep["plans"], "stems": out}
            return res, parts
        return res

    def assemble(self, prep):
        """goofer_assemble_batch alone (asynchronous): envelope rows, f0 and voicing mask of the batch."""
        ctx = self.ctx
        ctx._check(ctx.lib.goofer_assemble_batch(ctx.h, C.byref(prep["assembly"]), ctx._stream()))

    def run(self, prep, seed: int = 0, keep_stems: bool = False, split: bool = False):
        """The device work of one batch (asynchronous): goofer_render_batch, i.e. assembly + synthesis as one call;
        ``split=True`` issues goofer_assemble_batch and goofer_synth_batch separately (same results)."""
        ctx = self.ctx
        par = prep["params"]
        # Notes with the 'sg' pulse layer or the 'sr' volume jitter are synthesised by the one-kernel-per-step pipeline (those
        # layers edit the pulse train / the stems between its steps), everything else by the stem walkers — and the library
        # picks the pipeline per BATCH.  So that a note renders to the same bits whatever company it keeps (the two pipelines
        # agree to fp32 rounding, not to the bit: the walkers fold the voiced frames' bin blur into the synthesis window), a
        # mixed batch is synthesised as two: one per pipeline.
        slow = np.nonzero((par["subharm_weight"] > 0) | (par["vol_jitter_harm"] != 0) | (par["vol_jitter_breath"] != 0))[0]
        if 0 < slow.size < len(par):
            self.assemble(prep)
            out = self._synth_partitioned(prep, [np.setdiff1d(np.arange(len(par)), slow), slow], seed, keep_stems)
        else:
            if split:
                self.assemble(prep)
            out = ctx.synth_batch(prep["env"], prep["env_lens"], prep["f0"], prep["mask"], prep["lens"], par,
                                  formants=prep["formants"], phi=prep["phi"], seed=seed, want_rec=False, want_mix=True,
                                  offsets=prep["offsets"], noise_f0=prep["noise_f0"], noise_vol=prep["noise_vol"],
                                  subharm=S.SUBHARM if prep["subharm"] else None,
                                  mix_only=prep["post"] is None and not keep_stems,
                                  assembly=None if split else prep["assembly"])
        if prep["post"] is not None:
            self._post_chain(prep, out, seed)
        return out

    def _synth_partitioned(self, prep, groups, seed, keep_stems):
        """goofer_synth_batch once per group of notes of an assembled batch; the stems land at the notes' places."""
        ctx = self.ctx
        so = prep["sample_off"]
        total = int(so[-1])
        out = {k: torch.empty(total, dtype=torch.float32, device=ctx.device) for k in ("harm", "uv", "bre", "mix")}
        cat = lambda t, idxs: None if t is None else torch.cat([t[int(so[i]):int(so[i + 1])] for i in idxs])
        for idxs in groups:
            idxs = [int(i) for i in idxs]
            sub = self._subset(prep, idxs)
            par = prep["params"][idxs].copy()
            lens, env_lens = [prep["lens"][i] for i in idxs], [prep["env_lens"][i] for i in idxs]
            nv = prep["noise_vol"]
            wants_sub = bool(np.any(par["subharm_weight"] > 0))
            wants_vol = nv is not None and bool(np.any((par["vol_jitter_harm"] != 0) | (par["vol_jitter_breath"] != 0)))
            o = ctx.synth_batch(sub["env"], env_lens, sub["f0"], sub["mask"], lens, par, formants=sub["formants"], phi=sub["phi"],
                                seed=seed, want_rec=False, want_mix=True,
                                noise_f0=cat(prep["noise_f0"], idxs) if bool(np.any(par["f0_jitter"] != 0)) else None,
                                noise_vol=(cat(nv[0], idxs), cat(nv[1], idxs)) if wants_vol else None,
                                subharm=S.SUBHARM if (prep["subharm"] and wants_sub) else None,
                                mix_only=prep["post"] is None and not keep_stems)
            sub_off = np.concatenate([[0], np.cumsum(lens)])
            for k in out:
                for j, i in enumerate(idxs):
                    out[k][int(so[i]):int(so[i + 1])] = o[k][int(sub_off[j]):int(sub_off[j + 1])]
        return out

    # -- sample-domain post chain (SillySampler.py:1037-1182) ------------------------------------------------------
    def _subset(self, prep, idxs):
        """The assembled inputs of the notes ``idxs`` as their own ragged batch (views when that is every note)."""
        ctx = self.ctx
        n = len(prep["lens"])
        if list(idxs) == list(range(n)):
            return {"env": prep["env"], "f0": prep["f0"], "mask": prep["mask"], "formants": prep["formants"], "phi": prep["phi"]}
        so, eo = prep["sample_off"], prep["env_off"]
        fo = np.concatenate([[0], np.cumsum(ctx.frame_counts(prep["lens"]))])
        cat = lambda t, off: torch.cat([t[int(off[i]):int(off[i + 1])] for i in idxs])
        env = ctx.rows(int(sum(eo[i + 1] - eo[i] for i in idxs)), ctx.n_bins)
        env.copy_(cat(prep["env"], eo))
        phi = None
        if prep["phi"] is not None:
            phi = ctx.rows(int(sum(fo[i + 1] - fo[i] for i in idxs)), ctx.n_bins)
            phi.copy_(cat(prep["phi"], fo))
        return {"env": env, "f0": cat(prep["f0"], so), "mask": cat(prep["mask"], so), "formants": cat(prep["formants"], eo), "phi": phi}

    def _extra_synth(self, prep, idxs, seed, edit=None, **kw):
        """One more gf.synthesize call for the notes ``idxs`` (the su / sj / sa layers re-enter the synth)."""
        sub = self._subset(prep, idxs)
        par = prep["params"][list(idxs)].copy()
        for k in ("f0_jitter", "vol_jitter_harm", "vol_jitter_breath", "subharm_weight"):
            par[k] = 0                                          # the layer calls pass none of the jitter / sg arguments
        if edit is not None:
            edit(sub, par)
        lens, env_lens = [prep["lens"][i] for i in idxs], [prep["env_lens"][i] for i in idxs]
        out = self.ctx.synth_batch(sub["env"], env_lens, sub["f0"], sub["mask"], lens, par, formants=sub["formants"], phi=sub["phi"],
                                   seed=seed, want_rec=False, want_mix=False, **kw)
        out["_sub"] = sub
        return out, np.concatenate([[0], np.cumsum(lens)])

    def _post_chain(self, prep, out, seed):
        ctx, post, jobs = self.ctx, prep["post"].copy(), prep["jobs"]
        n = len(jobs)
        extra = {}
        su = [i for i, (_, r) in enumerate(jobs) if r.subharm_gain > 0.0]
        sj = [i for i, (_, r) in enumerate(jobs) if r.growl_mix > 0.0]
        sa = [i for i, (_, r) in enumerate(jobs) if r.aperiodic_mix > 0.0]
        if su:                                                  # f0 * 0.5 == pitch_shift 0.5 on the fp32 f0          :1038-1049
            def half(sub, par):
                par["pitch_shift"] = 0.5
            extra["su"], off = self._extra_synth(prep, su, seed, edit=half)
            post["su_off"][su] = off[:-1]
        if sj:
            so = prep["sample_off"]
            def growl(sub, par):                                # f0_new * (0.5 * 2^noise), written by the assembly   :1065
                sub["f0"] = torch.cat([prep["f0_growl"][int(so[i]):int(so[i + 1])] for i in sj])
            extra["sj"], off = self._extra_synth(prep, sj, seed, edit=growl)
            post["sj_off"][sj] = off[:-1]
        if sa:                                                  # all-voiced, full-strength noise, transition sigma 1   :1153-1168
            def whisper(sub, par):
                sub["mask"] = torch.ones_like(sub["mask"])
                par["uv_strength"], par["breath_strength"] = 1.0, 1.0
            extra["sa"], off = self._extra_synth(prep, sa, seed ^ 0x5A5A5A5A, edit=whisper, transition_sigma=1.0)
            post["sa_off"][sa] = off[:-1]
        s_host = np.ascontiguousarray(prep["offsets"]["s_off"], dtype=np.int64)
        ptr = lambda t: t.data_ptr() if t is not None else None
        g = lambda k, s: extra[k][s] if k in extra else None
        P = _lib.Post(n_notes=n, total_samples=int(s_host[-1]), sample_off=prep["offsets"]["d_s"].data_ptr(),
                      sample_off_host=s_host.ctypes.data, params=prep["offsets"]["d_par"].data_ptr(), notes=post.ctypes.data,
                      f0=prep["f0"].data_ptr(), mask=prep["mask"].data_ptr(), bend=ptr(prep["bend_out"]),
                      harm=out["harm"].data_ptr(), uv=out["uv"].data_ptr(), bre=out["bre"].data_ptr(),
                      su_harm=ptr(g("su", "harm")), sj_harm=ptr(g("sj", "harm")), sa_uv=ptr(g("sa", "uv")), sa_bre=ptr(g("sa", "bre")),
                      mix=out["mix"].data_ptr())
        ctx._check(ctx.lib.goofer_post_batch(ctx.h, C.byref(P), ctx._stream()))
        out["_keep_post"] = (extra, post, s_host)

    def prepare(self, jobs, phi_seeds=None, note_ids=None, trim_rows: bool = True):
        """Plan every note on the host and make the batch resident in HBM (plans, tables, sources).
        ``note_ids`` key the on-device noise phases (default: the position in ``jobs``), so a note can render to the
        same bits whatever batch or rank it lands in.
        ``trim_rows``: assemble only the envelope rows synthesize can reach.  The reference's L0 loop hands over more
        envelope frames than the note has STFT frames (every cross-faded repeat is appended again, SillySampler.py:657-672)
        and ``gf.synthesize`` cuts the envelope to ``1 + n // hop`` frames (GOOFER.py:1115-1119): the rows behind that are
        never read — a fifth of the rows of a one-second note.  ``False`` keeps them (tests that compare the whole envelope)."""
        ctx = self.ctx
        sr, n_fft = jobs[0][0].sr, jobs[0][0].n_fft
        if any(j[0].sr != sr or j[0].n_fft != n_fft for j in jobs):
            raise ValueError("one batch must share sr / n_fft")
        ctx.plan(sr, n_fft, self.hop)
        B, ld = ctx.n_bins, row_stride(ctx.n_bins)
        n = len(jobs)
        for src, req in jobs:
            T_src = src.knots.shape[1]
            if T_src != 1 + src.ylen // self.hop:             # frames of the analysis STFT (GOOFER.py:355-370)
                raise ValueError(f"bad features: envelope has {T_src} frames, y_len {src.ylen} at hop {self.hop} implies "
                                 f"{1 + src.ylen // self.hop}")
        # notes that share cut points / lengths share one index plan; their formant tracks are processed as one array
        plans = S.plan_notes([(req, src.sr, src.ylen, src.knots.shape[1], src.formants) for src, req in jobs], self.hop)

        # tables shared across notes
        lerp_keys, tilt_keys, fw_keys, es_keys = {}, {}, {}, {}
        lerp_tabs, tilt_tabs, fw_tabs, es_taps, es_off = [], [], [], [], 0
        P = np.zeros(n, dtype=_lib.NOTE_PLAN)
        col = {k: [] for k in ("knot_off", "K", "lerp_plan", "n_src_rows", "reverse", "row_lo", "n_edit", "edit_off", "tilt", "es_mode",
                               "es_amount", "es_taps_off", "es_radius", "fw_plan", "tap_off", "env_off", "n_out_rows", "env_f64", "fst",
                               "src_sample_off", "ylen", "out_sample_off", "n_out", "n_pre", "s_pre", "s_tail", "tail_len",
                               "want_samples", "n_before_vel", "vel_active", "vel_factor", "pre_new", "force_voiced", "bend_off", "n_bend",
                               "pitch_m", "pitch_t", "tick_dt", "fry_hz", "fry_dir", "fry_const_lo", "fry_const_hi", "fry_glide_lo",
                               "fry_glide_hi", "fry_a", "fry_b", "fry_fade", "pd_on", "pd_base")}
        knots_cat, mask_cat, bend_cat, tapi_cat, tapw_cat, fst_cat, F_cat = [], [], [], [], [], [], []
        k_off = e_off = t_off = s_off = o_off = b_off = 0
        hz_ids, live_rows, env_lens, src_at = {}, {}, [], {}
        for i, ((src, req), p) in enumerate(zip(jobs, plans)):
            K = src.knots.shape[0]
            if src.hz_knots is None:                           # dense source: rows are the envelope, no lerp plan
                if K != B:
                    raise ValueError("dense envelope has %d bins, the plan has %d" % (K, B))
                lp = -1
            else:
                lp = hz_ids.get(id(src.hz_knots))              # the same array object again (one voicebank): no hashing of its bytes
                if lp is None:
                    key = (K, src.hz_knots.tobytes())
                    if key not in lerp_keys:
                        lerp_keys[key] = len(lerp_tabs)
                        lerp_tabs.append(_lerp_plan(sr, n_fft, src.hz_knots))
                    lp = hz_ids[id(src.hz_knots)] = lerp_keys[key]
            c = col
            # a voicebank sample rendered by several notes of the batch is uploaded once (same Source object)
            placed = src_at.get(id(src))
            if placed is None:
                placed = src_at[id(src)] = (k_off, s_off)
                knots_cat.append(src.knot_rows())
                mask_cat.append(src.mask[:src.ylen])
                k_off += src.knots.size
                s_off += src.ylen
            c["knot_off"].append(placed[0]); c["K"].append(K); c["lerp_plan"].append(lp); c["n_src_rows"].append(src.knots.shape[1])
            c["reverse"].append(int(req.reverse))
            tilt = -1
            if req.brightness_env != 1.0:
                tk = float(req.brightness_env)
                if tk not in tilt_keys:
                    tilt_keys[tk] = len(tilt_tabs)
                    tilt_tabs.append(_tilt(sr, B, req.brightness_env))
                tilt = tilt_keys[tk]
            c["tilt"].append(tilt)
            es_mode, es_amount, es_toff, es_rad = 0, 0.0, 0, 0
            if req.env_shape != 0.0:
                s_ = abs(req.env_shape)
                es_mode = 1 if req.env_shape < 0.0 else 2
                sigma = (1.0 + 6.0 * s_) if es_mode == 1 else (0.8 + 4.0 * s_)
                ek = (es_mode, sigma)
                if ek not in es_keys:
                    taps = S.gauss_taps(sigma)
                    es_keys[ek] = (es_off, (taps.size - 1) // 2)
                    es_taps.append(taps)
                    es_off += taps.size
                es_amount = 5 * s_
                es_toff, es_rad = es_keys[ek]
            c["es_mode"].append(es_mode); c["es_amount"].append(es_amount); c["es_taps_off"].append(es_toff); c["es_radius"].append(es_rad)
            fw = -1
            if req.formant_width != 0.0:
                fk = float(req.formant_width)
                if fk not in fw_keys:
                    fw_keys[fk] = len(fw_tabs)
                    fw_tabs.append(_fw_plan(B, req.formant_width))
                fw = fw_keys[fk]
            c["fw_plan"].append(fw)
            T_env = p.tap_idx.shape[0]
            row_lo, row_hi = p.row_lo, p.row_hi
            if trim_rows and T_env > 1 + p.n_out // self.hop:
                T_env = 1 + p.n_out // self.hop
                lim = live_rows.get(id(p.tap_idx))             # notes of one geometry share the tap arrays
                if lim is None or lim[0] != T_env:
                    used = p.tap_idx[:T_env][p.tap_w[:T_env] != 0.0]
                    lim = live_rows[id(p.tap_idx)] = (T_env, int(used.min()) if used.size else 0, int(used.max()) + 1 if used.size else 0)
                row_lo, row_hi = lim[1], lim[2]
            c["row_lo"].append(row_lo); c["n_edit"].append(row_hi - row_lo); c["edit_off"].append(e_off)
            c["tap_off"].append(t_off); c["env_off"].append(t_off); c["n_out_rows"].append(T_env); c["env_f64"].append(int(p.env_f64))
            c["fst"].append(req.formant_strength)
            c["src_sample_off"].append(placed[1]); c["ylen"].append(src.ylen); c["out_sample_off"].append(o_off)
            c["n_out"].append(p.n_out); c["n_pre"].append(p.n_pre); c["s_pre"].append(p.extra["s_pre"]); c["s_tail"].append(p.extra["s_tail"])
            c["tail_len"].append(p.tail_len); c["want_samples"].append(p.want_samples); c["n_before_vel"].append(p.n_before_vel)
            c["vel_active"].append(int(p.vel_active)); c["vel_factor"].append(p.vel_factor)
            c["pre_new"].append(max(1, int(round(p.n_pre * p.vel_factor))) if p.vel_active else p.n_pre)
            c["force_voiced"].append(int(req.force_voiced))
            c["bend_off"].append(b_off); c["n_bend"].append(len(req.bend))
            c["pitch_m"].append(float(req.pitch_m))
            tc = req.flags.get("t", 0)
            c["pitch_t"].append((tc / 100.0) if tc else 0.0)
            c["tick_dt"].append(60.0 / (req.tempo * 96.0))
            fx = p.extra
            c["fry_hz"].append(req.fry_hz); c["fry_dir"].append(fx["fry_dir"])
            c["fry_const_lo"].append(fx["fry_const"][0]); c["fry_const_hi"].append(fx["fry_const"][1])
            c["fry_glide_lo"].append(fx["fry_glide"][0]); c["fry_glide_hi"].append(fx["fry_glide"][1])
            c["fry_a"].append(fx["fry_mask"][0]); c["fry_b"].append(fx["fry_mask"][1]); c["fry_fade"].append(fx["fry_fade"])
            c["pd_on"].append(int(req.pitch_dyn != 0.0)); c["pd_base"].append(req.pitch_m + ((req.flags.get("t", 0) or 0) / 100.0))
            semis = req.bend.astype(np.float64) / 100.0 + req.pitch_m      # SillySampler.py:838-846
            if tc:
                semis = semis + (tc / 100.0)
            bend_cat.append(semis)
            tapi_cat.append(p.tap_idx[:T_env])
            tapw_cat.append(p.tap_w[:T_env])
            fst_cat.append(p.fst_tracks[:T_env])
            F_cat.append(p.formants[:T_env])
            env_lens.append(T_env)
            e_off += row_hi - row_lo
            t_off += T_env
            o_off += p.n_out
            b_off += len(req.bend)
        for name, vals in col.items():                         # one column assignment per field instead of 45 scalar stores per note
            P[name] = vals
        if any(pl.n_out <= 0 for pl in plans):
            raise ValueError("a note assembles to zero samples")

        def cat_tab(tabs, k, dtype):
            return ctx.tensor(np.concatenate([t[k] for t in tabs]).astype(dtype)) if tabs else None

        d = dict(
            notes=ctx.tensor(P.view(np.uint8)),
            knots=ctx.tensor(np.concatenate(knots_cat).view(np.uint16)),
            lerp_idx=cat_tab(lerp_tabs, 0, np.int32), lerp_w0=cat_tab(lerp_tabs, 1, np.float32), lerp_w1=cat_tab(lerp_tabs, 2, np.float32),
            tilts=ctx.tensor(np.concatenate(tilt_tabs)) if tilt_tabs else None,
            es_taps=ctx.tensor(np.concatenate(es_taps)) if es_taps else None,
            fw_lo=cat_tab(fw_tabs, 0, np.int32), fw_hi=cat_tab(fw_tabs, 1, np.int32), fw_frac=cat_tab(fw_tabs, 2, np.float64),
            tap_idx=ctx.tensor(np.concatenate(tapi_cat).astype(np.int32, copy=False)), tap_w=ctx.tensor(np.concatenate(tapw_cat)),
            fst_tracks=ctx.tensor(np.concatenate(fst_cat).astype(np.float32, copy=False)),
            mask_src=ctx.tensor(np.concatenate(mask_cat).astype(np.float32, copy=False)),
            bend=ctx.tensor(np.concatenate(bend_cat).astype(np.float64, copy=False)),
        )
        env = ctx.rows(t_off, B)
        f0 = torch.empty(o_off, dtype=torch.float32, device=ctx.device)
        mask = torch.empty(o_off, dtype=torch.float32, device=ctx.device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        any_pd = any(r.pitch_dyn != 0.0 for _, r in jobs)
        any_fry = any(pl.extra["fry_mask"][1] > pl.extra["fry_mask"][0] for pl in plans)
        bend_out = torch.zeros(o_off, dtype=torch.float32, device=ctx.device) if any_pd else None
        a = _lib.Assembly(n_notes=n, n_bins=B, ld=ld, sr=sr, max_K=int(max(s.knots.shape[0] for s, _ in jobs)),
                          total_edit_rows=e_off, total_out_rows=t_off, total_samples=o_off,
                          notes=ptr(d["notes"]), knots=ptr(d["knots"]), lerp_idx=ptr(d["lerp_idx"]), lerp_w0=ptr(d["lerp_w0"]),
                          lerp_w1=ptr(d["lerp_w1"]), tilts=ptr(d["tilts"]), es_taps=ptr(d["es_taps"]), fw_lo=ptr(d["fw_lo"]),
                          fw_hi=ptr(d["fw_hi"]), fw_frac=ptr(d["fw_frac"]), tap_idx=ptr(d["tap_idx"]), tap_w=ptr(d["tap_w"]),
                          fst_tracks=ptr(d["fst_tracks"]), mask_src=ptr(d["mask_src"]), bend=ptr(d["bend"]), edit_rows=None,
                          env_out=env.data_ptr(), f0_out=f0.data_ptr(), mask_out=mask.data_ptr(), bend_out=ptr(bend_out),
                          any_fry=int(any_fry))
        # per-note synthesize parameters
        par = default_params(n)
        reqs = [r for _, r in jobs]
        par["formant_shift"] = [r.formant_shift for r in reqs]
        par["f_shift"] = [r.f_shift for r in reqs]
        par["normalize"] = [r.normalize for r in reqs]
        par["mix_harm"] = [r.harmonic_mix for r in reqs]
        par["mix_breath"] = [r.breathiness_mix for r in reqs]
        par["mix_unvoiced"] = [r.unvoiced_mix for r in reqs]
        par["volume"] = [r.volume for r in reqs]
        nids = np.asarray(note_ids if note_ids is not None else range(n), dtype=np.uint64)   # Philox stream of the note: its id, not its batch position
        par["seed"] = np.stack([nids & np.uint64(0xFFFFFFFF), (nids >> np.uint64(32)) & np.uint64(0xFFFFFFFF)], axis=1)
        par["f0_jitter"] = [r.f0_jitter_strength if r.f0_jitter else 0.0 for r in reqs]
        par["vol_jitter_harm"] = [r.volume_jitter_strength if r.volume_jitter else 0.0 for r in reqs]
        par["vol_jitter_breath"] = [r.volume_jitter_strength * 2 if r.volume_jitter else 0.0 for r in reqs]
        par["subharm_weight"] = [r.subharm_weight if r.add_subharm else 0.0 for r in reqs]
        lens = [p.n_out for p in plans]
        # sh / sr draws come from the legacy global np.random stream, note by note, in the reference's order
        # (f0 jitter, harmonic volume, breath volume: GOOFER.py:666, 653)
        noise_f0 = noise_vol = None
        if any(r.f0_jitter for _, r in jobs) or any(r.volume_jitter for _, r in jobs):
            nf, nh, nb = [], [], []
            for (_, req), n_ in zip(jobs, lens):
                nf.append(np.random.randn(n_) if req.f0_jitter else np.zeros(n_))
                nh.append(np.random.randn(n_) if req.volume_jitter else np.zeros(n_))
                nb.append(np.random.randn(n_) if req.volume_jitter else np.zeros(n_))
            if any(r.f0_jitter for _, r in jobs):
                noise_f0 = ctx.tensor(np.concatenate(nf))
            if any(r.volume_jitter for _, r in jobs):
                noise_vol = (ctx.tensor(np.concatenate(nh)), ctx.tensor(np.concatenate(nb)))
        # sample-domain post chain: per-note table (offsets into the extra synth calls are filled in by run())
        post = np.zeros(n, dtype=_lib.POST_NOTE)
        post["su_off"] = post["sj_off"] = post["sa_off"] = -1
        post["su_gain"] = [r.subharm_gain for r in reqs]
        post["sj_mix"] = [r.growl_mix for r in reqs]
        post["sa_mix"] = [r.aperiodic_mix for r in reqs]
        post["sd_strength"] = [r.sd_strength for r in reqs]
        post["tension"] = [r.tension for r in reqs]
        post["pitch_dyn"] = [r.pitch_dyn for r in reqs]
        post["fry_a"] = [pl.extra["fry_mask"][0] for pl in plans]
        post["fry_b"] = [pl.extra["fry_mask"][1] for pl in plans]
        post["fry_fade"] = [pl.extra["fry_fade"] for pl in plans]
        growl = {}
        for i, (req, pl) in enumerate(zip(reqs, plans)):
            if req.growl_mix > 0.0:                            # 'sj': f0 * 0.5 * 2^N(0, mix^2), a fresh generator per call  :1063-1065
                rng = np.random.default_rng(phi_seeds[i]) if phi_seeds is not None else np.random.default_rng()
                growl[i] = 0.5 * (2.0 ** rng.normal(loc=0.0, scale=req.growl_mix ** 2, size=pl.n_out))
        f0_growl = None
        if growl:
            # the layer's f0 is the fp64 pitch curve times the factor, rounded to fp32 once (SillySampler.py:1065): the
            # assembly kernel writes it next to f0 while it still holds the fp64 value
            so = np.concatenate([[0], np.cumsum(lens)])
            mul = np.ones(int(so[-1]), dtype=np.float64)
            for i, gv in growl.items():
                mul[int(so[i]):int(so[i + 1])] = gv
            d["f0_mul"] = ctx.tensor(mul)
            f0_growl = torch.zeros(int(so[-1]), dtype=torch.float32, device=ctx.device)
            a.f0_mul, a.f0_mul_out = d["f0_mul"].data_ptr(), f0_growl.data_ptr()
        has_post = any(r.subharm_gain > 0 or r.growl_mix > 0 or r.aperiodic_mix > 0 or r.sd_strength > 0 or r.tension != 0
                       or r.pitch_dyn != 0 for _, r in jobs) or any_fry
        phi = None
        if phi_seeds is not None:
            mats = []
            for p, sd in zip(plans, phi_seeds):
                T = 1 + p.n_out // self.hop
                mats.append(np.random.default_rng(sd).uniform(0.0, 2.0 * np.pi, size=(B, T)).astype(np.float32).T)
            phi = ctx.rows_from(np.concatenate(mats))
        offsets = ctx.device_offsets(env_lens, lens, par)
        frames = int(sum(ctx.frame_counts(lens)))
        ctx.reserve(frames, int(sum(lens)), n)
        torch.cuda.synchronize(ctx.device)
        return {"assembly": a, "keep": d, "env": env, "f0": f0, "mask": mask, "params": par, "lens": lens, "env_lens": env_lens,
                "noise_f0": noise_f0, "noise_vol": noise_vol, "subharm": any(r.add_subharm for _, r in jobs),
                "post": post if has_post else None, "growl": growl, "f0_growl": f0_growl, "bend_out": bend_out, "jobs": jobs,
                "formants": ctx.tensor(np.concatenate(F_cat)), "phi": phi, "plans": plans, "offsets": offsets,
                "sample_off": np.concatenate([[0], np.cumsum(lens)]), "env_off": np.concatenate([[0], np.cumsum(env_lens)]),
                "frames": frames, "samples": int(sum(lens)), "edit_rows": e_off}


# ---------------------------------------------------------------------------------------------
# the reference's one-note call surface
# ---------------------------------------------------------------------------------------------
class GooferResampler:
    """``GooferResampler(in_file, out_file, pitch, velocity, flags, offset, length, consonant, cutoff, volume,
    modulation, tempo, pitch_string)`` — construction renders and writes ``out_file`` (SillySampler.py:285-413).

    Uses ``<in_stem>_features.goofy`` next to the input wav; when it is missing the wav is analysed and the cache written
    first, like the reference does (``goofer_amd.trackers``: needs a tracker for the Praat half — parity unpinned).  The first
    render then uses the features as cached (fp16 knots), where the reference's first render still holds the unquantised
    envelope.  wav output uses the stdlib ``wave`` module (PCM16, what soundfile's default WAV subtype writes)."""

    def __init__(self, in_file, out_file, pitch, velocity, flags="", offset=0, length=1000, consonant=0, cutoff=0,
                 volume=100, modulation=0, tempo="!120", pitch_string="AA", renderer: Renderer | None = None, seed=None, tracker=None):
        from pathlib import Path
        from . import core
        self.in_file, self.out_file = Path(in_file), Path(out_file)
        self.request = S.decode_request(pitch, velocity, flags, offset, length, consonant, cutoff, volume, modulation, tempo,
                                        pitch_string)
        # cached features, or — the first render of a sample — analysed now and cached (SillySampler.py:415-432): the envelope
        # half on the GPU, the f0 / formant tracks from the tracker (goofer_amd.trackers; raises when none is available)
        from . import trackers
        self.renderer = renderer or Renderer()
        feat = trackers.ensure_features(self.in_file, hop_length=self.renderer.hop, tracker=tracker, ctx=self.renderer.ctx)
        env, f0, mask, forms, sr, ylen = core.load_features(feat)
        self.source = Source.from_pack(env, f0, mask, forms, sr, ylen)
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
        self.out = self.renderer.render([(self.source, self.request)], seed=seed)[0]
        write_wav(self.out_file, self.out, sr)


def write_wav(path, x, sr):
    import wave
    pcm = np.clip(np.asarray(x, dtype=np.float64), -1.0, 1.0 - 1.0 / 32768)
    pcm = np.round(pcm * 32768.0).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm.tobytes())
