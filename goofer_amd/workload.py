"""Synthetic synthesize-level workloads for bench.py / smoke() / scaling tests.

A workload is a ragged batch of *assembled* notes — what ``SillySampler.resample`` hands to
``gf.synthesize`` (SillySampler.py:1006-1035): envelope rows, per-sample f0 and voicing mask,
per-frame formant tracks, per-note flag scalars.  Everything is seeded by the global note id, so a
note is the same bytes whichever rank or batch it lands in.

The envelope comes from the ``.goofy`` knot codec (decoded on the device by k_knot_decode), the
sustain is extended by tiling source frames, the pitch curve is the request's MIDI note + pitch-bend
string sampled per sample exactly as SillySampler.py:835-855 does.
"""
from __future__ import annotations

import re

import numpy as np
import torch

from . import _lib
from . import synthetic as syn
from .device import Context, default_params

_FLAG = re.compile(r"([A-Za-z]{1,4})([+-]?\d+)?")
_B64 = {c: i for i, c in enumerate("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/")}
_NOTE = re.compile(r"([A-G]#?)(-?\d+)")
_SEMI = {"C": 0, "C#": 1, "D": 2, "D#": 3, "E": 4, "F": 5, "F#": 6, "G": 7, "G#": 8, "A": 9, "A#": 10, "B": 11}


def parse_flags(s: str) -> dict:
    return {k: (int(v) if v else None) for k, v in _FLAG.findall(s.replace("/", ""))}


def decode_bend(s: str) -> np.ndarray:
    parts, out = s.split("#"), []
    for i in range(0, len(parts), 2):
        seg = parts[i]
        for j in range(0, len(seg), 2):
            v = (_B64[seg[j]] << 6) | _B64[seg[j + 1]]
            out.append(v - 4096 if v & 0x800 else v)
        if i + 1 < len(parts):
            out += [out[-1]] * int(parts[i + 1])
    return np.array(out if out else [0.0], dtype=np.float32)


def note_midi(name: str) -> int:
    m = _NOTE.match(name)
    return (int(m.group(2)) + 1) * 12 + _SEMI[m.group(1)]


def params_from_flags(flags: dict, volume: float) -> np.ndarray:
    """Flag -> synthesize kwargs scaling (SillySampler.py:313-343)."""
    p = default_params(1)
    g = lambda k, d=0: (flags.get(k, d) if flags.get(k, d) is not None else d)
    p["formant_shift"] = 1.0 + g("g") / 200.0
    p["f_shift"] = [1.0 + g(k) / 100.0 for k in ("fa", "fb", "fc", "fd")]
    p["normalize"] = (np.clip(flags["P"], 0, 100) / 100.0) if "P" in flags else 1.0
    p["mix_harm"] = np.clip(g("V", 100), 0, 100) / 100.0
    p["mix_breath"] = (g("B") + 100) / 100.0
    p["mix_unvoiced"] = (g("U") + 100) / 100.0
    p["volume"] = volume
    return p


def assembled_note(config: int, i: int) -> dict:
    """Host-side description of note ``i`` of a BASELINE config (numpy, no device work)."""
    src, req, phi_seed = syn.config_note(config, i)
    geo = syn.config_geometry(config)
    sr, hop = geo["sr"], geo["hop"]
    flags = parse_flags(req["flags"])
    off, cons, length = float(req["offset"]) / 1e3, float(req["consonant"]) / 1e3, float(req["length"]) / 1e3
    cut = float(req["cutoff"]) / 1e3
    n_src = src["y_len"]
    s0 = int(off * sr)
    s1 = s0 + int(cons * sr)
    s2 = int((n_src / sr - cut) * sr)
    want = int(length * sr)
    idx_tail = s1 + (np.arange(want) % max(1, s2 - s1))
    sample_idx = np.concatenate([np.arange(s0, s1), idx_tail])
    n = sample_idx.size
    f_0, f_1, f_2 = s0 // hop, s1 // hop, s2 // hop
    want_f = int(np.ceil(length * sr / hop))
    frame_idx = np.concatenate([np.arange(f_0, f_1), f_1 + (np.arange(want_f) % max(1, f_2 - f_1))])
    mask = src["mask"][sample_idx]
    semis = decode_bend(req["pitch_string"]).astype(np.float64) / 100.0 + note_midi(req["pitch"])
    if flags.get("t"):
        semis = semis + flags["t"] / 100.0
    t_p = np.arange(len(semis)) * (60.0 / (float(req["tempo"].lstrip("!")) * 96.0))
    midi = np.interp(np.clip(np.arange(n) / sr, t_p[0], t_p[-1]), t_p, semis)
    f0 = (mask * (440.0 * 2 ** ((midi - 69) / 12))).astype(np.float32)
    F = np.stack([np.asarray(src["formants"][k], dtype=np.float64)[frame_idx] for k in (1, 2, 3, 4)], axis=1)
    par = params_from_flags(flags, float(req["volume"]) / 100.0)
    par["seed"] = [phi_seed & 0xFFFFFFFF, 0]
    return {"knots": src["env_pack"]["knot_vals_log"], "hz_knots": src["env_pack"]["hz_knots"], "frame_idx": frame_idx,
            "f0": f0, "mask": mask.astype(np.float32), "formants": F, "params": par, "n": n, "rows": frame_idx.size,
            "phi_seed": phi_seed, "geo": geo}


class SynthWorkload:
    """A batch resident in HBM, ready for ``Context.synth_batch``."""

    def __init__(self, ctx: Context, config: int, note_ids):
        notes = [assembled_note(config, int(i)) for i in note_ids]
        geo = notes[0]["geo"]
        ctx.plan(geo["sr"], geo["n_fft"], geo["hop"])
        self.ctx, self.config, self.geo, self.notes = ctx, config, geo, notes
        self.lens = [nt["n"] for nt in notes]
        self.env_lens = [nt["rows"] for nt in notes]
        # knot decode of every source frame on the device, then gather the assembled rows
        knots = np.concatenate([nt["knots"].T for nt in notes]).astype(np.float16)         # [sum T_src, K]
        base = np.concatenate([[0], np.cumsum([nt["knots"].shape[1] for nt in notes])])
        src_env = ctx.knot_decode(ctx.tensor(knots.view(np.uint16)).view(torch.float16), notes[0]["hz_knots"])
        gather = np.concatenate([base[j] + nt["frame_idx"] for j, nt in enumerate(notes)])
        self.env = ctx.rows(int(gather.size), ctx.n_bins)
        self.env.copy_(src_env[ctx.tensor(gather)])
        self.f0 = ctx.tensor(np.concatenate([nt["f0"] for nt in notes]))
        self.mask = ctx.tensor(np.concatenate([nt["mask"] for nt in notes]))
        self.formants = ctx.tensor(np.concatenate([nt["formants"] for nt in notes]))
        p = np.zeros(len(notes), dtype=_lib.NOTE_PARAMS)
        for j, nt in enumerate(notes):
            p[j] = nt["params"][0]
        self.params = p
        self.frames = int(sum(ctx.frame_counts(self.lens)))
        self.samples = int(sum(self.lens))
        ctx.reserve(self.frames, self.samples, len(notes))
        torch.cuda.synchronize()

    def step(self, phi=None, want_rec=False):
        return self.ctx.synth_batch(self.env, self.env_lens, self.f0, self.mask, self.lens, self.params,
                                    formants=self.formants, phi=phi, seed=0, want_rec=want_rec, want_mix=True)

    def host_note(self, j: int) -> dict:
        """Inputs of note j as the numpy arrays gf.synthesize takes ([bins, frames] layout)."""
        e0 = int(np.sum(self.env_lens[:j]))
        nt = self.notes[j]
        env = self.env[e0:e0 + nt["rows"]].cpu().numpy().T.copy()
        p = nt["params"][0]
        kw = dict(formant_shift=float(p["formant_shift"]), F1_shift=float(p["f_shift"][0]), F2_shift=float(p["f_shift"][1]),
                  F3_shift=float(p["f_shift"][2]), F4_shift=float(p["f_shift"][3]), normalize=float(p["normalize"]))
        return {"env": env, "f0": nt["f0"].astype(np.float64), "mask": nt["mask"], "n": nt["n"],
                "formants": {k + 1: nt["formants"][:, k] for k in range(4)}, "kw": kw, "params": p,
                "phi_seed": nt["phi_seed"]}


class SamplerWorkload:
    """BASELINE config N as the reference would see it: ``.goofy`` features + the 13-argument request per note.
    Planning happens once on the host; plans, tables and sources stay resident in HBM; ``step()`` is the
    device work of one render of the whole batch (assemble + synthesize + mix)."""

    def __init__(self, ctx: Context, config: int, note_ids, unvoiced_share: float = 0.0):
        from .render import Renderer, Source
        from . import sampler as S
        self.geo = syn.config_geometry(config)
        self.config = config
        self.ctx = ctx
        self.renderer = Renderer(ctx, hop=self.geo["hop"])
        self.raw = []
        jobs = []
        for i in note_ids:
            src, req, phi_seed = syn.config_note(config, int(i))
            if unvoiced_share > 0.0:
                src = syn.with_unvoiced_gaps(src, unvoiced_share, 1000 + int(i))
            self.raw.append((src, req, phi_seed))
            jobs.append((Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]),
                         S.decode_request(*syn.request_args(req))))
        self.prep = self.renderer.prepare(jobs, note_ids=[int(i) for i in note_ids])
        self.frames, self.samples = self.prep["frames"], self.prep["samples"]
        self.notes = jobs

    def step(self):
        return self.renderer.run(self.prep, seed=0)
