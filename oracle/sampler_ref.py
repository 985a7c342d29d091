"""CPU oracle for the UTAU resampler front half — TEST INFRASTRUCTURE, not product code.

Restates ``/root/reference/SillySampler.py``: flag / pitch-string decode (:50-93, :286-411),
segment slicing (:449-500), envelope edits br/es/fw (:502-574), loop modes (:625-763), velocity
prefix stretch (:765-788), formant-strength bells (:791-833), pitch curve (:835-855), pd / fry
(:857-997), the up-to-four ``synthesize`` calls and the sample-domain post chain (:1003-1182).
Pinned by ``tests/golden/sampler_*.npz``, ``index_plans.npz``, ``flags_pitch.npz``,
``post_chain.npz`` (all produced by the reference itself).

Randomness: each ``default_rng()`` call of the reference becomes ``default_rng(seed)`` (what the
golden harness pins); legacy ``np.random`` draws (sh / sr flags) come from the global state, so a
test seeds it with ``np.random.seed`` exactly as the harness did.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field

import numpy as np

from . import goofer_ref as G

F32 = np.float32
N_FFT = 1024          # SillySampler.py:14-15 — not user-settable from the CLI
HOP = N_FFT // 4

_NOTE = re.compile(r"([A-G]#?)(-?\d+)")
_FLAG = re.compile(r"([A-Za-z]{1,4})([+-]?\d+)?")
_SEMITONE = {"C": 0, "C#": 1, "D": 2, "D#": 3, "E": 4, "F": 5, "F#": 6, "G": 7, "G#": 8, "A": 9, "A#": 10, "B": 11}


# ---------------------------------------------------------------------------------------------
# string decode — integer path, bit-exact                                  SillySampler.py:50-93
# ---------------------------------------------------------------------------------------------
def parse_flags(s: str) -> dict:
    """'/' stripped; 1-4 letters + optional signed int; a bare letter maps to None."""
    return {k: (int(v) if v else None) for k, v in _FLAG.findall(s.replace("/", ""))}


def _sextet(c: str) -> int:
    o = ord(c)
    if o >= 97:
        return o - 71
    if o >= 65:
        return o - 65
    if o >= 48:
        return o + 4
    if o == 43:
        return 62
    if o == 47:
        return 63
    raise ValueError(f"Bad b64 '{c}'")


def _int12_run(s: str) -> list:
    out = []
    for i in range(0, len(s), 2):
        pair = s[i:i + 2]
        v = (_sextet(pair[0]) << 6) | _sextet(pair[1])   # IndexError on an odd tail, like the reference
        out.append(v - 4096 if v & 0x800 else v)
    return out


def pitch_string_to_cents(x: str) -> np.ndarray:
    """UTAU pitch bend: base64 12-bit pairs, '#n#' repeats the last value n more times; an empty
    result becomes [0.0]; fp32."""
    parts = x.split("#")
    vals = []
    for i in range(0, len(parts), 2):
        vals += _int12_run(parts[i])
        if i + 1 < len(parts):
            vals += [vals[-1]] * int(parts[i + 1])
    a = np.array(vals, dtype=F32)
    return a if a.size else np.array([0.0], dtype=F32)


def note_to_midi(name: str) -> int:
    m = _NOTE.match(name)
    if not m:
        raise ValueError(f"Bad note '{name}'")
    return (int(m.group(2)) + 1) * 12 + _SEMITONE[m.group(1)]


def midi_to_hz(m):
    return 440.0 * 2 ** ((m - 69) / 12)


def split_arguments(s: str) -> list:
    """HTTP body -> 13 args: last 11 space-separated tokens + the first two ``\\S+\\.wav`` matches
    of the rest (SillySampler.py:1187-1194)."""
    toks = s.split(" ")
    wavs = re.findall(r"([^\s]+\.wav)", " ".join(toks[:-11]))
    if len(wavs) < 2:
        raise ValueError("Missing .wav file paths in POST string")
    return wavs[:2] + toks[-11:]


def _ci(flags: dict, name: str, default=0):
    """First flag whose key matches case-insensitively (dict order), else default."""
    return next((v for k, v in flags.items() if k.lower() == name), default)


@dataclass
class NoteParams:
    """Scalars of one note after flag scaling (GooferResampler.__init__, SillySampler.py:286-411)."""
    pitch_m: int
    velocity: float
    flags: dict
    offset: float
    length: float
    consonant: float
    cutoff: float
    volume: float
    modulation: float
    tempo: float
    bend: np.ndarray
    use_editor: bool = False
    formant_shift: float = 1.0
    brightness_env: float = 1.0
    F_shift: tuple = (1.0, 1.0, 1.0, 1.0)
    f0_jitter: bool = False
    f0_jitter_strength: float = 0.0
    volume_jitter: bool = False
    volume_jitter_strength: float = 0.0
    sd_strength: float = 0.0
    breathiness_mix: float = 1.0
    unvoiced_mix: float = 1.0
    harmonic_mix: float = 1.0
    loop_mode: str = "concat"
    tension: float = 0.0
    subharm_weight: float = 0.0
    add_subharm: bool = False
    reverse: bool = False
    growl_mix: float = 0.0
    aperiodic_mix: float = 0.0
    subharm_gain: float = 0.0
    normalize: float = 1.0
    env_shape: float = 0.0
    force_voiced: bool = False
    pitch_dyn: float = 0.0
    formant_width: float = 0.0
    formant_strength: tuple = (0.0, 0.0, 0.0, 0.0)
    extra: dict = field(default_factory=dict)


def decode_request(pitch, velocity, flags="", offset=0, length=1000, consonant=0, cutoff=0,
                   volume=100, modulation=0, tempo="!120", pitch_string="AA") -> NoteParams:
    fl = parse_flags(flags)
    g = fl.get
    p = NoteParams(
        pitch_m=note_to_midi(pitch), velocity=float(velocity), flags=fl,
        offset=float(offset) / 1000.0, length=float(length) / 1000.0, consonant=float(consonant) / 1000.0,
        cutoff=float(cutoff) / 1000.0, volume=float(volume) / 100.0, modulation=float(modulation) / 100.0,
        tempo=float(tempo.lstrip("!")), bend=pitch_string_to_cents(pitch_string))
    p.use_editor = _ci(fl, "se") == 1
    p.formant_shift = 1.0 + (g("g", 0) / 200.0)
    p.brightness_env = (g("br", 0) + 100) / 100.0
    p.F_shift = tuple(1.0 + (g(k, 0) / 100.0) for k in ("fa", "fb", "fc", "fd"))
    sh, sr_ = g("sh", None), g("sr", None)
    p.f0_jitter = sh is not None and sh > 0
    p.f0_jitter_strength = (sh or 0) / 50.0
    p.volume_jitter = sr_ is not None and sr_ > 0
    p.volume_jitter_strength = (sr_ or 0) / 50.0
    p.sd_strength = float(g("sd", None) or 0)
    p.breathiness_mix = (g("B", 0) + 100) / 100.0
    p.unvoiced_mix = (g("U", 0) + 100) / 100.0
    p.harmonic_mix = np.clip(g("V", 100), 0, 100) / 100.0
    lkey = next((k for k in fl if k.lower() == "l"), None)
    p.loop_mode = {1: "avg", 2: "stretch"}.get(fl[lkey], "concat") if lkey else "concat"
    p.tension = g("st", 0) / 100.0
    sg = g("sg", 0)
    p.subharm_weight = (sg / 100.0) * 1.5
    p.add_subharm = sg > 0
    p.reverse = g("R", 0) == 1
    p.growl_mix = np.clip(g("sj", 0) or 0, 0, 100) / 100.0
    p.aperiodic_mix = np.clip(g("sa", 0) or 0, 0, 100) / 100.0
    p.subharm_gain = np.clip(g("su", 0) or 0, 0, 100) / 100.0
    p.normalize = (np.clip(fl["P"], 0, 100) / 100.0) if "P" in fl else 1.0
    p.env_shape = float(np.clip(_ci(fl, "es") or 0, -100, 100)) / 100.0
    p.force_voiced = g("FV", 0) == 1
    p.pitch_dyn = float(int(np.clip(_ci(fl, "pd") or 0, -100, 100))) / 100.0
    p.formant_width = ((g("fw", 0) or 0) / 100.0) * 0.1
    glob = float(np.clip(_ci(fl, "fst") or 0, -100, 100)) / 100.0
    p.formant_strength = tuple(float(np.clip(glob + ((_ci(fl, "fst" + c) or 0) / 100.0), -1.0, 1.0)) for c in "abcd")
    p.extra["formant_strength_global"] = glob
    return p


# ---------------------------------------------------------------------------------------------
# helpers                                                                  SillySampler.py:95-283
# ---------------------------------------------------------------------------------------------
def dynamic_filter(signal, f0, sr, cutoff_factor, order=4, btype="lowpass") -> np.ndarray:
    """Cascade of ``order`` one-pole LP/HP sections whose per-sample coefficient follows
    cutoff_factor * (5-tap box-smoothed f0), clamped to [60|20 Hz, 0.45 sr]; all fp32
    (SillySampler.py:95-174)."""
    x = np.asarray(signal, dtype=F32)
    n = len(x)
    if n == 0:
        return x
    f0 = np.asarray(f0, dtype=F32)
    if f0.size != n:
        pos = np.linspace(0, n - 1, num=f0.size, dtype=np.float64)
        f0 = G.LinInterp(pos, f0.astype(np.float64))(np.arange(n, dtype=np.float64)).astype(F32)
    if np.any(f0 > 0):
        f0s = np.convolve(np.pad(f0, (2, 2), mode="edge"), np.ones(5, dtype=F32) / 5, mode="valid")
    else:
        f0s = f0
    f0s = np.asarray(f0s, dtype=F32)
    cf = F32(cutoff_factor)
    fc = np.where(f0s > 0.0, f0s * cf, cf).astype(F32)
    fc = np.maximum(fc, F32(60.0 if btype == "lowpass" else 20.0))
    fc = np.minimum(fc, F32(0.45 * sr))
    w = (2.0 * np.pi) * fc.astype(np.float64)
    alpha = ((w / (w + sr)) if btype == "lowpass" else (sr / (w + sr))).astype(F32)
    y = x.copy()
    lib = G._native()
    if lib is not None:
        lib.onepole_cascade(G._p(y), G._p(np.ascontiguousarray(alpha)), n, int(max(1, int(order))), int(btype != "lowpass"))
        return y
    for _ in range(max(1, int(order))):
        yp = F32(0.0)
        prev = y[0]
        for i in range(n):
            a, xp = alpha[i], y[i]
            if btype == "lowpass":
                yp = yp + a * (xp - yp)
            else:
                yp = a * (yp + xp - prev)
                prev = xp
            y[i] = yp
    return y


def _prefix_positions(n: int, pre_len: int, factor: float):
    pre_new = max(1, int(round(pre_len * factor)))
    idx = np.arange(pre_new + (n - pre_len), dtype=np.float64)
    return np.where(idx < pre_new, idx / factor, (idx - pre_new) + pre_len)


def stretch_prefix_1d(x, pre_len, factor):
    """Time-scale the first pre_len points by factor, keep the rest (SillySampler.py:176-187)."""
    n = len(x)
    if pre_len <= 1 or n <= 1 or abs(factor - 1.0) < 1e-6:
        return x
    return G.LinInterp(np.arange(n, dtype=np.float64), x)(_prefix_positions(n, pre_len, factor))


def stretch_prefix_2d(M, pre_len, factor):
    n = M.shape[1]
    if pre_len <= 1 or n <= 1 or abs(factor - 1.0) < 1e-6:
        return M
    pos = _prefix_positions(n, pre_len, factor)
    xo = np.arange(n, dtype=np.float64)
    return np.stack([G.LinInterp(xo, row)(pos) for row in M], axis=0)


def sanitize_formant(track, T, sr, min_hz=120.0, max_hz=None, sigma_frames=3):
    """Fit to T frames, replace non-finite / out-of-range values by interpolation over the good
    ones (300 Hz if none), then Gaussian-smooth along frames (SillySampler.py:264-283)."""
    max_hz = max_hz or (sr * 0.48)
    x = np.asarray(track, dtype=F32)
    x = np.pad(x, (0, T - len(x)), mode="edge") if len(x) < T else x[:T]
    # NOTE: no copy.  When ``track`` is already fp32 and not padded, ``x`` aliases it and the repair
    # below edits the caller's array in place — the reference relies on this: the tracks handed to
    # synthesize() carry the repaired (but unsmoothed) values.
    bad = (~np.isfinite(x)) | (x < min_hz) | (x > max_hz)
    if bad.any():
        good = np.where(~bad)[0]
        if good.size:
            x[bad] = G.LinInterp(good.astype(F32), x[~bad])(np.where(bad)[0].astype(F32))
        else:
            x = np.full_like(x, 300.0)
    if sigma_frames > 0:
        x = G.gauss1d(x, sigma_frames)
    return x.astype(F32)


def _canon_formants(d: dict, T: int) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(k, (int, np.integer)):
            name = f"F{int(k)}"
        elif isinstance(k, str) and k.upper().startswith("F"):
            name = k.upper()
        else:
            try:
                name = f"F{int(k)}"
            except Exception:
                continue
        a = np.asarray(v, dtype=F32)
        a = np.pad(a, (0, T - len(a)), mode="edge") if len(a) < T else a[:T]
        out[name] = a
    return out


# ---------------------------------------------------------------------------------------------
# note assembly                                                            SillySampler.py:449-1001
# ---------------------------------------------------------------------------------------------
def segment_indices(p: NoteParams, sr, ylen, hop=HOP) -> dict:
    """Sample / frame cut points from offset, consonant, cutoff (int() truncation, // hop);
    negative cutoff is relative to offset; R1 mirrors the window (SillySampler.py:453-487)."""
    total = ylen / sr
    a0 = p.offset
    b0 = (p.offset - p.cutoff) if p.cutoff < 0 else (total - p.cutoff)
    if p.reverse:
        L = b0 - a0
        off = total - b0
        cut = total - (off + L)
    else:
        off, cut = p.offset, p.cutoff
    s0 = int(off * sr)
    s1 = s0 + int(p.consonant * sr)
    s2 = int(((off - cut) if cut < 0 else (total - cut)) * sr)
    return {"start_sample": s0, "consonant_sample": s1, "end_sample": s2,
            "start_frame": s0 // hop, "consonant_frame": s1 // hop, "end_frame": s2 // hop}


def _loop_env_concat(tail, want):
    """L0: repeat the tail with <=8-frame linear cross-fades; reproduces the reference's
    over-production (each pass re-appends a full copy) (SillySampler.py:654-696)."""
    n = tail.shape[1]
    reps, rem = want // n, want % n
    chain = [tail.copy()]
    for _ in range(reps - 1):
        prev = chain[-1]
        k = min(8, n // 2)
        up = np.linspace(0, 1, k)[None, :]
        dn = np.linspace(1, 0, k)[None, :]
        mixed = prev[:, -k:] * dn + tail[:, :k] * up
        chain[-1] = np.concatenate([prev[:, :-k], mixed, tail[:, k:]], axis=1)
        chain.append(tail.copy())
    if rem:
        last = tail[:, :rem]
        prev = chain[-1]
        k = min(8, rem // 2)
        if k > 0:
            up = np.linspace(0, 1, k)[None, :]
            dn = np.linspace(1, 0, k)[None, :]
            mixed = prev[:, -k:] * dn + last[:, :k] * up
            chain[-1] = np.concatenate([prev[:, :-k], mixed, last[:, k:]], axis=1)
        else:
            chain[-1] = np.concatenate([prev, last], axis=1)
    return np.concatenate(chain, axis=1)


def assemble(features, p: NoteParams, hop=HOP) -> dict:
    """Everything ``resample`` does before the first synthesize call.  ``features`` =
    (env [bins,T] or knots dict, f0 [N], mask [N], formants dict, sr, ylen).  The input arrays may
    be modified in place exactly where the reference does (views)."""
    env_spec, f0_src, vmask, forms, sr, ylen = features
    if isinstance(env_spec, dict) and env_spec.get("mode") == "knots":
        env_spec = G.decode_env_from_knots(env_spec)
    if p.reverse:                                                     # render(), :438-444
        env_spec = env_spec[:, ::-1]
        f0_src = f0_src[::-1]
        vmask = vmask[::-1]
        forms = {k: list(forms[k])[::-1] for k in forms}
    seg = segment_indices(p, sr, ylen, hop)
    s0, s1, s2 = seg["start_sample"], seg["consonant_sample"], seg["end_sample"]
    f_0, f_1, f_2 = seg["start_frame"], seg["consonant_frame"], seg["end_frame"]

    env_pre, env_tail = env_spec[:, f_0:f_1], env_spec[:, f_1:f_2]
    f0_pre, f0_tail = f0_src[s0:s1], f0_src[s1:s2]
    mask_pre, mask_tail = vmask[s0:s1], vmask[s1:s2]

    if p.brightness_env != 1.0 and (env_pre.size or env_tail.size):   # br, :502-515
        nb = env_spec.shape[0]
        fr = np.linspace(1e-6, sr * 0.5, nb, dtype=F32)
        nf = np.clip(fr / (sr * 0.5), 0.02, 1.0)
        tilt = nf ** np.clip(p.brightness_env - 1.0, -0.9, 1.0)      # np.float64 exponent -> fp64
        tilt /= (tilt.mean() + 1e-12)
        if env_pre.size:
            env_pre *= tilt[:, None].astype(env_pre.dtype)
        if env_tail.size:
            env_tail *= tilt[:, None].astype(env_tail.dtype)

    if p.env_shape != 0.0 and (env_pre.size or env_tail.size):        # es, :517-551
        s = abs(p.env_shape)

        def rematch(orig, mod):
            m0 = np.mean(orig, axis=0, keepdims=True)
            m1 = np.mean(mod, axis=0, keepdims=True)
            return (mod * (m0 / (m1 + 1e-12))).astype(orig.dtype)

        def shape_block(b):
            if not b.size:
                return b
            if p.env_shape < 0.0:
                return np.maximum(0.0, rematch(b, G.gauss1d(b, 1.0 + 6.0 * s, axis=0)))
            out = b + (5 * s) * (b - G.gauss1d(b, 0.8 + 4.0 * s, axis=0))
            return rematch(b, np.maximum(0.0, out))

        env_pre, env_tail = shape_block(env_pre), shape_block(env_tail)

    if p.formant_width != 0.0 and env_spec.size:                      # fw, :553-574
        def widen(e):
            nb = e.shape[0]
            c = nb / 2.0
            w = np.clip((np.arange(nb, dtype=np.float64) - c) * (1.0 + p.formant_width) + c, 0, nb - 1)
            lo = np.floor(w).astype(int)
            hi = np.minimum(lo + 1, nb - 1)
            fr = (w - lo)[:, None]
            out = np.empty_like(e)
            out[:] = (1 - fr) * e[lo, :] + fr * e[hi, :]
            return out
        if env_pre.size:
            env_pre = widen(env_pre)
        if env_tail.size:
            env_tail = widen(env_tail)

    if p.force_voiced:                                                # FV, :619-623
        if mask_pre.size:
            mask_pre[:] = 1.0
        if mask_tail.size:
            mask_tail[:] = 1.0

    want_s = int(p.length * sr)
    n_tail = env_tail.shape[1]
    want_f = int(np.ceil(p.length * sr / hop))
    if n_tail >= want_f:
        tail_env = env_tail[:, :want_f]
    else:
        reps, rem = want_f // n_tail, want_f % n_tail                 # ZeroDivisionError on empty tail
        if p.loop_mode == "stretch":
            tail_env = G.stretch_feature(env_tail, want_f / n_tail)
        elif p.loop_mode == "avg":
            tile = (env_tail + env_tail[:, ::-1]) / 2.0
            tail_env = np.concatenate([tile] * reps + ([tile[:, :rem]] if rem else []), axis=1)
        else:
            tail_env = _loop_env_concat(env_tail, want_f)

    n_ts = len(f0_tail)
    if n_ts >= want_s:
        f0_loop, mask_loop = f0_tail[:want_s], mask_tail[:want_s]
    else:
        reps, rem = want_s // n_ts, want_s % n_ts
        f0_loop = np.concatenate([f0_tail] * reps + ([f0_tail[:rem]] if rem else []))
        mask_loop = np.concatenate([mask_tail] * reps + ([mask_tail[:rem]] if rem else []))

    fm_new = {}
    for k in forms:                                                   # :714-749
        pre = forms[k][f_0:f_1]
        tr = np.asarray(forms[k][f_1:f_2], dtype=F32)
        if tr.size == 0:
            lp = np.zeros(want_f, dtype=F32)
        elif p.loop_mode == "stretch":
            lp = G.stretch_feature(tr, want_f / float(tr.size)).astype(F32)
        else:
            reps, rem = want_f // tr.size, want_f % tr.size
            tile = (tr + tr[::-1]) * 0.5 if p.loop_mode == "avg" else tr
            lp = np.tile(tile, reps)
            if rem > 0:
                lp = np.concatenate([lp, tile[:rem]])
            lp = lp.astype(F32)
        fm_new[k] = np.concatenate([pre, lp])

    env_new = np.concatenate([env_pre, tail_env], axis=1)
    f0_new = np.concatenate([f0_pre, f0_loop])
    mask_new = np.concatenate([mask_pre, mask_loop])
    T_target = env_new.shape[1]
    for k in fm_new:
        f = fm_new[k]
        fm_new[k] = np.pad(f, (0, T_target - len(f)), mode="edge") if len(f) < T_target else f[:T_target]

    vel = float(2.0 ** (1.0 - (p.velocity / 100.0)))                  # :765-788
    n_pre_f, n_pre_s = env_pre.shape[1], len(f0_pre)
    if abs(vel - 1.0) > 1e-6 and n_pre_f > 1 and n_pre_s > 1:
        env_new = stretch_prefix_2d(env_new, n_pre_f, vel)
        Tn = env_new.shape[1]
        for k in list(fm_new):
            f = stretch_prefix_1d(np.asarray(fm_new[k], dtype=np.float64), n_pre_f, vel)
            fm_new[k] = np.pad(f, (0, Tn - len(f)), mode="edge") if len(f) < Tn else f[:Tn]
        f0_new = stretch_prefix_1d(f0_new, n_pre_s, vel)
        mask_new = stretch_prefix_1d(mask_new, n_pre_s, vel)

    fm_new = _canon_formants(fm_new, T_target)                        # :791-833
    T = env_new.shape[1]
    tracks = [sanitize_formant(fm_new.get(nm, np.zeros(T)), T, sr, min_hz=lo, sigma_frames=4)
              for nm, lo in (("F1", 120.0), ("F2", 300.0), ("F3", 1500.0), ("F4", 2000.0))]
    fr = np.linspace(0.0, sr / 2.0, env_new.shape[0], dtype=F32)
    gain = np.ones_like(env_new, dtype=F32)
    for k, (tr, sv, sg) in enumerate(zip(tracks, p.formant_strength, (100.0, 200.0, 350.0, 500.0))):
        if abs(sv) < 1e-6:
            continue
        for t in range(T):
            fc = float(tr[t])
            if not np.isfinite(fc) or fc <= 50.0 or fc >= (sr * 0.5):
                continue
            w = np.exp(-0.5 * ((fr - fc) / sg) ** 2).astype(F32)
            gain[:, t] *= 1.0 + ((1.0 + sv) - 1.0) * w
    env_new *= gain

    n_tot = len(f0_new)                                               # pitch curve, :835-855
    t_s = np.arange(n_tot) / sr
    semis = p.bend.astype(np.float64) / 100.0 + p.pitch_m
    tc = p.flags.get("t", 0)
    if tc:
        semis = semis + (tc / 100.0)
    t_p = np.arange(len(semis)) * (60.0 / (p.tempo * 96.0))
    midi_curve = G.LinInterp(t_p, semis)(np.clip(t_s, t_p[0], t_p[-1]))
    f0_new = mask_new * midi_to_hz(midi_curve)

    dyn_gain = None                                                   # pd, :857-881
    if p.pitch_dyn != 0.0:
        base = p.pitch_m + ((p.flags.get("t", 0) or 0) / 100.0)
        bend_s = G.gauss1d((midi_curve - base).astype(F32), max(1, int(0.010 * sr)))
        ref = float(np.percentile(np.abs(bend_s), 95)) + 1e-8
        v = np.clip(bend_s / ref, -1.0, 1.0)
        db = (12.0 * abs(p.pitch_dyn)) * (v if p.pitch_dyn > 0 else -v)
        dyn_gain = np.clip(np.power(10.0, db / 20.0).astype(F32), 1e-3, 1e3)
        dyn_gain = 1.0 + (dyn_gain - 1.0) * G.gauss1d(mask_new.astype(F32), int(0.01 * sr))

    vf = float(p.flags.get("vf", 0))                                  # fry, :883-997
    vh = max(1.0, float(p.flags.get("vh", 50)))
    vl = np.clip(float(p.flags.get("vl", 15)), 0.0, 100.0)
    fry_mask = None
    if vf != 0:
        vf = float(np.clip(vf, -100.0, 100.0))
        n = len(f0_new)
        L = int(round(n * (abs(vf) / 100.0)))
        if L > 0:
            gl = int(np.clip(int(round(L * (vl / 100.0))), 0, L))
            cl = L - gl
            if vf > 0:
                if cl > 0:
                    f0_new[:cl] = vh * (mask_new[:cl] > 0)
                if gl > 0:
                    w = np.linspace(0.0, 1.0, gl, endpoint=True)
                    f0_new[cl:L] = (1.0 - w) * (vh * (mask_new[cl:L] > 0)) + w * f0_new[cl:L]
            else:
                st = n - L
                if gl > 0:
                    w = np.linspace(1.0, 0.0, gl, endpoint=True)
                    sl = slice(st, st + gl)
                    f0_new[sl] = (1.0 - w) * (vh * (mask_new[sl] > 0)) + w * f0_new[sl]
                if cl > 0:
                    f0_new[st + gl:n] = vh * (mask_new[st + gl:n] > 0)
        mid = n // 2
        if vf > 0:
            a, b = 0, max(0, min(n, int(round(mid * (vf / 100.0)))))
        else:
            a, b = max(0, n - int(round((n - mid) * (abs(vf) / 100.0)))), n
        if b > a:
            fry_mask = np.zeros(n, dtype=F32)
            fry_mask[a:b] = 1.0
            fade = int(0.01 * sr)
            if fade > 0:
                a1 = min(b, a + fade)
                if a1 > a:
                    fry_mask[a:a1] *= np.linspace(0.0, 1.0, a1 - a, endpoint=True)
                b0 = max(a, b - fade)
                if b > b0:
                    fry_mask[b0:b] *= np.linspace(1.0, 0.0, b - b0, endpoint=True)
    if fry_mask is not None and env_new.size:
        nb, nf = env_new.shape
        centers = np.minimum(len(fry_mask) - 1, (np.arange(nf) * hop + hop // 2)).astype(int)
        fm_fr = fry_mask[centers]
        bins = np.arange(nb, dtype=np.float64)
        for j in np.nonzero(fm_fr > 1e-6)[0]:
            s = 1.0 - float(fm_fr[j]) * (1.0 - 0.92)
            if abs(s - 1.0) < 1e-6:
                continue
            src = np.clip(bins / s, 0.0, nb - 1.0)
            lo = np.floor(src).astype(np.int32)
            hi = np.minimum(lo + 1, nb - 1)
            fr_ = src - lo
            col = env_new[:, j]
            env_new[:, j] = (1.0 - fr_) * col[lo] + fr_ * col[hi]

    return {"env": env_new, "f0": f0_new, "mask": mask_new, "formants": fm_new, "sr": sr, "seg": seg,
            "dyn_gain": dyn_gain, "fry_mask": fry_mask, "want_frames": want_f, "want_samples": want_s}


# ---------------------------------------------------------------------------------------------
# synthesis calls + post chain                                             SillySampler.py:1003-1182
# ---------------------------------------------------------------------------------------------
def _phi_for(seed, env, n, hop):
    T = 1 + n // hop          # frames the pulse STFT will have; env is matched to it
    return np.random.default_rng(seed).uniform(0.0, 2.0 * np.pi, size=(env.shape[0], T)).astype(F32)


def render(features, p: NoteParams, seed=0, n_fft=N_FFT, hop=HOP, return_parts=False):
    """One note: assemble -> synthesize (+ su / sj / sa layers) -> post chain -> mix.
    Returns the fp64 output array the reference hands to ``sf.write``."""
    a = assemble(features, p, hop)
    env, f0, mask, fm, sr = a["env"], a["f0"], a["mask"], a["formants"], a["sr"]
    n = len(mask)
    dummy = np.empty(n, dtype=np.bool_)
    shifts = dict(formant_shift=p.formant_shift, formants=fm, F1_shift=p.F_shift[0], F2_shift=p.F_shift[1],
                  F3_shift=p.F_shift[2], F4_shift=p.F_shift[3], n_fft=n_fft, hop_length=hop)
    phi = _phi_for(seed, env, n, hop)

    _, harm, uv, bre = G.synthesize(
        env, f0, mask, dummy, sr, **shifts,
        f0_jitter=p.f0_jitter, f0_jitter_strength=p.f0_jitter_strength,
        volume_jitter=p.volume_jitter, volume_jitter_strength_harm=p.volume_jitter_strength,
        volume_jitter_strength_breath=p.volume_jitter_strength * 2,
        add_subharm=p.add_subharm, subharm_weight=p.subharm_weight, subharm_semitones=12,
        subharm_vibrato=True, subharm_vibrato_rate=75, subharm_vibrato_depth=3, subharm_vibrato_delay=0.01,
        cut_subharm_below_f0=True, subharm_f0_jitter=0, normalize=p.normalize, phi=phi)
    stems0 = (harm.copy(), uv.copy(), bre.copy())

    def hp_pair(x, ref_f0):
        x = dynamic_filter(x, ref_f0, sr, cutoff_factor=1.0, order=6, btype="highpass")
        return dynamic_filter(x, ref_f0, sr, cutoff_factor=1.0, order=6, btype="highpass")

    if p.subharm_gain > 0.0:                                          # su, :1037-1059
        _, h2, _, _ = G.synthesize(env, f0 * 0.5, mask, dummy, sr, **shifts, normalize=p.normalize, phi=phi)
        harm += hp_pair(h2, np.maximum(f0, 120.0)) * p.subharm_gain

    if p.growl_mix > 0.0:                                             # sj, :1061-1081
        z = np.random.default_rng(seed).normal(loc=0.0, scale=p.growl_mix ** 2, size=len(f0))
        _, h3, _, _ = G.synthesize(env, f0 * (0.5 * (2.0 ** z)), mask, dummy, sr, **shifts,
                                   normalize=p.normalize, phi=phi)
        harm = (1.0 - p.growl_mix) * harm + p.growl_mix * hp_pair(h3, np.maximum(f0, 120.0))

    if a["fry_mask"] is not None:                                     # fry part 2, :1083-1099
        fmk = a["fry_mask"]
        ones = np.ones_like(f0)
        h_hp = dynamic_filter(harm, ones, sr, cutoff_factor=200, order=6, btype="highpass")
        b_hp = dynamic_filter(bre, ones, sr, cutoff_factor=200, order=6, btype="highpass")
        harm = harm * (1.0 - fmk) + h_hp * fmk
        bre = bre * (1.0 - fmk) + b_hp * fmk

    if p.sd_strength > 0:                                             # sd, :1101-1112
        j = G.volume_jitter_curve(len(bre), sr, speed=150.0, strength=p.sd_strength / 200.0, vibrato=True)
        bre *= 1.0 + (j - 1.0) * G.gauss1d(mask.astype(float), 20)
        bre *= 1.0 + (p.sd_strength / 100.0) * 10

    if p.tension != 0:                                                # st, :1114-1140
        before = G.rms(harm + bre)
        t = abs(p.tension)
        if p.tension < 0:
            order = np.clip(int(np.round(1 + (t * 4))), 1, 6)
            harm = dynamic_filter(harm, f0, sr, 2.0 - t * 0.75, order=order, btype="lowpass")
            bre = dynamic_filter(bre, f0, sr, t, order=4, btype="highpass")
        else:
            hi = dynamic_filter(harm, f0, sr, t * 4, order=4, btype="highpass")
            harm += hi * (1.0 + t * 20.0)
            bre = dynamic_filter(bre, f0, sr, (2.0 - t) / 0.5, order=6, btype="lowpass")
            bre *= (1.0 - t)
        after = G.rms(harm + bre)
        if after > 0:
            harm *= before / after
            bre *= before / after

    out = ((harm * p.harmonic_mix + bre * p.breathiness_mix) + uv * p.unvoiced_mix) * p.volume   # :1142-1151

    if p.aperiodic_mix > 0.0:                                         # sa, :1153-1172
        _, _, u4, b4 = G.synthesize(env, f0, np.ones_like(mask, dtype=mask.dtype), dummy, sr, **shifts,
                                    uv_strength=1.0, breath_strength=1.0, noise_transition_smoothness=1,
                                    normalize=p.normalize, phi=phi)
        out = out * (1.0 - p.aperiodic_mix) + ((u4 + b4) * p.volume) * p.aperiodic_mix

    if a["dyn_gain"] is not None:                                     # pd apply, :1174-1182
        dg = a["dyn_gain"]
        if len(dg) != len(out):
            xo = np.linspace(0.0, 1.0, num=len(dg), dtype=F32)
            xn = np.linspace(0.0, 1.0, num=len(out), dtype=F32)
            dg = G.LinInterp(xo, dg)(xn).astype(F32)
        out = out * dg
    if return_parts:
        return out, a, stems0
    return out
