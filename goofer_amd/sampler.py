"""UTAU resampler front half for the MI355X backend: flag decode and note-assembly *plans*.

Mirrors ``SillySampler.py``'s 13-argument call surface (``GooferResampler``, flag string, pitch-bend
string, CLI / HTTP argument split).  The host side only does what is inherently host work: string
decode, scalar flag scaling, and *planning* — integer cut points, per-frame gather indices with lerp
weights, per-note formant tracks (arrays of a few hundred values).  Every ``[bins x frames]`` or
per-sample array operation is executed on the GPU by ``goofer_assemble_batch`` + ``goofer_synth_batch``.

Reference: ``SillySampler.py:50-93`` (decode), ``:286-411`` (flag scaling), ``:449-500`` (slicing),
``:625-763`` (loop modes), ``:765-788`` (velocity), ``:791-833`` (formant strength), ``:835-855`` (pitch).
"""
from __future__ import annotations

import functools
import re
from dataclasses import dataclass, field

import numpy as np

N_FFT = 1024            # SillySampler.py:14-15 (not settable from the CLI)
HOP = N_FFT // 4
VERSION = "v2.6.1-mi355x"

_NOTE = re.compile(r"([A-G]#?)(-?\d+)")
_FLAG = re.compile(r"([A-Za-z]{1,4})([+-]?\d+)?")
_SEMI = {"C": 0, "C#": 1, "D": 2, "D#": 3, "E": 4, "F": 5, "F#": 6, "G": 7, "G#": 8, "A": 9, "A#": 10, "B": 11}

# the resampler's fixed sub-harmonic layer settings (SillySampler.py:1027-1033)
SUBHARM = {"semitones": 12, "vibrato": True, "rate": 75, "depth": 3, "delay": 0.01}


# ---------------------------------------------------------------------------------------------
# string decode (integer path — bit-exact)
# ---------------------------------------------------------------------------------------------
def parse_flags(text: str) -> dict:
    return {k: (int(v) if v else None) for k, v in _FLAG.findall(text.replace("/", ""))}


def _b64(c: str) -> int:
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


_B64_LUT = np.full(256, -1, dtype=np.int16)
for _c in range(256):
    try:
        _B64_LUT[_c] = _b64(chr(_c))
    except ValueError:
        pass


def _pitch_slow(text: str) -> np.ndarray:
    vals = []
    parts = text.split("#")
    for i in range(0, len(parts), 2):
        seg = parts[i]
        for j in range(0, len(seg), 2):
            v = (_b64(seg[j]) << 6) | _b64(seg[j + 1])
            vals.append(v - 4096 if v & 0x800 else v)
        if i + 1 < len(parts):
            vals += [vals[-1]] * int(parts[i + 1])
    a = np.array(vals, dtype=np.float32)
    return a if a.size else np.array([0.0], dtype=np.float32)


def pitch_string_to_cents(text: str) -> np.ndarray:
    """UTAU pitch-bend string: base64 pairs = 12-bit two's complement cents, ``#n#`` = repeat the last value n more times
    (SillySampler.py:56-84).  Well-formed strings decode as one table lookup over all pairs plus one np.repeat for the runs;
    anything else (odd segment, run without a value in front, bad character) goes through the character loop, which raises
    what the reference raises."""
    parts = text.split("#")
    segs, counts = parts[0::2], parts[1::2]
    if not text.isascii() or any(len(sg) & 1 for sg in segs) or not all(c.isdigit() for c in counts):
        return _pitch_slow(text)
    joined = "".join(segs)
    if not joined:
        return _pitch_slow(text)
    d = _B64_LUT[np.frombuffer(joined.encode("ascii"), dtype=np.uint8)]
    if (d < 0).any():
        return _pitch_slow(text)
    v = (d[0::2].astype(np.int32) << 6) | d[1::2]
    v = np.where(v & 0x800, v - 4096, v).astype(np.float32)
    if counts:
        ends = np.cumsum([len(sg) >> 1 for sg in segs[:len(counts)]]) - 1
        if ends[0] < 0:
            return _pitch_slow(text)                          # a run with nothing in front: IndexError, like the reference
        run = np.ones(v.size, dtype=np.int64)
        np.add.at(run, ends, np.array([int(c) for c in counts], dtype=np.int64))
        v = np.repeat(v, run)
    return v


def note_to_midi(name: str) -> int:
    m = _NOTE.match(name)
    if not m:
        raise ValueError(f"Bad note '{name}'")
    return (int(m.group(2)) + 1) * 12 + _SEMI[m.group(1)]


def midi_to_hz(m):
    return 440.0 * 2 ** ((m - 69) / 12)


def split_arguments(body: str) -> list:
    toks = body.split(" ")
    wavs = re.findall(r"([^\s]+\.wav)", " ".join(toks[:-11]))
    if len(wavs) < 2:
        raise ValueError("Missing .wav file paths in POST string")
    return wavs[:2] + toks[-11:]


def _ci(flags: dict, name: str, default=0):
    return next((v for k, v in flags.items() if k.lower() == name), default)


@dataclass
class Request:
    """One resampler call after flag scaling (GooferResampler.__init__)."""
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
    formant_shift: float = 1.0
    brightness_env: float = 1.0
    f_shift: tuple = (1.0, 1.0, 1.0, 1.0)
    breathiness_mix: float = 1.0
    unvoiced_mix: float = 1.0
    harmonic_mix: float = 1.0
    loop_mode: str = "concat"
    reverse: bool = False
    normalize: float = 1.0
    env_shape: float = 0.0
    force_voiced: bool = False
    formant_width: float = 0.0
    formant_strength: tuple = (0.0, 0.0, 0.0, 0.0)
    use_editor: bool = False
    f0_jitter: bool = False
    f0_jitter_strength: float = 0.0
    volume_jitter: bool = False
    volume_jitter_strength: float = 0.0
    add_subharm: bool = False
    subharm_weight: float = 0.0
    sd_strength: float = 0.0
    tension: float = 0.0
    growl_mix: float = 0.0
    aperiodic_mix: float = 0.0
    subharm_gain: float = 0.0
    pitch_dyn: float = 0.0
    fry: float = 0.0          # vf, clipped to [-100, 100]
    fry_hz: float = 50.0      # vh
    fry_glide: float = 15.0   # vl


@functools.lru_cache(maxsize=4096)
def _decode_flags(flags: str) -> dict:
    """Flag string -> the Request fields it sets (SillySampler.py:307-410).  Cached: a render job repeats a handful of flag
    strings over thousands of notes.  Exceptions (a bare flag letter -> TypeError, like the reference) are not cached."""
    fl = parse_flags(flags)
    g = fl.get
    r = {}
    r["use_editor"] = _ci(fl, "se") == 1
    r["formant_shift"] = 1.0 + (g("g", 0) / 200.0)                     # TypeError on a bare 'g', like the reference
    r["brightness_env"] = (g("br", 0) + 100) / 100.0
    r["f_shift"] = tuple(1.0 + (g(k, 0) / 100.0) for k in ("fa", "fb", "fc", "fd"))
    sh, sr_ = g("sh", None), g("sr", None)                              # roughness / harshness   :325-330
    r["f0_jitter"] = sh is not None and sh > 0
    r["f0_jitter_strength"] = (sh or 0) / 50.0
    r["volume_jitter"] = sr_ is not None and sr_ > 0
    r["volume_jitter_strength"] = (sr_ or 0) / 50.0
    sg = g("sg", 0)                                                     # sub-harmonic pulse layer  :364-366
    r["subharm_weight"] = (sg / 100.0) * 1.5
    r["add_subharm"] = sg > 0
    r["breathiness_mix"] = (g("B", 0) + 100) / 100.0
    r["unvoiced_mix"] = (g("U", 0) + 100) / 100.0
    r["harmonic_mix"] = float(np.clip(g("V", 100), 0, 100) / 100.0)
    lkey = next((k for k in fl if k.lower() == "l"), None)
    r["loop_mode"] = {1: "avg", 2: "stretch"}.get(fl[lkey], "concat") if lkey else "concat"
    r["reverse"] = g("R", 0) == 1
    r["normalize"] = float(np.clip(fl["P"], 0, 100) / 100.0) if "P" in fl else 1.0
    r["env_shape"] = float(np.clip(_ci(fl, "es") or 0, -100, 100)) / 100.0
    r["force_voiced"] = g("FV", 0) == 1
    r["formant_width"] = ((g("fw", 0) or 0) / 100.0) * 0.1
    glob = float(np.clip(_ci(fl, "fst") or 0, -100, 100)) / 100.0
    r["formant_strength"] = tuple(float(np.clip(glob + ((_ci(fl, "fst" + c) or 0) / 100.0), -1.0, 1.0)) for c in "abcd")
    # sample-domain post chain                                                         :332-334, 360-393, 884-888
    r["sd_strength"] = float(g("sd", None) or 0)
    r["tension"] = g("st", 0) / 100.0                                  # TypeError on a bare 'st', like the reference
    r["growl_mix"] = float(np.clip(g("sj", 0) or 0, 0, 100) / 100.0)
    r["aperiodic_mix"] = float(np.clip(g("sa", 0) or 0, 0, 100) / 100.0)
    r["subharm_gain"] = float(np.clip(g("su", 0) or 0, 0, 100) / 100.0)
    r["pitch_dyn"] = float(int(np.clip(_ci(fl, "pd") or 0, -100, 100))) / 100.0
    r["fry"] = float(np.clip(float(g("vf", 0)), -100.0, 100.0))
    r["fry_hz"] = max(1.0, float(g("vh", 50)))
    r["fry_glide"] = float(np.clip(float(g("vl", 15)), 0.0, 100.0))
    return {"flags": fl, "fields": r}


def decode_request(pitch, velocity, flags="", offset=0, length=1000, consonant=0, cutoff=0, volume=100,
                   modulation=0, tempo="!120", pitch_string="AA") -> Request:
    d = _decode_flags(flags)
    return Request(pitch_m=note_to_midi(pitch), velocity=float(velocity), flags=dict(d["flags"]), offset=float(offset) / 1000.0,
                   length=float(length) / 1000.0, consonant=float(consonant) / 1000.0, cutoff=float(cutoff) / 1000.0,
                   volume=float(volume) / 100.0, modulation=float(modulation) / 100.0, tempo=float(tempo.lstrip("!")),
                   bend=pitch_string_to_cents(pitch_string), **d["fields"])


def pitch_strings_to_cents(texts) -> list:
    """``pitch_string_to_cents`` for a batch: one call into the C-ABI library's host decoder (goofer_host_decode_bends) for
    all strings, views of one array back.  A string the decoder does not take (malformed, non-ASCII) goes through
    ``pitch_string_to_cents``, which raises — or answers — what the reference does."""
    lib = _host_lib()
    texts = list(texts)
    if lib is None or not texts or not all(t.isascii() for t in texts):
        return [pitch_string_to_cents(t) for t in texts]
    blob = "".join(texts).encode("ascii")
    off = np.zeros(len(texts) + 1, dtype=np.int64)
    np.cumsum([len(t) for t in texts], out=off[1:])
    out_off = np.empty(len(texts) + 1, dtype=np.int64)
    total = lib.goofer_host_decode_bends(blob, off.ctypes.data, len(texts), None, 0, out_off.ctypes.data)
    if total < 0:                                             # note -(total + 1) is not well formed: one by one
        return [pitch_string_to_cents(t) for t in texts]
    vals = np.empty(total, dtype=np.float32)
    lib.goofer_host_decode_bends(blob, off.ctypes.data, len(texts), vals.ctypes.data, total, out_off.ctypes.data)
    return [vals[out_off[i]:out_off[i + 1]] for i in range(len(texts))]


def decode_requests(arg_lists) -> list:
    """``decode_request`` for a batch of 13-argument lists (positions 2.. as in the resampler call: pitch, velocity, flags,
    offset, length, consonant, cutoff, volume, modulation, tempo, pitch_string) — same Requests, the pitch strings decoded
    together."""
    arg_lists = [tuple(a) for a in arg_lists]
    full = [a + _REQ_DEFAULTS[len(a) - 2:] if len(a) < 11 else a for a in arg_lists]
    bends = pitch_strings_to_cents([a[10] for a in full])
    out = []
    for a, bend in zip(full, bends):
        d = _decode_flags(a[2])
        out.append(Request(pitch_m=note_to_midi(a[0]), velocity=float(a[1]), flags=dict(d["flags"]), offset=float(a[3]) / 1000.0,
                           length=float(a[4]) / 1000.0, consonant=float(a[5]) / 1000.0, cutoff=float(a[6]) / 1000.0,
                           volume=float(a[7]) / 100.0, modulation=float(a[8]) / 100.0, tempo=float(a[9].lstrip("!")), bend=bend,
                           **d["fields"]))
    return out


_REQ_DEFAULTS = ("", 0, 1000, 0, 0, 100, 0, "!120", "AA")       # flags .. pitch_string of decode_request


# ---------------------------------------------------------------------------------------------
# a batch of requests as columns
# ---------------------------------------------------------------------------------------------
_SCALAR_FIELDS = ("pitch_m", "velocity", "offset", "length", "consonant", "cutoff", "volume", "modulation", "tempo", "formant_shift",
                  "brightness_env", "breathiness_mix", "unvoiced_mix", "harmonic_mix", "reverse", "normalize", "env_shape",
                  "force_voiced", "formant_width", "use_editor", "f0_jitter", "f0_jitter_strength", "volume_jitter",
                  "volume_jitter_strength", "add_subharm", "subharm_weight", "sd_strength", "tension", "growl_mix", "aperiodic_mix",
                  "subharm_gain", "pitch_dyn", "fry", "fry_hz", "fry_glide")
_FLAG_SCALARS = tuple(f for f in _SCALAR_FIELDS if f not in ("pitch_m", "velocity", "offset", "length", "consonant", "cutoff", "volume",
                                                             "modulation", "tempo"))
_LOOP_NAMES = ("concat", "avg", "stretch")


class RequestBatch:
    """n requests after flag scaling (what n ``Request`` objects hold) as columns: ``col[name]`` is a float64 array per scalar
    field of ``Request`` (booleans as 0 / 1), ``f_shift`` / ``formant_strength`` are [n, 4], ``loop_code`` the loop mode as the
    planner's code (0 concat, 1 avg, 2 stretch), ``t_cents`` the ``t`` flag (0 where absent), ``bend`` the notes' pitch bends
    back to back (float32) with ``bend_off`` [n + 1].  A render job plans, assembles and synthesises whole batches; nothing
    in that path wants a Python object per note."""

    __slots__ = ("n", "col", "f_shift", "formant_strength", "loop_code", "t_cents", "bend", "bend_off")

    def __len__(self):
        return self.n

    @classmethod
    def from_requests(cls, reqs):
        b = cls()
        b.n = n = len(reqs)
        b.col = {f: np.array([getattr(r, f) for r in reqs], dtype=np.float64).reshape(n) for f in _SCALAR_FIELDS}
        b.f_shift = np.array([r.f_shift for r in reqs], dtype=np.float64).reshape(n, 4)
        b.formant_strength = np.array([r.formant_strength for r in reqs], dtype=np.float64).reshape(n, 4)
        b.loop_code = np.array([_LOOP_NAMES.index(r.loop_mode) for r in reqs], dtype=np.int32).reshape(n)
        b.t_cents = np.array([r.flags.get("t", 0) or 0 for r in reqs], dtype=np.float64).reshape(n)
        lens = np.array([len(r.bend) for r in reqs], dtype=np.int64).reshape(n)
        b.bend_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        b.bend = np.concatenate([np.asarray(r.bend, dtype=np.float32) for r in reqs]) if n else np.zeros(0, np.float32)
        return b

    def request(self, i: int) -> Request:
        """Note i as a ``Request`` (flags dict: only ``t`` is carried)."""
        c = self.col
        kw = {f: c[f][i] for f in _SCALAR_FIELDS}
        for f in ("reverse", "force_voiced", "use_editor", "f0_jitter", "volume_jitter", "add_subharm"):
            kw[f] = bool(kw[f])
        for f in _SCALAR_FIELDS:
            if f not in ("reverse", "force_voiced", "use_editor", "f0_jitter", "volume_jitter", "add_subharm"):
                kw[f] = float(kw[f])
        kw["pitch_m"] = int(kw["pitch_m"])
        t = self.t_cents[i]
        return Request(flags={"t": int(t)} if t else {}, bend=self.bend[self.bend_off[i]:self.bend_off[i + 1]],
                       f_shift=tuple(float(v) for v in self.f_shift[i]), formant_strength=tuple(float(v) for v in self.formant_strength[i]),
                       loop_mode=_LOOP_NAMES[int(self.loop_code[i])], **kw)


def _text_blob(vals):
    """(ascii bytes of the strings joined by NUL, offsets[n + 1]: string i is blob[off[i]:off[i + 1]] with the separator behind it,
    which the library's parsers drop) or None when something is not a plain ASCII string without NULs.  The offsets come from
    one vectorised scan for the separators instead of a len() per string."""
    try:
        blob = "\x00".join(vals).encode("ascii")
    except (TypeError, UnicodeEncodeError):
        return None
    n = len(vals)
    pos = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
    if pos.size != max(0, n - 1):                               # a NUL inside an argument: the one-by-one path answers
        return None
    off = np.empty(n + 1, dtype=np.int64)
    off[0] = 0
    off[1:n] = pos + 1
    off[n] = len(blob)
    return blob, off


def _float_column(vals, strip_bang: bool = False):
    """float(v) of every element (strings as the CLI / HTTP front ends hand them over, or numbers): plain decimal literals in
    one call of the library's parser (strtod: correctly rounded, like float()), everything else — and everything when the
    library is not built — through float() itself, which answers or raises what the reference does."""
    n = len(vals)
    lib = _host_lib()
    tb = _text_blob(vals) if lib is not None and n else None
    if tb is None:
        return np.array([float(v.lstrip("!") if strip_bang else v) for v in vals], dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    ok = np.empty(n, dtype=np.uint8)
    bad = lib.goofer_host_parse_floats(tb[0], tb[1].ctypes.data, n, int(strip_bang), out.ctypes.data, ok.ctypes.data)
    if bad:
        for i in np.nonzero(ok == 0)[0]:
            out[i] = float(vals[i].lstrip("!") if strip_bang else vals[i])
    return out


def _bend_columns(texts):
    """The batch's pitch strings decoded back to back: (float32 values, offsets[n + 1]).  One pass of the library's decoder
    into a buffer sized for the usual case (two characters per value), a second only when run-length counts overflow it."""
    n = len(texts)
    lib = _host_lib()
    tb = _text_blob(texts) if lib is not None and n else None
    if tb is not None:
        out_off = np.empty(n + 1, dtype=np.int64)
        cap = len(tb[0]) + 64
        vals = np.empty(cap, dtype=np.float32)
        total = lib.goofer_host_decode_bends(tb[0], tb[1].ctypes.data, n, vals.ctypes.data, cap, out_off.ctypes.data)
        if total > cap:
            vals = np.empty(total, dtype=np.float32)
            lib.goofer_host_decode_bends(tb[0], tb[1].ctypes.data, n, vals.ctypes.data, total, out_off.ctypes.data)
        if total >= 0:
            return vals[:total], out_off
    bends = [pitch_string_to_cents(t) for t in texts]          # malformed / non-ASCII / no library: one by one (raises like the reference)
    lens = np.fromiter((len(v) for v in bends), dtype=np.int64, count=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return (np.concatenate([np.asarray(v, dtype=np.float32) for v in bends]) if n else np.zeros(0, np.float32)), off


def decode_request_batch(arg_lists) -> RequestBatch:
    """``decode_request`` for a batch of argument lists, straight to columns: the numeric strings are parsed per column, the
    flag strings and the note names once per distinct string (``_decode_flags`` cache), the pitch strings in one library call.
    Same values as ``RequestBatch.from_requests(decode_requests(arg_lists))`` (tested); an argument the reference would refuse
    raises here what ``decode_request`` raises."""
    arg_lists = [tuple(a) for a in arg_lists]
    full = [a + _REQ_DEFAULTS[len(a) - 2:] if len(a) < 11 else a for a in arg_lists]
    n = len(full)
    b = RequestBatch()
    b.n = n
    cols = list(zip(*full)) if n else [()] * 11
    # distinct note names / flag strings in first-seen order, and every note's index among them: dict.fromkeys and a mapped
    # dict lookup run inside the interpreter's C loops (a generator with setdefault per note was 0.1 ms per column)
    notes_u = {v: i for i, v in enumerate(dict.fromkeys(cols[0]))}
    note_ix = np.fromiter(map(notes_u.__getitem__, cols[0]), dtype=np.int64, count=n)
    midi = np.array([note_to_midi(v) for v in notes_u], dtype=np.float64)
    flags_u = {v: i for i, v in enumerate(dict.fromkeys(cols[2]))}
    flag_ix = np.fromiter(map(flags_u.__getitem__, cols[2]), dtype=np.int64, count=n)
    dec = [_decode_flags(v) for v in flags_u]
    c = {}
    c["pitch_m"] = midi[note_ix] if n else np.zeros(0)
    c["velocity"] = _float_column(cols[1])
    for name, k, div in (("offset", 3, 1000.0), ("length", 4, 1000.0), ("consonant", 5, 1000.0), ("cutoff", 6, 1000.0),
                         ("volume", 7, 100.0), ("modulation", 8, 100.0)):
        c[name] = _float_column(cols[k]) / div
    c["tempo"] = _float_column(cols[9], strip_bang=True)
    for f in _FLAG_SCALARS:
        c[f] = np.array([float(d["fields"][f]) for d in dec], dtype=np.float64)[flag_ix] if n else np.zeros(0)
    b.col = c
    take = lambda vals, width: (np.array(vals, dtype=np.float64).reshape(len(dec), width)[flag_ix] if n else np.zeros((0, width)))
    b.f_shift = take([d["fields"]["f_shift"] for d in dec], 4)
    b.formant_strength = take([d["fields"]["formant_strength"] for d in dec], 4)
    b.loop_code = (np.array([_LOOP_NAMES.index(d["fields"]["loop_mode"]) for d in dec], dtype=np.int32)[flag_ix] if n else np.zeros(0, np.int32))
    b.t_cents = (np.array([float(d["flags"].get("t", 0) or 0) for d in dec], dtype=np.float64)[flag_ix] if n else np.zeros(0))
    b.bend, b.bend_off = _bend_columns(cols[10])
    return b


# ---------------------------------------------------------------------------------------------
# planning
# ---------------------------------------------------------------------------------------------
def segment_indices(r: Request, sr: int, ylen: int, hop: int = HOP) -> dict:
    """Cut points (SillySampler.py:453-487): int() truncation, // hop, negative cutoff relative to the
    offset, R1 mirrors the window."""
    total = ylen / sr
    a0 = r.offset
    b0 = (r.offset - r.cutoff) if r.cutoff < 0 else (total - r.cutoff)
    if r.reverse:
        L = b0 - a0
        off = total - b0
        cut = total - (off + L)
    else:
        off, cut = r.offset, r.cutoff
    s0 = int(off * sr)
    s1 = s0 + int(r.consonant * sr)
    s2 = int(((off - cut) if cut < 0 else (total - cut)) * sr)
    return {"start_sample": s0, "consonant_sample": s1, "end_sample": s2,
            "start_frame": s0 // hop, "consonant_frame": s1 // hop, "end_frame": s2 // hop}


def _clip_slice(a: int, b: int, n: int):
    """Python slice semantics of x[a:b] on length n -> (start, stop) with stop >= start."""
    s = slice(a, b).indices(n)
    return s[0], max(s[0], s[1])


class Taps:
    """Sparse frame combinations: out[t] = sum_k w[t,k] * src[idx[t,k]] (k < 2 here).  ``f64`` says the
    reference holds this frame in float64 (lerps / cross-fades) rather than float32 (copies, L1 mean)."""

    def __init__(self, idx, w, f64):
        self.idx = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
        self.w = np.asarray(w, dtype=np.float64).reshape(-1, 2)
        self.f64 = bool(f64)

    @staticmethod
    def copy(rows):
        rows = np.asarray(rows, dtype=np.int64)
        return Taps(np.stack([rows, rows], 1), np.stack([np.ones(len(rows)), np.zeros(len(rows))], 1), False)

    def __len__(self):
        return self.idx.shape[0]

    def __getitem__(self, sl):
        return Taps(self.idx[sl], self.w[sl], self.f64)

    @staticmethod
    def concat(parts):
        parts = [p for p in parts]
        return Taps(np.concatenate([p.idx for p in parts]), np.concatenate([p.w for p in parts]),
                    any(p.f64 for p in parts if len(p)))


def _interp_taps(n_old: int, x_old: np.ndarray, x_new: np.ndarray):
    """np.interp(x_new, x_old, y) as (j, j+1, 1-c, c) with c = (x - x_j)/(x_{j+1} - x_j)."""
    j = np.clip(np.searchsorted(x_old, x_new, side="right") - 1, 0, max(n_old - 2, 0))
    if n_old == 1:
        z = np.zeros(len(x_new), dtype=np.int64)
        return z, z, np.ones(len(x_new)), np.zeros(len(x_new))
    c = (x_new - x_old[j]) / (x_old[j + 1] - x_old[j])
    c = np.where(x_new == x_old[j], 0.0, c)
    last = x_new >= x_old[-1]
    j = np.where(last, n_old - 2, j)
    c = np.where(last, 1.0, c)
    return j, j + 1, 1.0 - c, c


def _loop_frames(tail: Taps, want: int, mode: str) -> Taps:
    """Tail frames extended to ``want`` frames (SillySampler.py:631-696), on the index level."""
    n = len(tail)
    if n >= want:
        return tail[:want]
    reps, rem = want // n, want % n                     # ZeroDivisionError on an empty tail, like the reference
    rows = tail.idx[:, 0]
    if mode == "stretch":
        n_new = int(n * (want / n))
        jo, j1, w0, w1 = _interp_taps(n, np.linspace(0, 1, n), np.linspace(0, 1, n_new))
        return Taps(np.stack([rows[jo], rows[j1]], 1), np.stack([w0, w1], 1), True)
    if mode == "avg":
        tile = Taps(np.stack([rows, rows[::-1]], 1), np.full((n, 2), 0.5), False)
        return Taps.concat([tile] * reps + ([tile[:rem]] if rem else []))
    chain = [tail]
    for _ in range(reps - 1):
        prev = chain[-1]                                   # always a fresh, pure copy of the tail
        k = min(8, n // 2)
        if k == 0:
            # a one-frame tail (n == 1): the reference's fades are empty, prev[:, -0:] is all of prev — ONE column, which
            # broadcasts against the empty fade to an empty cross-fade — and prev[:, :-0] is empty as well, so the chunk is
            # the tail itself (SillySampler.py:657-672 under numpy's slicing / broadcasting rules)
            chain[-1] = tail
            chain.append(tail)
            continue
        up, dn = np.linspace(0, 1, k), np.linspace(1, 0, k)
        mixed = Taps(np.stack([prev.idx[len(prev) - k:, 0], tail.idx[:k, 0]], 1), np.stack([dn, up], 1), True)
        chain[-1] = Taps.concat([prev[:len(prev) - k], mixed, tail[k:]])
        chain.append(tail)
    if rem:
        last, prev = tail[:rem], chain[-1]
        k = min(8, rem // 2)
        if k > 0:
            up, dn = np.linspace(0, 1, k), np.linspace(1, 0, k)
            mixed = Taps(np.stack([prev.idx[len(prev) - k:, 0], last.idx[:k, 0]], 1), np.stack([dn, up], 1), True)
            chain[-1] = Taps.concat([prev[:len(prev) - k], mixed, last[k:]])
        else:
            chain[-1] = Taps.concat([prev, last])
    return Taps.concat(chain)


def _prefix_positions(n: int, pre_len: int, factor: float):
    pre_new = max(1, int(round(pre_len * factor)))
    idx = np.arange(pre_new + (n - pre_len), dtype=np.float64)
    return np.where(idx < pre_new, idx / factor, (idx - pre_new) + pre_len)


def _lin_interp(x, y, q, fill="extrapolate"):
    """The reference's interp1d (GOOFER.py:173-239) for small host-side tracks."""
    x, y, q = np.asarray(x), np.asarray(y), np.asarray(q)
    if x.size == 1:
        return np.full_like(q, y[0], dtype=y.dtype)
    sl = (y[1] - y[0]) / (x[1] - x[0] + 1e-10)
    sr_ = (y[-1] - y[-2]) / (x[-1] - x[-2] + 1e-10)
    out = np.interp(q, x, y)
    lo, hi = q < x[0], q > x[-1]
    if lo.any():
        out[lo] = y[0] + sl * (q[lo] - x[0])
    if hi.any():
        out[hi] = y[-1] + sr_ * (q[hi] - x[-1])
    return out


def gauss_taps(sigma: float, truncate: float = 4.0):
    r = int(truncate * sigma + 0.5)
    t = np.arange(-r, r + 1)
    k = np.exp(-0.5 * (t / sigma) ** 2)
    return k / k.sum()


@functools.lru_cache(maxsize=64)
def _gauss_taps_cached(sigma: float):
    return gauss_taps(sigma)


def _gauss_tracks(x, sigma):
    """gaussian_filter1d along the last axis of a [rows, T] array of short tracks (numpy 'reflect' padding), fp64: the taps are
    applied in ascending order, product then sum, row by row — every row gets the same additions in the same order whatever
    the number of rows, so a note's tracks are the same bits planned alone or inside a batch.  The loop itself lives in the
    C-ABI library (goofer_host_gauss_rows, host code); without the library the same arithmetic runs as numpy column ops."""
    k = _gauss_taps_cached(float(sigma))
    r = (k.size - 1) // 2
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = x.shape[-1]
    if x.size == 0:
        return x.copy()
    lib = _host_lib()
    if lib is not None:
        out = np.empty_like(x)
        rc = lib.goofer_host_gauss_rows(x.ctypes.data, x.size // T, T, k.ctypes.data, r, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("goofer_host_gauss_rows failed (%d)" % rc)
        return out
    pad = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(r, r)], mode="reflect")
    out = np.empty(x.shape, dtype=np.float64)
    flat_p, flat_o = pad.reshape(-1, pad.shape[-1]), out.reshape(-1, T)
    rows = max(1, (1 << 15) // max(T, 1))                     # row blocks that stay in cache; rows are independent
    tmp = np.empty((min(rows, flat_o.shape[0]), T), dtype=np.float64)
    for r0 in range(0, flat_o.shape[0], rows):
        pb, acc = flat_p[r0:r0 + rows], flat_o[r0:r0 + rows]
        tb = tmp[:acc.shape[0]]
        np.multiply(pb[:, 0:T], k[0], out=acc)
        for j in range(1, k.size):
            np.multiply(pb[:, j:j + T], k[j], out=tb)
            acc += tb
    return out


_HOST_LIB = [False]


def _host_lib():
    """The C-ABI library's host helpers, if the library is built (the planner itself must stay usable without it)."""
    if _HOST_LIB[0] is False:
        try:
            from . import _lib
            _HOST_LIB[0] = _lib.load()
        except Exception:
            _HOST_LIB[0] = None
    return _HOST_LIB[0]


def _fit_len(a, T):
    """pad_trim_to_len along the last axis (edge-pad / truncate, GOOFER.py:64-70); rows of zero length stay what the caller made them."""
    L = a.shape[-1]
    if L >= T:
        return a[..., :T]
    if L == 0:
        return np.pad(a, [(0, 0)] * (a.ndim - 1) + [(0, T)], mode="edge")   # numpy's own error for an empty edge pad
    return np.concatenate([a, np.repeat(a[..., -1:], T - L, axis=-1)], axis=-1)


def _repair_tracks(tracks, T, sr, min_hz):
    """SillySampler.py:264-279 for [rows, L] tracks at once: out-of-range / non-finite values are re-interpolated from the good
    ones (row by row: the repair is data dependent and rare), all-bad rows become 300 Hz.  Keeps the reference's aliasing:
    when ``tracks`` is already fp32 and at least T long, the repair edits the caller's array in place (those repaired values
    reach synthesize); an all-bad row is replaced in a fresh array and stays as it was."""
    max_hz = sr * 0.48
    x = np.asarray(tracks, dtype=np.float32)
    x = _fit_len(x, T)                                        # a view when long enough, a padded copy otherwise
    bad = (~np.isfinite(x)) | (x < min_hz) | (x > max_hz)
    all_bad = []
    for i in np.nonzero(bad.any(axis=-1))[0]:
        b = bad[i]
        good = np.where(~b)[0]
        if good.size:
            x[i, b] = _lin_interp(good.astype(np.float32), x[i, ~b], np.where(b)[0].astype(np.float32))
        else:
            all_bad.append(i)
    if all_bad:
        x = x.copy()
        x[all_bad] = 300.0
    return x


def _sanitize_tracks(tracks, T, sr, min_hz, sigma_frames=4):
    """sanitize_smooth_formant (SillySampler.py:264-283): repair, then the sigma-4 blur."""
    return _gauss_tracks(_repair_tracks(tracks, T, sr, min_hz), sigma_frames).astype(np.float32)


def _interp_rows(x_old, y, x_new):
    """np.interp(x_new, x_old, y[i]) for every row of y, by numpy's own formula (compiled_base.c: slope * (x - xp[j]) + fp[j],
    fp[j] itself on an exact hit or at the right end), plus interp1d's linear extrapolation (GOOFER.py:204-205).  fp64."""
    x_old, x_new = np.asarray(x_old, dtype=np.float64), np.asarray(x_new, dtype=np.float64)
    y = np.asarray(y)
    yd = y.astype(np.float64)
    n = x_old.size
    if n == 1:
        return np.repeat(yd[..., :1], x_new.size, axis=-1).astype(y.dtype if y.dtype == np.float64 else np.float64)
    j = np.clip(np.searchsorted(x_old, x_new, side="right") - 1, 0, n - 2)
    with np.errstate(invalid="ignore"):                        # (inf - inf of a broken track: nan, as np.interp gives, without the warning)
        slope = (yd[..., j + 1] - yd[..., j]) / (x_old[j + 1] - x_old[j])
        out = slope * (x_new - x_old[j]) + yd[..., j]
    out = np.where(x_new == x_old[j], yd[..., j], out)
    out = np.where(x_new >= x_old[-1], yd[..., -1:], out)
    lo, hi = x_new < x_old[0], x_new > x_old[-1]
    if lo.any():
        sl = (y[..., 1] - y[..., 0]) / (x_old[1] - x_old[0] + 1e-10)
        out[..., lo] = (y[..., :1] + sl[..., None] * (x_new[lo] - x_old[0]))
    if hi.any():
        sr_ = (y[..., -1] - y[..., -2]) / (x_old[-1] - x_old[-2] + 1e-10)
        out[..., hi] = (y[..., -1:] + sr_[..., None] * (x_new[hi] - x_old[-1]))
    return out


@dataclass
class NotePlan:
    req: Request
    seg: dict
    sr: int
    hop: int
    # edited source rows: rows [row_lo, row_hi) of the (possibly reversed) source envelope
    row_lo: int = 0
    row_hi: int = 0
    n_src_rows: int = 0
    # output frames: 4 taps into the edited rows (relative to row_lo)
    tap_idx: np.ndarray = None     # int32 [T, 4]
    tap_w: np.ndarray = None       # float64 [T, 4]
    env_f64: bool = False
    formants: np.ndarray = None    # float64 [T, 4] tracks handed to synthesize (after the in-place repair)
    fst_tracks: np.ndarray = None  # float32 [T, 4] sanitised + smoothed tracks of the formant-strength gain
    # samples
    n_out: int = 0
    n_pre: int = 0
    tail_len: int = 0
    want_samples: int = 0
    vel_factor: float = 1.0
    vel_active: bool = False
    n_before_vel: int = 0
    extra: dict = field(default_factory=dict)


def _geometry_key(req: Request, sr: int, ylen: int, n_src_frames: int, hop: int):
    """Everything the index plan depends on: notes that agree here share cut points, frame taps and sample counts and
    differ only in their formant tracks (and in pitch, which the plan does not touch)."""
    return (sr, ylen, n_src_frames, hop, req.offset, req.length, req.consonant, req.cutoff, req.velocity, req.loop_mode,
            bool(req.reverse), req.fry, req.fry_glide)


def _plan_geometry(req: Request, sr: int, ylen: int, n_src_frames: int, hop: int) -> NotePlan:
    """The part of ``resample``'s decisions that does not look at the formant tracks (SillySampler.py:449-788): cut points,
    loop-mode frame taps, velocity stretch, sample counts, fry ranges."""
    seg = segment_indices(req, sr, ylen, hop)
    p = NotePlan(req=req, seg=seg, sr=sr, hop=hop, n_src_rows=n_src_frames)
    T_src = n_src_frames
    f0a, f0b = _clip_slice(seg["start_frame"], seg["consonant_frame"], T_src)
    f1a, f1b = _clip_slice(seg["consonant_frame"], seg["end_frame"], T_src)
    pre = Taps.copy(np.arange(f0a, f0b))
    tail = Taps.copy(np.arange(f1a, f1b))
    want_f = int(np.ceil(req.length * sr / hop))
    tail_l = _loop_frames(tail, want_f, req.loop_mode)
    stage1 = Taps.concat([pre, tail_l])
    T_target = len(stage1)

    # samples
    s0a, s0b = _clip_slice(seg["start_sample"], seg["consonant_sample"], ylen)
    s1a, s1b = _clip_slice(seg["consonant_sample"], seg["end_sample"], ylen)
    p.n_pre, p.tail_len = s0b - s0a, s1b - s1a
    p.want_samples = int(req.length * sr)
    if p.tail_len < p.want_samples and p.tail_len == 0:
        raise ZeroDivisionError("integer division or modulo by zero")      # SillySampler.py:704
    p.extra["s_pre"], p.extra["s_tail"] = s0a, s1a
    p.n_before_vel = p.n_pre + p.want_samples

    # velocity prefix stretch   :765-788
    vel = float(2.0 ** (1.0 - (req.velocity / 100.0)))
    n_pre_f = len(pre)
    if abs(vel - 1.0) > 1e-6 and n_pre_f > 1 and p.n_pre > 1:
        n1 = len(stage1)
        pos = _prefix_positions(n1, n_pre_f, vel)
        jo, j1, w0, w1 = _interp_taps(n1, np.arange(n1, dtype=np.float64), pos)
        idx = np.concatenate([stage1.idx[jo], stage1.idx[j1]], axis=1)
        w = np.concatenate([stage1.w[jo] * w0[:, None], stage1.w[j1] * w1[:, None]], axis=1)
        p.tap_idx, p.tap_w, p.env_f64 = idx, w, True
        p.vel_active, p.vel_factor = True, vel
        pre_new = max(1, int(round(p.n_pre * vel)))
        p.n_out = pre_new + (p.n_before_vel - p.n_pre)
    else:
        z = np.zeros((len(stage1), 2))
        p.tap_idx = np.concatenate([stage1.idx, stage1.idx], axis=1)
        p.tap_w = np.concatenate([stage1.w, z], axis=1)
        p.env_f64 = stage1.f64
        p.n_out = p.n_before_vel
    # what the formant-track recipe needs
    p.extra.update(want_f=want_f, T_target=T_target, n_pre_f=n_pre_f, vel=vel)

    # edited-row window
    used = p.tap_idx[p.tap_w != 0.0] if p.tap_idx.size else np.zeros(0, dtype=np.int64)
    p.row_lo = int(used.min()) if used.size else 0
    p.row_hi = int(used.max()) + 1 if used.size else 0
    p.extra.update(fry_plan(req, sr, p.n_out))
    return p


def _plan_tracks(g: NotePlan, tracks: list):
    """The formant tracks of every note that shares the geometry ``g``: ``tracks[k]`` is a [notes, T_k] array of formant k + 1
    (fp64).  Slice, loop like the tail (but without the concat mode's duplicated frames), pad / trim, velocity-stretch, then
    the two things the device gets: the tracks gf.synthesize warps by, [notes, T_env, 4] fp64, and the repaired + smoothed
    tracks of the formant-strength gain, [notes, T_env, 4] fp32 (SillySampler.py:714-763, 771-806)."""
    req, seg, sr = g.req, g.seg, g.sr
    want_f, T_target, n_pre_f, vel = g.extra["want_f"], g.extra["T_target"], g.extra["n_pre_f"], g.extra["vel"]
    T_env = g.tap_idx.shape[0]
    n_notes = tracks[0].shape[0] if tracks else 0
    def recipe(src):
        """slice / loop / pad / velocity-stretch along the LAST axis of src ([..., T_src]); leading axes ride along."""
        if req.reverse:
            src = src[..., ::-1]
        pre_t = src[..., slice(seg["start_frame"], seg["consonant_frame"])]
        tr = np.asarray(src[..., slice(seg["consonant_frame"], seg["end_frame"])], dtype=np.float32)
        L = tr.shape[-1]
        lead = src.shape[:-1]
        if L == 0:
            lp = np.zeros(lead + (want_f,), dtype=np.float32)
        elif req.loop_mode == "stretch":
            factor = want_f / float(L)
            if factor == 1.0:
                lp = tr.copy()
            else:
                n_new = int(L * factor)
                lp = _interp_rows(np.linspace(0, 1, L), tr, np.linspace(0, 1, n_new)).astype(np.float32)
        else:
            reps, rem = want_f // L, want_f % L
            tile = (tr + tr[..., ::-1]) * 0.5 if req.loop_mode == "avg" else tr
            lp = np.tile(tile, (1,) * len(lead) + (reps,))
            if rem > 0:
                lp = np.concatenate([lp, tile[..., :rem]], axis=-1)
            lp = lp.astype(np.float32)
        f = np.concatenate([pre_t, lp], axis=-1)
        if f.shape[-1] == 0 and T_target > 0:
            # the reference edge-pads every track to the envelope's frames (SillySampler.py:755-760) and np.pad refuses an empty
            # one: a stretch-mode tail LONGER than wanted is cut for the envelope (:631-636) but resampled for the tracks
            # (:721-726), to int(L * (want / L)) frames — 0 for want = 1 and L = 49, 98, 103, ...; with no consonant frames
            # in front of it the track is empty and the render fails there
            raise ValueError("can't extend empty axis 0 using modes other than 'constant' or 'empty'")
        f = _fit_len(f, T_target) if f.shape[-1] else np.zeros(lead + (0,))
        if g.vel_active:
            Lk = f.shape[-1]
            if Lk > 1:
                f = _interp_rows(np.arange(Lk, dtype=np.float64), np.asarray(f, dtype=np.float64), _prefix_positions(Lk, n_pre_f, vel))
            else:
                f = np.asarray(f, dtype=np.float64)
            f = _fit_len(f, T_env) if f.shape[-1] else f
        return f

    arrs = [np.asarray(t) for t in tracks]
    if arrs and all(a.shape == arrs[0].shape and a.dtype == arrs[0].dtype for a in arrs):
        # the usual case: F1..F4(5) of a source have one length — one pass over [formants, notes, frames] instead of one per formant
        fm = list(recipe(np.stack(arrs)))
    else:
        fm = [recipe(a) for a in arrs]
    # canon + formant-strength tracks   :791-806 (canon uses the PRE-velocity frame count)
    canon = []
    for v in fm:
        a = np.asarray(v, dtype=np.float32)
        canon.append(_fit_len(a, T_target) if a.shape[1] else a)
    while len(canon) < 4:
        canon.append(None)
    F = np.zeros((n_notes, T_env, 4), dtype=np.float64)
    rep = np.empty((4, n_notes, T_env), dtype=np.float32)
    for c, lo in enumerate((120.0, 300.0, 1500.0, 2000.0)):
        a = canon[c]
        rep[c] = _repair_tracks(a if a is not None and a.shape[1] else np.zeros((n_notes, T_env), dtype=np.float32), T_env, sr, lo)
        if a is not None and a.shape[1]:
            F[:, :, c] = _fit_len(np.asarray(a, dtype=np.float64), T_env)      # after the repair: it may have edited `a` in place
    fst = np.ascontiguousarray(np.moveaxis(_gauss_tracks(rep, 4).astype(np.float32), 0, 2))   # one blur call for the four formants
    return F, fst


def _track_arrays(formants_list):
    """[{1..4: track}] of notes with one geometry -> four [notes, T_k] arrays (keys sorted, as the reference iterates them)."""
    keys = sorted(formants_list[0])
    if any(sorted(f) != keys for f in formants_list):
        return None
    out = []
    for k in keys:
        rows = [np.asarray(f[k]) for f in formants_list]
        if any(r.ndim != 1 or r.shape != rows[0].shape for r in rows):
            return None
        out.append(np.stack(rows))
    return out


def _finish_plan(g: NotePlan, req: Request, F, fst) -> NotePlan:
    p = NotePlan(req=req, seg=g.seg, sr=g.sr, hop=g.hop, row_lo=g.row_lo, row_hi=g.row_hi, n_src_rows=g.n_src_rows,
                 tap_idx=g.tap_idx, tap_w=g.tap_w, env_f64=g.env_f64, formants=F, fst_tracks=fst, n_out=g.n_out, n_pre=g.n_pre,
                 tail_len=g.tail_len, want_samples=g.want_samples, vel_factor=g.vel_factor, vel_active=g.vel_active,
                 n_before_vel=g.n_before_vel, extra=g.extra)
    return p


_GEO_CACHE = {}


def plan_notes(jobs, hop: int = HOP) -> list:
    """``plan_note`` for a batch: jobs = [(request, sr, ylen, n_src_frames, formants)].  Notes that agree on the geometry key
    (a render job repeats few lengths / cut points) share one index plan — taps, cut points, sample counts are the same
    arrays — and their formant tracks are processed together as [notes, frames] arrays.  Same results as plan_note note by
    note (tested bit for bit); ~30x less host time on uniform batches."""
    groups = {}
    for i, (req, sr, ylen, T_src, forms) in enumerate(jobs):
        groups.setdefault(_geometry_key(req, sr, ylen, T_src, hop), []).append(i)
    plans = [None] * len(jobs)
    for idxs in groups.values():
        req0, sr, ylen, T_src, _ = jobs[idxs[0]]
        key = _geometry_key(req0, sr, ylen, T_src, hop)
        g = _GEO_CACHE.get(key)
        if g is None:                                         # index plans are reused across calls: a song repeats its note lengths
            g = _plan_geometry(req0, sr, ylen, T_src, hop)
            if len(_GEO_CACHE) >= 8192:
                _GEO_CACHE.clear()
            _GEO_CACHE[key] = g
        arrs = _track_arrays([jobs[i][4] for i in idxs])
        if arrs is None:                                      # ragged / oddly keyed formant dicts: note by note
            for i in idxs:
                a1 = _track_arrays([jobs[i][4]])
                if a1 is None:
                    a1 = [np.atleast_2d(np.asarray(v)) for _, v in sorted(jobs[i][4].items())]
                F, fst = _plan_tracks(g, a1)
                plans[i] = _finish_plan(g, jobs[i][0], F[0], fst[0])
            continue
        F, fst = _plan_tracks(g, arrs)
        for j, i in enumerate(idxs):
            plans[i] = _finish_plan(g, jobs[i][0], F[j], fst[j])
    return plans


def plan_note(req: Request, sr: int, ylen: int, n_src_frames: int, formants_src: dict, hop: int = HOP) -> NotePlan:
    """Everything ``resample`` decides before touching an array (SillySampler.py:449-833), for one note."""
    return plan_notes([(req, sr, ylen, n_src_frames, formants_src)], hop)[0]


# ---------------------------------------------------------------------------------------------
# the same plans from the C-ABI library's host planner (csrc/planner.hip), a batch per call
# ---------------------------------------------------------------------------------------------
_LOOP_CODE = {"concat": 0, "avg": 1, "stretch": 2}


class PlannedBatch:
    """Plans of a batch as arrays: ``geo`` [n] records (``_lib.PLAN_GEOMETRY``), and per planned envelope row
    ``tap_idx`` int32 [rows, 4], ``tap_w`` fp64 [rows, 4], ``formants`` fp64 [rows, 4], ``fst`` fp32 [rows, 4]; note i owns
    rows geo["tap_off"][i] .. + geo["n_out_rows"][i]."""

    def __init__(self, geo, tap_idx, tap_w, formants, fst, owner=None):
        self.geo, self.tap_idx, self.tap_w, self.formants, self.fst = geo, tap_idx, tap_w, formants, fst
        self._owner = owner                                   # keeps library memory alive while the views are

    def note(self, i):
        g = self.geo[i]
        a, b = int(g["tap_off"]), int(g["tap_off"]) + int(g["n_out_rows"])
        return g, self.tap_idx[a:b], self.tap_w[a:b], self.formants[a:b], self.fst[a:b]


class _PlansHandle:
    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        if self.h:
            self.lib.goofer_host_plans_free(self.h)
            self.h = None


def source_tracks64(formants: dict):
    """The F1..F4 tracks of a source as the native planner takes them: four 1-D float64 arrays, or None when the dict is not
    the plain case (keys that do not start 1, 2, 3, 4 in sorted order, tracks of another dtype or shape) — the numpy planner
    then handles it with the reference's promotion and aliasing rules."""
    try:
        ks = sorted(formants)
    except TypeError:
        return None
    if ks[:4] != [1, 2, 3, 4]:
        return None
    out = []
    for k in (1, 2, 3, 4):
        v = formants[k]
        a = np.asarray(v)
        if a.ndim != 1 or a.dtype != np.float64:
            return None
        out.append(np.ascontiguousarray(a))
    t = _Tracks(out)
    t.ptrs = [a.ctypes.data if a.size else 0 for a in out]     # made once per source (ctypes objects are slow to make)
    t.lens = [a.size for a in out]
    return t


class _Tracks(tuple):
    """Four fp64 track arrays + their addresses / lengths as the planner records want them."""


def plan_native(records, hop: int, trim_rows: bool, keep=None, threads: int = 0):
    """goofer_host_plan_notes over ``records`` (``_lib.PLAN_REQUEST`` array).  Returns a PlannedBatch, or None when the library
    is not there or some note is a case the reference answers with an exception (the caller then runs the numpy planner,
    which raises it)."""
    import ctypes as C
    from . import _lib
    lib = _host_lib()
    if lib is None:
        return None
    records = np.ascontiguousarray(records, dtype=_lib.PLAN_REQUEST)
    n = records.shape[0]
    taps = np.ascontiguousarray(_gauss_taps_cached(4.0))
    h = C.c_void_p()
    rc = lib.goofer_host_plan_notes(records.ctypes.data, n, int(hop), int(bool(trim_rows)), taps.ctypes.data, (taps.size - 1) // 2,
                                    int(threads), C.byref(h))
    if rc != 0:
        return None
    owner = _PlansHandle(lib, h)
    g_p, ti_p, tw_p, f_p, fs_p, rows = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    lib.goofer_host_plans_view(h, C.byref(g_p), C.byref(rows), C.byref(ti_p), C.byref(tw_p), C.byref(f_p), C.byref(fs_p))
    R = rows.value

    def view(ptr, count, dtype):
        if count == 0 or not ptr.value:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype, count=count)

    geo = view(g_p, n, _lib.PLAN_GEOMETRY)
    if n and (geo["status"] != 0).any():
        return None
    return PlannedBatch(geo, view(ti_p, R * 4, np.int32).reshape(-1, 4), view(tw_p, R * 4, np.float64).reshape(-1, 4),
                        view(f_p, R * 4, np.float64).reshape(-1, 4), view(fs_p, R * 4, np.float32).reshape(-1, 4), owner=(owner, keep))


def plan_native_into(records, hop: int, trim_rows: bool, geo_out, row_capacity: int, tap_idx, tap_w, formants, fst, keep=None, threads: int = 0):
    """goofer_host_plan_into: the batch's plans written into the caller's arrays (pinned staging memory).  Returns the
    PlannedBatch over those arrays; None when the library steps aside (not built, a note the reference refuses: the caller runs
    the numpy planner); raises ``StagingFull(rows)`` when ``row_capacity`` rows do not hold the batch."""
    import ctypes as C
    from . import _lib
    lib = _host_lib()
    if lib is None:
        return None
    records = np.ascontiguousarray(records, dtype=_lib.PLAN_REQUEST)
    n = records.shape[0]
    taps = _gauss_taps_cached(4.0)
    rows = C.c_int64(0)
    rc = lib.goofer_host_plan_into(records.ctypes.data, n, int(hop), int(bool(trim_rows)), taps.ctypes.data, (taps.size - 1) // 2,
                                   int(threads), geo_out.ctypes.data, int(row_capacity), tap_idx.ctypes.data, tap_w.ctypes.data,
                                   formants.ctypes.data, fst.ctypes.data, C.byref(rows))
    if rc == 1:
        raise StagingFull(int(rows.value))
    if rc != 0 or (n and (geo_out["status"] != 0).any()):
        return None
    R = int(rows.value)
    return PlannedBatch(geo_out, tap_idx[:R], tap_w[:R], formants[:R], fst[:R], owner=keep)


class StagingFull(Exception):
    """The staging block handed to the planner is too small: args[0] = rows (or bytes) needed."""


def plan_records(reqs, srs, ylens, n_src_frames, tracks, track_ptrs=None, track_lens=None, skip_unused_fst: bool = False):
    """``_lib.PLAN_REQUEST`` records of a batch: the requests' scalars as columns (``reqs``: a RequestBatch or a list of
    Requests), ``tracks`` = per note the tuple of ``source_tracks64`` (kept alive by the caller) — or their addresses / lengths
    as [n, 4] arrays (``track_ptrs`` / ``track_lens``: what render.SourceArena keeps per resident sample).
    ``skip_unused_fst``: the smoothed track of a formant whose 'fst' strength is off for the note (|s| < 1e-6: the assembly's
    own test, SillySampler.py:817-830) is not computed — its column of ``fst`` stays 0 and is never read (the render path;
    the numpy planner always makes all four)."""
    from . import _lib
    rb = reqs if isinstance(reqs, RequestBatch) else RequestBatch.from_requests(reqs)
    n = rb.n
    c = rb.col
    rec = np.zeros(n, dtype=_lib.PLAN_REQUEST)
    rec["offset"], rec["length"], rec["consonant"], rec["cutoff"] = c["offset"], c["length"], c["consonant"], c["cutoff"]
    rec["fry"], rec["fry_glide"], rec["reverse"] = c["fry"], c["fry_glide"], c["reverse"]
    # Python's pow, like the reference (:765) — once per distinct velocity
    vel, inv = np.unique(c["velocity"], return_inverse=True)
    rec["vel_factor"] = np.array([float(2.0 ** (1.0 - (float(v) / 100.0))) for v in vel], dtype=np.float64)[inv] if n else 0.0
    rec["loop_mode"] = rb.loop_code
    rec["sr"], rec["ylen"], rec["n_src_frames"] = srs, ylens, n_src_frames
    if skip_unused_fst:
        rec["fst_skip"] = np.abs(rb.formant_strength) < 1e-6
    if track_ptrs is not None:
        rec["tracks"], rec["track_len"] = track_ptrs, track_lens
    else:
        rec["tracks"] = [t.ptrs for t in tracks]
        rec["track_len"] = [t.lens for t in tracks]
    return rec


def plan_notes_arrays(jobs, hop: int = HOP, trim_rows: bool = False, threads: int = 0) -> PlannedBatch:
    """The plans of a batch (jobs as for ``plan_notes``) as a PlannedBatch: from the library's host planner when every source's
    tracks are the plain float64 case, otherwise — and for every batch holding a note the reference refuses — from
    ``plan_notes`` (same values; tests/test_planner_native.py)."""
    tracks = [source_tracks64(j[4]) for j in jobs]
    if jobs and all(t is not None for t in tracks):
        uniq = {}
        tracks = [uniq.setdefault(tuple(t.ptrs) + tuple(t.lens), t) for t in tracks]
        rec = plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
        pb = plan_native(rec, hop, trim_rows, keep=(tracks, rec), threads=threads)
        if pb is not None:
            return pb
    return plans_to_arrays(plan_notes(jobs, hop), hop, trim_rows)


def plans_to_arrays(plans, hop: int, trim_rows: bool) -> PlannedBatch:
    """NotePlans (the numpy planner) as a PlannedBatch."""
    from . import _lib
    n = len(plans)
    geo = np.zeros(n, dtype=_lib.PLAN_GEOMETRY)
    ti, tw, F, fst = [], [], [], []
    live = {}
    off = 0
    for i, p in enumerate(plans):
        g = geo[i]
        T_env = p.tap_idx.shape[0]
        row_lo, row_hi = p.row_lo, p.row_hi
        if trim_rows and p.n_out > 0 and T_env > 1 + p.n_out // hop:
            T_env = 1 + p.n_out // hop
            lim = live.get(id(p.tap_idx))                      # notes of one geometry share the tap arrays
            if lim is None or lim[0] != T_env:
                used = p.tap_idx[:T_env][p.tap_w[:T_env] != 0.0]
                lim = live[id(p.tap_idx)] = (T_env, int(used.min()) if used.size else 0, int(used.max()) + 1 if used.size else 0)
            row_lo, row_hi = lim[1], lim[2]
        for k in ("start_sample", "consonant_sample", "end_sample", "start_frame", "consonant_frame", "end_frame"):
            g[k] = p.seg[k]
        g["tap_off"], g["n_rows"], g["n_out_rows"], g["row_lo"], g["row_hi"], g["env_f64"] = off, p.tap_idx.shape[0], T_env, row_lo, row_hi, int(p.env_f64)
        g["n_out"], g["n_pre"], g["s_pre"], g["s_tail"], g["tail_len"] = p.n_out, p.n_pre, p.extra["s_pre"], p.extra["s_tail"], p.tail_len
        g["want_samples"], g["n_before_vel"], g["vel_active"], g["vel_factor"] = p.want_samples, p.n_before_vel, int(p.vel_active), p.vel_factor
        g["pre_new"] = max(1, int(round(p.n_pre * p.vel_factor))) if p.vel_active else p.n_pre
        fx = p.extra
        g["fry_dir"], g["fry_const_lo"], g["fry_const_hi"] = fx["fry_dir"], fx["fry_const"][0], fx["fry_const"][1]
        g["fry_glide_lo"], g["fry_glide_hi"] = fx["fry_glide"]
        g["fry_a"], g["fry_b"], g["fry_fade"] = fx["fry_mask"][0], fx["fry_mask"][1], fx["fry_fade"]
        ti.append(p.tap_idx[:T_env]); tw.append(p.tap_w[:T_env]); F.append(p.formants[:T_env]); fst.append(p.fst_tracks[:T_env])
        off += T_env
    cat = lambda parts, dt: np.concatenate(parts).astype(dt, copy=False).reshape(-1, 4) if parts else np.zeros((0, 4), dtype=dt)
    return PlannedBatch(geo, cat(ti, np.int32), cat(tw, np.float64), cat(F, np.float64), cat(fst, np.float32))


def fry_plan(req: Request, sr: int, n: int) -> dict:
    """Sample ranges of the vocal-fry edit (SillySampler.py:883-955): where f0 is pinned to ``vh`` Hz, where it
    glides back, and the fry mask's support and fade.  Integer arithmetic, bit-exact with the reference."""
    out = {"fry_dir": 0, "fry_const": (0, 0), "fry_glide": (0, 0), "fry_mask": (0, 0), "fry_fade": 0}
    vf = req.fry
    if vf == 0:
        return out
    L = int(round(n * (abs(vf) / 100.0)))
    if L > 0:
        gl = int(np.clip(int(round(L * (req.fry_glide / 100.0))), 0, L))
        cl = L - gl
        out["fry_dir"] = 1 if vf > 0 else -1
        if vf > 0:
            out["fry_const"], out["fry_glide"] = (0, cl), (cl, L)
        else:
            st = n - L
            out["fry_glide"], out["fry_const"] = (st, st + gl), (st + gl, n)
    mid = n // 2
    if vf > 0:
        a, b = 0, max(0, min(n, int(round(mid * (vf / 100.0)))))
    else:
        a, b = max(0, n - int(round((n - mid) * (abs(vf) / 100.0)))), n
    if b > a:
        out["fry_mask"], out["fry_fade"] = (a, b), int(0.01 * sr)
    return out
