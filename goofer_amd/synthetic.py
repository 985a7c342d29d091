"""Synthetic voicebank sources and UTAU note requests (SURVEY.md §8 d).

Nothing here touches the reference or the oracle: these are the seeded inputs that the golden
generator, the parity tests and ``bench.py`` all share, so that "identical inputs" means
identical bytes on every box.

A *source* is what a ``<stem>_features.goofy`` file holds (reference layout,
``GOOFER.py:287-304``): mel-knot log envelope in fp16 ``[K, T]``, per-sample f0 / voicing mask,
per-frame formant tracks F1..F4.  A *note request* is the 13-argument UTAU resampler call
(``SillySampler.py:1226-1234``).
"""
from __future__ import annotations

import numpy as np

_B64 = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"

FORMANT_HZ = np.array([700.0, 1200.0, 2500.0, 3500.0])
FORMANT_AMP = np.array([0.5, 0.4, 0.3, 0.2])
FORMANT_BW = np.array([150.0, 200.0, 300.0, 400.0])

NOTE_NAMES = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]


def mel_knots_hz(sr: int, n_knots: int) -> np.ndarray:
    """Mel-spaced knot frequencies 0..sr/2 in fp32 (same grid the reference codec uses,
    ``GOOFER.py:74-82``; O'Shaughnessy mel, 2595*log10(1+f/700))."""
    top = 2595.0 * np.log10(1.0 + (sr / 2.0) / 700.0)
    mel = np.linspace(0.0, top, n_knots, dtype=np.float32)
    return (700.0 * (10 ** (mel / 2595.0) - 1.0)).astype(np.float32)


def encode_cents(cents) -> str:
    """UTAU pitch-bend string: two base64 chars per 12-bit two's-complement value, with
    ``#n#`` run-length for repeats (inverse of ``SillySampler.py:56-84``)."""
    vals = [int(v) for v in cents]
    out, i = [], 0
    while i < len(vals):
        v = vals[i]
        if not -2048 <= v <= 2047:
            raise ValueError("pitch bend out of 12-bit range")
        u = v & 0xFFF
        out.append(_B64[u >> 6] + _B64[u & 63])
        run = 0
        while i + 1 + run < len(vals) and vals[i + 1 + run] == v:
            run += 1
        if run >= 2:
            out.append("#%d#" % run)
            i += run
        i += 1
    return "".join(out)


def midi_name(m: int) -> str:
    return "%s%d" % (NOTE_NAMES[m % 12], m // 12 - 1)


def make_source(seed: int, sr: int = 44100, n_fft: int = 1024, hop: int = 256,
                seconds: float = 0.6, n_knots: int = 64) -> dict:
    """One synthetic voicebank sample in ``.goofy`` form.

    env(f, t) = exp(-f/3000) * (1 + sum_k a_k exp(-((f-F_k)/bw_k)^2/2)) * (1 + 0.1 sin(t/7 + phi))
    sampled at the mel knots, stored as log in fp16 — i.e. what the knot codec would hold.
    """
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sr))
    T = 1 + n // hop
    F = FORMANT_HZ * rng.uniform(0.9, 1.1, 4)
    phi = rng.uniform(0.0, 2.0 * np.pi)
    hz = mel_knots_hz(sr, n_knots).astype(np.float64)
    t = np.arange(T, dtype=np.float64)
    spec = np.exp(-hz / 3000.0)
    spec = spec * (1.0 + (FORMANT_AMP[None, :] * np.exp(-0.5 * ((hz[:, None] - F[None, :]) / FORMANT_BW[None, :]) ** 2)).sum(1))
    env = spec[:, None] * (1.0 + 0.1 * np.sin(t / 7.0 + phi))[None, :]
    walk = np.cumsum(rng.normal(0.0, 5.0, (4, T)), axis=1)
    formants = {k + 1: (F[k] + walk[k]).astype(np.float64) for k in range(4)}
    mask = np.ones(n, dtype=np.float32)
    mask[: int(0.08 * n)] = 0.0
    f0 = (220.0 * mask).astype(np.float32)
    return {
        "env_pack": {
            "mode": "knots",
            "knot_vals_log": np.log(env).astype(np.float16),
            "hz_knots": hz.astype(np.float32),
            "n_bins": n_fft // 2 + 1,
            "n_fft": n_fft,
            "sr": sr,
        },
        "f0": f0.astype(np.float16).astype(np.float32),
        "mask": mask.astype(np.float16).astype(np.float32),
        "formants": formants,
        "sr": sr,
        "y_len": n,
        "hop": hop,
        "n_fft": n_fft,
    }


def make_hard_source(seed: int, sr: int = 44100, n_fft: int = 1024, hop: int = 256, seconds: float = 0.6,
                     n_knots: int = 64) -> dict:
    """A voicebank sample with what real ``.goofy`` files hold and ``make_source`` does not: 3-6 interior
    voiced / unvoiced transitions of random length with short fp16 ramps between them and fractional mask plateaus
    (GOOFER.py:556-569, 1131-1144, 1179-1183), formant frames that are 0 / NaN / negative / above Nyquist, tracks that cross and
    one track in four that is invalid throughout (SillySampler.py:242-283, GOOFER.py:849-873), envelope frames 40 dB above / below
    their neighbours and a band of near-zero bins.  Seeded like make_source; same dict layout."""
    src = make_source(seed, sr, n_fft, hop, seconds, n_knots)
    rng = np.random.default_rng((seed << 8) ^ 0x5EED)
    n = src["y_len"]
    T = 1 + n // hop
    # voicing: alternating segments, the first one unvoiced or voiced at random
    k = int(rng.integers(3, 7))
    cuts = np.sort(rng.choice(np.arange(int(0.04 * n), int(0.97 * n)), size=k, replace=False))
    mask = np.empty(n, dtype=np.float64)
    level, a = float(rng.integers(0, 2)), 0
    plateau = [0.25, 0.5, 0.7, 0.999, 1.0, 1.0]
    for c in list(cuts) + [n]:
        mask[a:c] = level if level == 0.0 else plateau[int(rng.integers(0, len(plateau)))]
        level, a = (1.0 if level == 0.0 else 0.0), c
    for c in cuts:                                             # half of the transitions are 3-20 ms linear ramps, not steps
        if rng.random() < 0.5:
            w = int(rng.uniform(0.003, 0.02) * sr)
            lo, hi = max(0, c - w // 2), min(n, c + w // 2)
            if hi - lo > 1:
                mask[lo:hi] = np.linspace(mask[lo], mask[hi - 1], hi - lo)
    mask = mask.astype(np.float16).astype(np.float32)
    # formant tracks
    formants = {kk: v.copy() for kk, v in src["formants"].items()}
    for kk in (1, 2, 3, 4):
        tr = formants[kk]
        for bad in (0.0, np.nan, -120.0, 0.6 * sr, 30.0):
            i0 = int(rng.integers(0, T))
            tr[i0:i0 + int(rng.integers(1, 6))] = bad
    a0 = int(rng.integers(0, max(1, T - 12)))
    w = int(rng.integers(4, 12))
    f1, f2 = formants[1][a0:a0 + w].copy(), formants[2][a0:a0 + w].copy()
    formants[1][a0:a0 + w], formants[2][a0:a0 + w] = f2, f1      # crossing tracks
    if seed % 4 == 0:
        formants[int(rng.integers(1, 5))][:] = [np.nan, 0.0, 0.7 * sr][int(rng.integers(0, 3))]   # a track with no valid frame
    # envelope: 40 dB frame-to-frame jumps, a band of near-zero bins
    logk = src["env_pack"]["knot_vals_log"].astype(np.float64)
    for _ in range(int(rng.integers(2, 5))):
        t0 = int(rng.integers(0, T))
        logk[:, t0:t0 + int(rng.integers(1, 4))] += np.log(100.0) * (1.0 if rng.random() < 0.5 else -1.0)
    b0 = int(rng.integers(4, n_knots - 12))
    t0 = int(rng.integers(0, max(1, T - 20)))
    logk[b0:b0 + int(rng.integers(3, 10)), t0:t0 + int(rng.integers(5, 20))] += np.log(1e-6)
    out = dict(src)
    out["env_pack"] = dict(src["env_pack"], knot_vals_log=logk.astype(np.float16))
    out["mask"] = mask
    out["f0"] = (220.0 * mask).astype(np.float16).astype(np.float32)
    out["formants"] = formants
    return out


HARD_FLAGS = ["t0g0", "fa30fb-20fc10fd-10fw50fst40fsta20fstb-20fstc10fstd-10V80B20U-30", "L0g-30fst-50", "L1fa-25fc20B40", "L2fb30U30",
              "br40es-50fstd60", "br-40es60g40", "R1fa20fst30", "FV1fst40fa-30", "t-7g20fw-60", "L0es30fsta-40fb-30", "P50V70B-20fst60fc-20"]


def hard_case(i: int) -> tuple:
    """(source, request) of hard-source fixture ``i`` (tests/golden/sampler_hard_XX.npz; make_golden.gen_sampler_hard)."""
    rng = np.random.default_rng(8800 + i)
    src = make_hard_source(3000 + i, seconds=float(rng.uniform(0.35, 0.6)))
    req = make_request(3000 + i, HARD_FLAGS[i % len(HARD_FLAGS)], length_ms=float(rng.integers(250, 900)), offset_ms=float(rng.integers(0, 60)),
                       consonant_ms=float(rng.integers(20, 120)), cutoff_ms=float(rng.choice([-250, 30, 80])),
                       velocity=float(rng.choice([70, 100, 100, 130])), volume=float(rng.integers(60, 111)))
    return src, req


def make_request(seed: int, flags: str, length_ms: float = 1000.0, offset_ms: float = 50.0,
                 consonant_ms: float = 100.0, cutoff_ms: float = 100.0, velocity: float = 100.0,
                 volume: float = 100.0, tempo: float = 120.0) -> dict:
    """The 11 non-path arguments of one resampler call, seeded: uniform MIDI 55..72 and a smooth
    +-50 cent pitch-bend curve at 96 ticks per quarter note."""
    rng = np.random.default_rng(seed)
    midi = int(rng.integers(55, 73))
    total_s = (length_ms + consonant_ms) / 1000.0
    ticks = int(np.ceil(total_s * tempo * 96.0 / 60.0)) + 2
    ctrl = rng.uniform(-50.0, 50.0, max(2, ticks // 24 + 2))
    x = np.linspace(0.0, len(ctrl) - 1.0, ticks)
    bend = np.interp(x, np.arange(len(ctrl)), ctrl)
    bend = np.round(0.5 * (bend + np.roll(bend, 1))).astype(int)
    return {
        "pitch": midi_name(midi),
        "velocity": "%g" % velocity,
        "flags": flags,
        "offset": "%g" % offset_ms,
        "length": "%g" % length_ms,
        "consonant": "%g" % consonant_ms,
        "cutoff": "%g" % cutoff_ms,
        "volume": "%g" % volume,
        "modulation": "0",
        "tempo": "!%g" % tempo,
        "pitch_string": encode_cents(bend),
    }


def request_args(req: dict) -> list:
    """Positional tail (arguments 3..13) in the reference's order."""
    return [req[k] for k in ("pitch", "velocity", "flags", "offset", "length", "consonant",
                             "cutoff", "volume", "modulation", "tempo", "pitch_string")]


def phase_matrix(seed: int, n_bins: int, n_frames: int) -> np.ndarray:
    """The random-phase draw of the aperiodic branch, pinned: exactly the call the reference
    makes (``GOOFER.py:1151-1152``) but on a seeded generator.  Shape ``[n_bins, n_frames]``."""
    return np.random.default_rng(seed).uniform(0.0, 2.0 * np.pi, size=(n_bins, n_frames)).astype(np.float32)


# --- BASELINE.json configurations ------------------------------------------------------------

FULL_FORMANT_FLAGS = "fa30fb-20fc10fd-10fw50fst40fsta20fstb-20fstc10fstd-10V80B20U-30"


def _flip(flags: str, i: int) -> str:
    """Per-note sign flips for config 3: odd notes negate every signed formant flag."""
    if i % 2 == 0:
        return flags
    import re
    def neg(m):
        k, v = m.group(1), int(m.group(2))
        return "%s%d" % (k, v if k in ("V", "B", "U") else -v)
    return re.sub(r"([A-Za-z]+)([+-]?\d+)", neg, flags)


def config_flags(config: int, i: int) -> str:
    if config == 1:
        return "t0g0"
    if config == 2:
        return ("t12" if i % 2 == 0 else "t-12") + ("g50" if (i // 2) % 2 == 0 else "g-50")
    if config == 3:
        return _flip(FULL_FORMANT_FLAGS, i)
    if config == 4:
        return "L%d" % (i % 3)
    if config == 5:
        return "br40es-50" if i % 2 == 0 else "br-40es60"
    raise ValueError(config)


def config_geometry(config: int) -> dict:
    if config == 5:
        return {"sr": 96000, "n_fft": 2048, "hop": 96}
    return {"sr": 44100, "n_fft": 1024, "hop": 256}


def config_note(config: int, i: int, hard: bool = False) -> tuple:
    """(source, request, phi_seed) for note ``i`` of a BASELINE config (seeds 1000+i / 5000+i).  ``hard``: the same request on
    make_hard_source's version of the sample (parity tests; never the benchmark)."""
    geo = config_geometry(config)
    src = (make_hard_source if hard else make_source)(1000 + i, geo["sr"], geo["n_fft"], geo["hop"])
    length = 1000.0
    if config == 4:
        rng = np.random.default_rng(9000 + i)
        length = float(np.exp(rng.uniform(np.log(100.0), np.log(3000.0))))
    req = make_request(1000 + i, config_flags(config, i), length_ms=round(length))
    return src, req, 5000 + i


def config_note_frames(config: int, i: int) -> int:
    """Output frames of note ``i`` of a BASELINE config without building it (what a rank needs of EVERY note of a fixed job
    to derive the same longest-processing-time assignment as every other rank, shard.assign_lpt): the request's length and
    consonant in samples (SillySampler.py:449-500, velocity 100), 1 + n // hop frames."""
    geo = config_geometry(config)
    length = 1000.0
    if config == 4:
        rng = np.random.default_rng(9000 + i)
        length = float(np.exp(rng.uniform(np.log(100.0), np.log(3000.0))))
    n = int(100.0 / 1000.0 * geo["sr"]) + int(round(length) / 1000.0 * geo["sr"])
    return 1 + n // geo["hop"]


def random_flags(rng) -> str:
    """A random subset (3-8 flags) of the reference's flag vocabulary with in-range values (SillySampler.py:307-410):
    used by the flag-interaction fixtures (tests/golden/make_golden.py) and the fuzz tests."""
    pool = {
        "g": lambda: int(rng.integers(-80, 81)), "t": lambda: int(rng.integers(-300, 301)), "br": lambda: int(rng.integers(-60, 61)),
        "es": lambda: int(rng.integers(-80, 81)), "fw": lambda: int(rng.integers(-80, 81)), "fa": lambda: int(rng.integers(-30, 31)),
        "fb": lambda: int(rng.integers(-30, 31)), "fc": lambda: int(rng.integers(-20, 21)), "fd": lambda: int(rng.integers(-20, 21)),
        "fst": lambda: int(rng.integers(-60, 61)), "fsta": lambda: int(rng.integers(-40, 41)), "fstc": lambda: int(rng.integers(-40, 41)),
        "V": lambda: int(rng.integers(40, 101)), "B": lambda: int(rng.integers(-50, 51)), "U": lambda: int(rng.integers(-50, 51)),
        "P": lambda: int(rng.integers(0, 101)), "L": lambda: int(rng.integers(0, 3)), "R": lambda: int(rng.integers(0, 2)),
        "FV": lambda: int(rng.integers(0, 2)), "sh": lambda: int(rng.integers(10, 80)), "sr": lambda: int(rng.integers(10, 80)),
        "sg": lambda: int(rng.integers(10, 80)), "su": lambda: int(rng.integers(10, 80)), "sj": lambda: int(rng.integers(10, 60)),
        "sa": lambda: int(rng.integers(10, 60)), "st": lambda: int(rng.integers(-80, 81)), "sd": lambda: int(rng.integers(10, 60)),
        "vf": lambda: int(rng.integers(-60, 61)), "vh": lambda: int(rng.integers(30, 80)), "vl": lambda: int(rng.integers(0, 60)),
        "pd": lambda: int(rng.integers(-80, 81)),
    }
    keys = list(pool)
    chosen = rng.choice(len(keys), size=int(rng.integers(3, 9)), replace=False)
    return "".join(f"{keys[i]}{pool[keys[i]]()}" for i in sorted(chosen))


def with_unvoiced_gaps(src: dict, share: float, seed: int) -> dict:
    """A copy of a synthetic source with `share` of its samples unvoiced in 50 ms gaps at seeded random places (mask and f0
    zeroed): consonant-like holes.  The BASELINE notes are fully voiced behind their offset; this is the variant bench.py and
    scripts/voicing_sweep.py time beside them, because the noise walker skips the transform of a stem whose gain is exactly zero."""
    out = dict(src)
    n = src["y_len"]
    m, f = src["mask"].copy(), src["f0"].copy()
    rng = np.random.default_rng(seed + 7)
    gap = max(1, int(0.05 * src["sr"]))
    for _ in range(int(round(share * n / gap))):
        a0 = int(rng.integers(0, max(1, n - gap)))
        m[a0:a0 + gap] = 0.0
        f[a0:a0 + gap] = 0.0
    out["mask"], out["f0"] = m, f
    return out
