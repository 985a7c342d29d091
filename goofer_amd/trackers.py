"""The cold-cache path: a voicebank sample that has no ``<stem>_features.goofy`` yet.

The reference analyses the wav on the first render and writes the cache (SillySampler.py:425-432; folder mode :214-240).
Half of that analysis is its own arithmetic — STFT magnitude, sigma-2 blur, mel-knot fit — and runs on the GPU here
(``core.envelope_features``).  The other half is Praat's (f0 by autocorrelation, formants by Burg's method,
through ``praat-parselmouth``): third-party, unpinned, not in this image — **parity unpinned** for those tracks (SURVEY §8 c).
This module is the plug for that half and the host logic the reference wraps around it:

* a *tracker* is ``fn(y, sr, hop_length, n_frames) -> (f0_track [frames'], {1..5: formant track [n_frames]})``;
  ``praat_tracker`` makes the reference's very calls when ``parselmouth`` is importable (GOOFER.py:341-353, 768-792);
  ``get()`` resolves the tracker to use (argument, ``GOOFER_TRACKER=module:function``, Praat when present);
* ``fix_f0_gaps`` (GOOFER.py:415-435) and ``per_sample_f0`` (GOOFER.py:957-966) — pinned by ``tests/golden/cold_cache.npz``,
  which ``make_golden.py`` generates by running the reference's own ``extract_features`` over a fake ``parselmouth``;
* ``analyse`` / ``ensure_features`` / ``extract_folder``: wav -> features -> byte-compatible ``.goofy`` next to the wav.

Nothing here re-implements Praat.
"""
from __future__ import annotations

import importlib
import importlib.util
import logging
import os
from pathlib import Path

import numpy as np

AUDIO_SUFFIXES = (".wav", ".flac", ".aiff", ".aif", ".mp3")      # SillySampler.py:211-212


class TrackerUnavailable(NotImplementedError):
    """No f0 / formant tracker can be had: parselmouth is not installed and none was supplied (a RuntimeError)."""


# -- trackers ----------------------------------------------------------------------------------------------------
def praat_tracker(y, sr, hop_length, n_frames):
    """The reference's Praat calls, argument for argument: ``Sound.to_pitch`` with the AC method, time step hop / sr, floor
    75 Hz, ceiling 950 Hz (GOOFER.py:341-353 — extract_features never forwards its own f0_max), and ``Sound.to_formant_burg``
    with the same time step and five formants, read back frame by frame at the frame's own time (GOOFER.py:768-792)."""
    try:
        import parselmouth
    except ImportError as e:                                      # noqa: PERF203
        raise TrackerUnavailable("praat-parselmouth is not installed") from e
    step = hop_length / sr
    snd = parselmouth.Sound(y, sr)
    burg = snd.to_formant_burg(time_step=step, max_number_of_formants=5)
    tracks = {k: [] for k in range(1, 6)}
    for frame in range(1, burg.get_number_of_frames() + 1):       # Praat numbers frames from 1
        t = burg.get_time_from_frame_number(frame)
        for k in tracks:
            try:
                v = burg.get_value_at_time(k, t)
            except Exception:                                     # noqa: BLE001 - "undefined" is a zero in the reference
                v = None
            tracks[k].append(0.0 if v is None else v)
    pitch = parselmouth.Sound(y, sr).to_pitch(method=parselmouth.Sound.ToPitchMethod.AC, time_step=step, pitch_floor=75,
                                              pitch_ceiling=950)
    return pitch.selected_array["frequency"], fit_formants(tracks, n_frames)


_REGISTRY = {"praat": praat_tracker}


def register(name: str, fn) -> None:
    _REGISTRY[name] = fn


def get(tracker=None):
    """The tracker to use: the argument if it is callable; else the name given (or ``$GOOFER_TRACKER``) looked up in the
    registry or imported as ``module:function``; else Praat when parselmouth is importable.  Raises TrackerUnavailable."""
    if callable(tracker):
        return tracker
    name = tracker or os.environ.get("GOOFER_TRACKER")
    if name:
        if name in _REGISTRY:
            return _REGISTRY[name]
        if ":" in name:
            mod, attr = name.split(":", 1)
            return getattr(importlib.import_module(mod), attr)
        raise TrackerUnavailable(f"unknown tracker {name!r} (registered: {sorted(_REGISTRY)}; or module:function)")
    if importlib.util.find_spec("parselmouth") is not None:
        return praat_tracker
    raise TrackerUnavailable("this sample has no _features.goofy and no f0 / formant tracker is available: install "
                             "praat-parselmouth (what the reference uses), set GOOFER_TRACKER=module:function, or run the "
                             "reference's extractor once")


# -- the reference's own arithmetic around the tracks -----------------------------------------------------------------
def fit_formants(tracks: dict, n_frames: int) -> dict:
    """Every track zero-padded / cut to the STFT frame count (GOOFER.py:783-790)."""
    out = {}
    for k, v in tracks.items():
        v = list(v)[:n_frames]
        out[k] = v + [0.0] * (n_frames - len(v))
    return out


def fix_f0_gaps(f0_track, max_gap: int = 4):
    """Runs of exact zeros no longer than ``max_gap`` that have a neighbour on both sides are bridged linearly between those
    neighbours (GOOFER.py:415-435); longer runs and runs touching either end stay zero."""
    f0 = np.array(f0_track, dtype=np.float64)
    zero = f0 == 0.0
    edges = np.flatnonzero(np.diff(np.concatenate([[False], zero, [False]]).astype(np.int8)))
    for a, b in zip(edges[0::2], edges[1::2]):                    # zero run [a, b)
        gap = int(b - a)
        if a > 0 and b < f0.size and gap <= max_gap:
            left, right = f0[a - 1], f0[b]
            for j in range(gap):
                r = (j + 1) / (gap + 1)
                f0[a + j] = left * (1 - r) + right * r
    return f0


def per_sample_f0(f0_track, n_samples: int, sr, f0_min=75, f0_merge_range=2):
    """(f0 per sample, voicing mask) from a frame-rate track (GOOFER.py:957-966): NaN -> 0, short gaps bridged, linear
    interpolation over linspace(0, duration) grids of the track and of the samples (0 outside), clip to [1e-5, 2000],
    voiced where the result exceeds ``f0_min``."""
    track = fix_f0_gaps(np.nan_to_num(np.asarray(f0_track, dtype=np.float64)), f0_merge_range)
    dur = n_samples / sr
    t_track, t_samp = np.linspace(0, dur, num=len(track)), np.linspace(0, dur, num=n_samples)
    if len(track) == 0:
        raise ValueError("x cannot be empty")                     # what gf.interp1d says about an empty track
    if len(track) == 1:                                           # a one-point "interpolant": the fill value except AT the point
        f0 = np.zeros(n_samples)
        f0[np.isclose(t_samp, t_track[0])] = track[0]
    else:
        inside = (t_samp >= t_track[0]) & (t_samp <= t_track[-1])
        f0 = np.zeros(n_samples)
        f0[inside] = np.interp(t_samp[inside], t_track, track)
    f0 = np.clip(f0, 1e-5, 2000)
    return f0, (f0 > f0_min).astype(float)


# -- wav in -------------------------------------------------------------------------------------------------------
def read_audio(path):
    """(mono float64 samples, sr).  soundfile when it is installed (what the reference reads with, any format it knows);
    otherwise PCM / float WAV through the standard library.  Channels are averaged like the reference does."""
    try:
        import soundfile as sf
        y, sr = sf.read(str(path))
    except ImportError:
        y, sr = _read_wav_stdlib(path)
    y = np.asarray(y, dtype=np.float64)
    return (y.mean(axis=1) if y.ndim > 1 else y), int(sr)


def _read_wav_stdlib(path):
    import wave
    with wave.open(str(path), "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        y = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif width == 2:
        y = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        y = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / float(1 << 23)
    elif width == 4:
        y = np.frombuffer(raw, dtype="<i4").astype(np.float64) / float(1 << 31)
    else:
        raise ValueError(f"unsupported WAV sample width {width}")
    return (y.reshape(-1, ch) if ch > 1 else y), sr


# -- analysis -> .goofy --------------------------------------------------------------------------------------------
def analyse(y, sr, n_fft=1024, hop_length=256, f0_min=75, f0_merge_range=2, tracker=None, ctx=None):
    """gf.extract_features (GOOFER.py:940-969): (env_spec fp64 [bins, T], f0 per sample, voicing mask, formants, env_knots).
    Envelope and knots on the GPU; tracks from ``tracker`` (see ``get``)."""
    from . import core
    track_fn = get(tracker)
    env_spec, env_knots = core.envelope_features(y, sr, n_fft, hop_length, ctx=ctx)
    n_frames = env_spec.shape[1]
    f0_track, formants = track_fn(np.asarray(y), sr, hop_length, n_frames)
    f0, vmask = per_sample_f0(f0_track, len(y), sr, f0_min, f0_merge_range)
    return env_spec, f0, vmask, fit_formants(dict(formants), n_frames), env_knots


def features_path(audio_path) -> Path:
    p = Path(audio_path)
    return p.with_name(f"{p.stem}_features.goofy")


def ensure_features(audio_path, n_fft=1024, hop_length=256, tracker=None, ctx=None) -> Path:
    """The sample's ``.goofy``: returned as is when it exists, else analysed from the wav and written the way the reference
    writes it (knots mode, fp16 f0 / mask, formant dict: save_features) — through a temporary file, so a concurrent render
    never loads half a cache."""
    from . import core
    feat = features_path(audio_path)
    if feat.exists():
        return feat
    if not Path(audio_path).exists():
        raise FileNotFoundError(f"{audio_path} not found (and no {feat.name} beside it)")
    track_fn = get(tracker)                                      # before any work: the usual reason a cold sample cannot render
    logging.info("Extracting features")
    y, sr = read_audio(audio_path)
    _, f0, vmask, forms, knots = analyse(y, sr, n_fft, hop_length, tracker=track_fn, ctx=ctx)
    # a name of its own per writer (two threads of one process may analyse the same cold sample), gone again if anything fails
    import tempfile
    fd, tmp_name = tempfile.mkstemp(prefix=feat.name + ".tmp", dir=str(feat.parent))
    os.close(fd)
    tmp = Path(tmp_name)
    try:
        core.save_features(tmp, knots, f0, vmask, forms, sr, len(y))
        os.replace(tmp, feat)
    finally:
        if tmp.exists():
            tmp.unlink()
    return feat


def extract_folder(path, tracker=None, ctx=None) -> dict:
    """Folder mode (SillySampler.py:214-240): every audio file under ``path`` (or the file itself) gets its ``.goofy``;
    existing ones are skipped, a failing file is logged and does not stop the others.  Files go one after the other — the
    analysis of a file is GPU work plus the tracker, and one context serves one host thread.  Returns the tallies."""
    root = Path(path)
    files = [f for f in (sorted(root.rglob("*")) if root.is_dir() else [root]) if f.is_file() and f.suffix.lower() in AUDIO_SUFFIXES]
    track_fn = get(tracker)                                      # fail before the first file, not at every file
    done = {"extracted": 0, "skipped": 0, "failed": 0}
    for f in files:
        if features_path(f).exists():
            logging.info(f"[SKIP] {features_path(f).name} already exists")
            done["skipped"] += 1
            continue
        try:
            logging.info(f"[EXTRACT] {f}")
            ensure_features(f, tracker=track_fn, ctx=ctx)
            done["extracted"] += 1
        except Exception as e:                                    # noqa: BLE001 - per-file isolation like the reference
            logging.error(f"[ERROR] Failed to extract {f.name}: {e}")
            done["failed"] += 1
    logging.info(f"[DONE] Extracted features from {len(files)} files.")
    return done
