"""The GOOFER.py analyse/synth call surface, served by the MI355X backend.

Drop-in for the functions ``SillySampler.py`` / ``test.py`` call on the reference module
(``import goofer_amd.core as gf``): same names, positional order, array layouts (``[bins, frames]``
numpy arrays in, numpy arrays out) and error behaviour, but every array operation on the hot path
runs in libgoofer_hip.so on the GPU.  Host code here only marshals: it never computes audio.

Reference: GOOFER.py:355-413 (stft/istft), :473-554 (pulse train), :149-168 (knot decode),
:971-1220 (synthesize).  Per-call overheads (H2D/D2H of one note) make this single-note surface a
convenience; throughput comes from :func:`synthesize_batch`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .device import Context, default_context, default_params, row_stride

DSTORAGE = np.float16
DCOMPUTE = np.float32

def _ctx(sr, n_fft, hop, ctx=None) -> Context:
    return (ctx or default_context()).plan(sr, n_fft, hop)


MASK_DS = 4          # smooth_mask_ds decimation (GOOFER.py:556)


# -- feature files (host-side format code, byte-compatible with GOOFER.py:287-339) -----------------
def _formant_slot(key):
    """1..4 for a formant-track key (an int, or a string 'F<n>' / 'f<n>'), else None."""
    if isinstance(key, str):
        if not key[:1] in ("F", "f"):
            return None
        try:
            key = int(key[1:])
        except Exception:
            return None
    if isinstance(key, (int, np.integer)) and 1 <= int(key) <= 4:
        return int(key)
    return None


def formants_to_int_keys(d):
    """{1..4: ndarray} from a formant dict with int or 'F<n>' keys; absent tracks become one zero (GOOFER.py:48-62).
    The tracks keep the order the input listed them in (the dict is pickled into the .goofy file)."""
    slots = ((_formant_slot(k), v) for k, v in d.items()) if isinstance(d, dict) else ()
    tracks = {slot: np.asarray(v) for slot, v in slots if slot is not None}
    tracks.update({i: np.zeros(1, dtype=np.float64) for i in (1, 2, 3, 4) if i not in tracks})
    return tracks


def save_features(path, features, f0_interp, voicing_mask, formants, sr, y_len):
    fields = {}
    if isinstance(features, dict) and features.get("mode") == "knots":
        fields.update(mode=np.array(["knots"]), knot_vals_log=features["knot_vals_log"], hz_knots=features["hz_knots"],
                      n_bins=np.array([features["n_bins"]], dtype=np.int32),
                      n_fft=np.array([features["n_fft"]], dtype=np.int32),
                      env_sr=np.array([features["sr"]], dtype=np.int32))
    else:
        dense = np.asarray(features, dtype=DSTORAGE)
        fields.update(mode=np.array(["full"]), env_spec=dense, n_fft=np.array([dense.shape[0] * 2 - 2], dtype=np.int32))
    fields.update(f0_interp=np.asarray(f0_interp).astype(DSTORAGE), voicing_mask=np.asarray(voicing_mask).astype(DSTORAGE),
                  formants=formants_to_int_keys(formants), sr=np.array([sr], dtype=np.int32),
                  y_len=np.array([y_len], dtype=np.int64))
    with open(path, "wb") as fh:
        np.savez_compressed(fh, **fields)


def load_features(path):
    z = np.load(path, allow_pickle=True)
    if str(z["mode"][0]) == "knots":
        env = {"mode": "knots", "knot_vals_log": z["knot_vals_log"], "hz_knots": z["hz_knots"],
               "n_bins": int(z["n_bins"][0]), "n_fft": int(z["n_fft"][0]), "sr": int(z["env_sr"][0])}
    else:
        env = np.asarray(z["env_spec"], dtype=DCOMPUTE)
    return (env, np.asarray(z["f0_interp"], dtype=DCOMPUTE), np.asarray(z["voicing_mask"], dtype=DCOMPUTE),
            formants_to_int_keys(z["formants"].item()), int(z["sr"][0]), int(z["y_len"][0]))


# -- single-array entry points ---------------------------------------------------------------------
def _check_window(window, n_fft):
    """The kernels window with the plan's sqrt-Hann (GOOFER.py:12-18: ``np.sqrt(np.hanning(n_fft))`` in fp32, the only window the
    reference ever passes to stft / istft, :1099, :1146).  A caller's ``window`` must BE that window (to fp32 rounding): any other
    one used to be silently ignored, which is a wrong answer — it raises instead."""
    if window is None:
        return
    w = np.asarray(window)
    ref = np.sqrt(np.hanning(n_fft)).astype(np.float32)
    if w.shape != ref.shape or not np.allclose(w.astype(np.float64), ref.astype(np.float64), rtol=0.0, atol=3e-7):
        raise ValueError("goofer_amd.stft / istft support the reference's cached window only: sqrt(hanning(n_fft)) in fp32 "
                         "(GOOFER.py:12-18); got a different window of shape %s" % (w.shape,))


def stft(x, n_fft=2048, hop_length=512, window=None, sr=44100, ctx=None):
    """complex64 ``[bins, T]``.  ``window``: None or the reference's cached sqrt-Hann (anything else raises: `_check_window`)."""
    _check_window(window, n_fft)
    c = _ctx(sr, n_fft, hop_length, ctx)
    x = np.asarray(x, dtype=np.float32)
    n = len(x)
    T = 1 + n // hop_length
    S = c.rfft_frames(c.tensor(x), c.tensor(np.array([0, n], dtype=np.int64)), c.tensor(np.array([0, T], dtype=np.int64)), T)
    return np.ascontiguousarray(S.cpu().numpy().T)


def istft(S, hop_length=512, window=None, length=None, sr=44100, ctx=None):
    S = np.asarray(S, dtype=np.complex64)
    n_fft = (S.shape[0] - 1) * 2
    _check_window(window, n_fft)
    c = _ctx(sr, n_fft, hop_length, ctx)
    T = S.shape[1]
    n = hop_length * (T - 1) if length is None else int(length)
    # the backend derives T from the note length (T = 1 + n//hop); feed it a length in that class
    n_dev = hop_length * (T - 1)
    St = c.tensor(np.ascontiguousarray(S.T))
    y = c.irfft_ola(St, c.tensor(np.array([0, n_dev], dtype=np.int64)), c.tensor(np.array([0, T], dtype=np.int64)), n_dev)
    y = y.cpu().numpy()
    if n > n_dev:
        y = np.pad(y, (0, n - n_dev))
    return y[:n]


def pulse_train_numba(f0_interp, sr, Ra=0.02, Rg=1.7, Rk=0.8, ctx=None):
    """gf.pulse_train_numba (GOOFER.py:473-554).  Ra, Rg, Rk other than gf.synthesize's constants rebuild the handle's pulse
    tables for this call (Context.pulse_model) and put the constants back behind it."""
    c = ctx or default_context()
    if c.geom is None:
        c.plan(int(sr), 1024, 256)
    elif c.geom[0] != int(sr):
        c.plan(int(sr), c.geom[1], c.geom[2])
    f0 = np.asarray(f0_interp, dtype=np.float32)
    if f0.size == 0:
        return np.zeros(0, dtype=np.float32)
    before = getattr(c, "lf", (0.02, 1.7, 0.8))
    c.pulse_model(Ra, Rg, Rk)
    try:
        return c.pulse_train(c.tensor(f0), c.tensor(np.array([0, f0.size], dtype=np.int64))).cpu().numpy()
    finally:
        c.pulse_model(*before)


def decode_env_from_knots(env_pack, ctx=None):
    """fp32 ``[bins, T]`` from a knots dict (GOOFER.py:149-168)."""
    assert env_pack["mode"] == "knots"
    c = _ctx(int(env_pack["sr"]), int(env_pack["n_fft"]), int(env_pack["n_fft"]) // 4, ctx)
    knots = np.ascontiguousarray(np.asarray(env_pack["knot_vals_log"]).astype(np.float16).T)
    env = c.knot_decode(c.tensor(knots.view(np.uint16)).view(torch.float16), np.asarray(env_pack["hz_knots"], dtype=np.float32))
    out = env.cpu().numpy().T
    return np.ascontiguousarray(out[: int(env_pack["n_bins"])])


def gaussian_taps(sigma, truncate=4.0):
    """Normalised fp64 taps, radius int(truncate*sigma + 0.5) (GOOFER.py:247-252)."""
    r = int(truncate * sigma + 0.5)
    t = np.arange(-r, r + 1)
    k = np.exp(-0.5 * (t / sigma) ** 2)
    return k / k.sum()


# -- analysis: the half of extract_features that is not Praat (GOOFER.py:940-969, 97-147) ------------------
# -- small numeric helpers the reference exposes at module level (SillySampler.py / SillyEditor.py call them) ---------
def to_compute(x):
    return np.asarray(x, dtype=np.float32)                      # GOOFER.py:72


def hz_to_mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)                  # GOOFER.py:74


def mel_to_hz(m):
    return 700.0 * (10 ** (m / 2595.0) - 1.0)                   # GOOFER.py:75


def rms(x):
    return float(np.sqrt(np.mean(np.square(x)) + 1e-12))        # GOOFER.py:170-171 (a scalar reduction: host)


def gaussian_filter1d(input_array, sigma, axis=-1, truncate=4.0, ctx=None):
    """gf.gaussian_filter1d (GOOFER.py:241-261) on the device: numpy-'reflect' padding, fp64 accumulate in tap order,
    every 1-D line along ``axis`` filtered independently.  Returns float64 (complex128 for complex input)."""
    arr = np.asarray(input_array)
    radius = int(truncate * sigma + 0.5) if sigma > 0.0 else 0
    if radius <= 0 or arr.size == 0:                            # nothing to blur: a copy, dtype untouched
        return arr.copy()
    if np.iscomplexobj(arr):
        return (gaussian_filter1d(arr.real, sigma, axis, truncate, ctx) +
                1j * gaussian_filter1d(arr.imag, sigma, axis, truncate, ctx))
    c = ctx or default_context()
    moved = np.moveaxis(arr, axis, -1)
    lines = np.ascontiguousarray(moved, dtype=np.float64).reshape(-1, moved.shape[-1])
    out = c.gauss_rows_f64(c.tensor(lines), gaussian_taps(sigma, truncate)).cpu().numpy()
    return np.moveaxis(out.reshape(moved.shape), -1, axis)


def smooth_mask_ds(mask, sigma=100, ds=4, sr=44100, n_fft=1024, hop_length=256, ctx=None):
    """gf.smooth_mask_ds (GOOFER.py:556-569) on the device: mask[::4] -> Gaussian max(1, sigma / 4) -> linear upsample."""
    if ds != MASK_DS:
        raise ValueError("the device path decimates by %d (GOOFER.py:556 default)" % MASK_DS)
    c = _ctx(sr, n_fft, hop_length, ctx)
    m = np.ascontiguousarray(mask, dtype=np.float32)
    if m.size == 0:
        return m.copy()
    return c.smooth_mask_ds(c.tensor(m), sigma=float(sigma)).cpu().numpy()


def _axis_sigmas(sigma):
    """(sigma of axis 0, sigma of axis 1), negatives clamped to 0, from a number or a pair."""
    pair = tuple(sigma) if isinstance(sigma, (list, tuple)) else (sigma, sigma)
    if len(pair) != 2:
        raise ValueError("sigma must be a float or a 2-tuple for 2D arrays.")
    return tuple(max(float(v), 0.0) for v in pair)


def gaussian_filter(input_array, sigma, ctx=None):
    """gf.gaussian_filter (GOOFER.py:263-285): the separable blur of a matrix, one gaussian_filter1d pass (on the device)
    per axis whose sigma is positive."""
    arr = np.asarray(input_array)
    if arr.ndim != 2:
        raise ValueError("gaussian_filter expects a 2D array.")
    if 0 in arr.shape:
        return arr.copy()
    for axis, s in enumerate(_axis_sigmas(sigma)):
        if s > 0.0:
            arr = gaussian_filter1d(arr, s, axis=axis, ctx=ctx)
    return arr


class _LinearInterp:
    """The callable gf.interp1d returns (GOOFER.py:173-239).  Inside [x[0], x[-1]] it is np.interp; outside either the two
    end segments continued (their slopes carry the reference's +1e-10 in the denominator) or a constant.  A single point
    is a constant function (with a fill value: the fill everywhere but at that point)."""

    def __init__(self, x, y, fill_value):
        self.x, self.y = x, y
        self.extrapolate = fill_value == "extrapolate"
        self.fill_value = fill_value
        if len(x) > 1:
            self.edge_slopes = ((y[1] - y[0]) / (x[1] - x[0] + 1e-10), (y[-1] - y[-2]) / (x[-1] - x[-2] + 1e-10))

    def _fill(self):
        try:
            return float(self.fill_value)
        except (TypeError, ValueError):
            raise ValueError("fill_value must be 'extrapolate' or a number")

    def __call__(self, q):
        q = np.asarray(q)
        x, y = self.x, self.y
        if len(x) == 1:
            if self.extrapolate:
                return np.full_like(q, y[0], dtype=y.dtype)
            out = np.full_like(q, self._fill())
            out[np.isclose(q, x[0])] = y[0]
            return out
        below, above = q < x[0], q > x[-1]
        if self.extrapolate:
            out = np.interp(q, x, y)
            for side, x_end, y_end, slope in ((below, x[0], y[0], self.edge_slopes[0]), (above, x[-1], y[-1], self.edge_slopes[1])):
                if np.any(side):
                    out[side] = y_end + slope * (q[side] - x_end)
            return out
        fill = self._fill()
        inside = ~(below | above)
        out = np.empty_like(q)
        if np.any(inside):
            out[inside] = np.interp(q[inside], x, y)
        out[~inside] = fill
        return out


def interp1d(x, y, kind="linear", fill_value="extrapolate"):
    """gf.interp1d: a host-side convenience like the reference's (the hot path interpolates in its kernels)."""
    if kind != "linear":
        raise ValueError("Only 'linear' interpolation is supported.")
    x, y = np.asarray(x), np.asarray(y)
    if len(x) == 0:
        raise ValueError("x cannot be empty")
    return _LinearInterp(x, y, fill_value)


def stretch_feature(feature, stretch, kind="linear", ctx=None):
    """gf.stretch_feature (GOOFER.py:597-616) on the device (fp32 rows in, fp64 math, fp32 out like the synth uses it)."""
    feature = np.asarray(feature)
    if stretch == 1.0:
        return feature.copy()
    if kind != "linear":
        raise ValueError("Only 'linear' interpolation is supported.")
    c = ctx or default_context()
    n_new = int(feature.shape[-1] * stretch)
    if feature.ndim == 1:
        if len(feature) == 0:
            raise ValueError("x cannot be empty")
        return c.stretch_rows(c.tensor(feature.astype(np.float32)), n_new).cpu().numpy().astype(np.float64)
    if feature.ndim == 2:
        if feature.shape[1] == 0:
            raise ValueError("x cannot be empty")
        rows = torch.empty((feature.shape[1], feature.shape[0]), dtype=torch.float32, device=c.device)
        rows.copy_(torch.as_tensor(np.ascontiguousarray(feature.T, dtype=np.float32)))
        return c.stretch_rows(rows, n_new).cpu().numpy().T.astype(np.float64)
    raise ValueError("Only 1D or 2D features are supported.")


def _vibrato_wave(length, sr, speed, seeded):
    """sin(2 pi speed t + phase) with a 0.1 s linear fade-in; the phase is drawn (legacy RNG) only for a seeded call."""
    phase = np.random.uniform(0, 2 * np.pi) if seeded else 0
    wave = np.sin(2 * np.pi * speed * (np.arange(length) / sr) + phase)
    n_fade = int(0.1 * sr)
    if n_fade < length:
        wave[:n_fade] *= np.linspace(0, 1, n_fade)
    return wave


def _smoothed_noise(length, sr, speed, ctx):
    """legacy-RNG normal draws, Gaussian-smoothed (sigma = sr / (6 speed) samples, on the device), peak-normalised."""
    noise = gaussian_filter1d(np.random.randn(length), sigma=sr / (speed * 6), ctx=ctx)
    noise /= np.max(np.abs(noise) + 1e-6)
    return noise


def create_volume_jitter(length, sr, speed=6.0, strength=0.1, seed=None, vibrato=False, ctx=None):
    """gf.create_volume_jitter (GOOFER.py:638-660): the multiplicative volume curve 1 + strength * x of the 'sr' / 'sd' flags —
    x a faded sinusoid (clipped to [0.5, 1.5]) or smoothed noise.  Draws come from the legacy global generator in the
    reference's order (seed, then phase or noise)."""
    if seed is not None:
        np.random.seed(seed)
    if vibrato:
        return np.clip(1.0 + _vibrato_wave(length, sr, speed, seed is not None) * strength, 0.5, 1.5)
    return 1.0 + _smoothed_noise(len(np.arange(length)), sr, speed, ctx) * strength     # (np.arange: the reference's length rule)


def make_mel_knots(sr, n_fft, K):
    """(bin freqs fp32, mel-spaced knot Hz fp32) — the codec's knot grid (GOOFER.py:77-82)."""
    from .synthetic import mel_knots_hz
    return np.fft.rfftfreq(n_fft, 1.0 / sr).astype(np.float32), mel_knots_hz(sr, K)


def compress_env_to_knots(env_spec, sr, n_fft, eps=1e-2, K_start=32, K_step=16, K_max=192, smooth_sigma_bins=0.5, ctx=None,
                          _rows=None):
    """Smallest mel-knot count whose 2-tap lerp reproduces the (sigma 0.5 blurred) envelope to < eps max relative
    error on <= 256 probe frames; knots sampled at the nearest bin, log, fp16.  Blur, error metric and knot
    gather run on the device; the loop over the 9 candidate K is host logic."""
    c = _ctx(sr, n_fft, n_fft // 4, ctx)
    rows = _rows if _rows is not None else c.rows_from(np.asarray(env_spec, dtype=np.float32).T)
    T, nb = rows.shape
    taps = gaussian_taps(smooth_sigma_bins) if smooth_sigma_bins > 0 and int(4.0 * smooth_sigma_bins + 0.5) > 0 else np.ones(1)
    env2 = c.gauss_bins_f64(rows, taps)
    probe = c.tensor(np.linspace(0, T - 1, min(256, T), dtype=int).astype(np.int64))
    res = sr / n_fft
    chosen = None
    for K in list(range(K_start, K_max + 1, K_step)) + [None]:
        last = K is None
        _, hz = make_mel_knots(sr, n_fft, K_max if last else K)
        at = np.clip(np.round(hz / res).astype(int), 0, nb - 1).astype(np.int32)
        d_at = c.tensor(at)
        if not last and not (c.knot_fit_error(env2, probe, d_at, hz) < eps):
            continue
        vals = c.knot_gather(env2, d_at).cpu().numpy().T
        chosen = {"mode": "knots", "knot_vals_log": np.ascontiguousarray(vals), "hz_knots": hz.astype(np.float32),
                  "n_bins": int(nb), "n_fft": int(n_fft), "sr": int(sr)}
        break
    return chosen


def envelope_features(y, sr, n_fft=1024, hop_length=256, ctx=None):
    """(env_spec fp64 [bins, T], env_knots) = |stft| + 1e-8 -> sigma-2 bin blur -> knot encode (GOOFER.py:942-946, 968)."""
    c = _ctx(sr, n_fft, hop_length, ctx)
    y = np.asarray(y, dtype=np.float32)
    n = len(y)
    T = 1 + n // hop_length
    S = c.rfft_frames(c.tensor(y), c.tensor(np.array([0, n], dtype=np.int64)), c.tensor(np.array([0, T], dtype=np.int64)), T)
    env_rows = c.gauss_bins_f64(c.mag_rows(S), gaussian_taps(2.0))
    env_spec = np.ascontiguousarray(env_rows.cpu().numpy().T)
    rows32 = c.rows(T, c.n_bins)
    rows32.copy_(env_rows.to(torch.float32))                  # to_compute(env_spec)
    return env_spec, compress_env_to_knots(None, sr, n_fft, ctx=c, _rows=rows32)


def extract_features(y, sr, n_fft=1024, hop_length=256, f0_min=75, f0_max=600, f0_merge_range=2, pitch_tracker=None, ctx=None):
    """gf.extract_features (GOOFER.py:940-969) -> (env_spec fp64 [bins, T], f0 per sample, voicing mask, formants {1..5}, knots).
    The envelope half runs on the GPU.  The f0 and formant tracks come from a tracker
    ``pitch_tracker(y, sr, hop_length, n_frames) -> (f0_track [frames'], {1..5: [n_frames]})`` — the reference computes them
    with Praat (third-party, unpinned: SURVEY §8 c, parity unpinned): ``goofer_amd.trackers`` makes the reference's own
    parselmouth calls when that package is installed, ``GOOFER_TRACKER`` names another one, and without any this raises
    ``trackers.TrackerUnavailable`` (a NotImplementedError).  ``f0_max`` is accepted and unused, as in the reference."""
    from . import trackers
    return trackers.analyse(y, sr, n_fft, hop_length, f0_min, f0_merge_range, tracker=pitch_tracker, ctx=ctx)


# -- synthesize --------------------------------------------------------------------------------------
def _fit(x, T):
    x = np.asarray(x, dtype=np.float64)
    if x.size >= T:
        return x[:T]
    return np.zeros(T) if x.size == 0 else np.pad(x, (0, T - x.size), mode="edge")


def note_params_from_kwargs(n=1, **kw):
    p = default_params(n)
    p["pitch_shift"] = kw.get("pitch_shift", 1.0)
    p["formant_shift"] = kw.get("formant_shift", 1.0)
    p["f_shift"] = [kw.get("F1_shift", 1.0), kw.get("F2_shift", 1.0), kw.get("F3_shift", 1.0), kw.get("F4_shift", 1.0)]
    p["uv_strength"] = kw.get("uv_strength", 0.75)
    p["breath_strength"] = kw.get("breath_strength", 0.1)
    p["normalize"] = kw.get("normalize", 1.0)
    p["apply_brightness"] = int(bool(kw.get("apply_brightness", True)))
    p["cut_below_f0"] = int(bool(kw.get("cut_subharm_below_f0", True)))
    if kw.get("f0_jitter"):
        p["f0_jitter"] = kw.get("f0_jitter_strength", 1.5)
    if kw.get("volume_jitter"):
        p["vol_jitter_harm"] = kw.get("volume_jitter_strength_harm", 50)
        p["vol_jitter_breath"] = kw.get("volume_jitter_strength_breath", 100)
    if kw.get("add_subharm"):
        if kw.get("subharm_f0_jitter", 0) > 0.0:
            p["subharm_f0_jitter"] = kw["subharm_f0_jitter"]
        if np.size(kw.get("subharm_semitones", -12)) > 16:
            raise NotImplementedError("at most sixteen sub-harmonic ratios per call on the device path")
        p["subharm_weight"] = kw.get("subharm_weight", 0.5)
    return p


def subharm_from_kwargs(kw):
    """The call-level half of gf.synthesize's add_subharm arguments (GOOFER.py:979-980) for Context.synth_batch."""
    if not kw.get("add_subharm"):
        return None
    return {"semitones": kw.get("subharm_semitones", -12), "vibrato": kw.get("subharm_vibrato", False),
            "rate": kw.get("subharm_vibrato_rate", 6.0), "depth": kw.get("subharm_vibrato_depth", 0.1),
            "delay": kw.get("subharm_vibrato_delay", 0.1)}


def _stretch(c, x, a, b, factor):
    """concat(x[:a], stretch_feature(x[a:b], factor), x[b:]) along axis 0 of a device array (GOOFER.py:1019-1057)."""
    n = x.shape[0]
    a, b, _ = slice(a, b).indices(n)
    b = max(a, b)
    if b - a == 0:
        raise ValueError("x cannot be empty")                   # what gf.interp1d raises for an empty stretch region
    mid = c.stretch_rows(x[a:b], int((b - a) * factor))
    if x.dim() == 1:
        return torch.cat([x[:a], mid, x[b:]])
    out = c.rows(a + mid.shape[0] + (n - b), x.shape[1])
    out[:a].copy_(x[:a])
    out[a:a + mid.shape[0]].copy_(mid)
    out[a + mid.shape[0]:].copy_(x[b:])
    return out


def _stretch64(x, a, b, factor):
    """concat(x[:a], stretch_feature(x[a:b], factor), x[b:]) of a 1-D float32 array in the type the reference holds it in
    afterwards: FLOAT64 — its interp1d is np.interp on float64 abscissae (GOOFER.py:173-239, 597-606, 1019-1053).  None where
    the result is not float64 (a one-sample region comes back in y's own type) or the reference raises (an empty one)."""
    x = np.asarray(x)
    n = x.shape[0]
    a, b, _ = slice(a, b).indices(n)
    b = max(a, b)
    seg = x[a:b]
    if seg.size < 2:
        return None
    mid = np.interp(np.linspace(0, 1, int(seg.size * factor)), np.linspace(0, 1, seg.size), seg)
    return np.concatenate([x[:a].astype(np.float64), mid, x[b:].astype(np.float64)])


def _roughness_params(params, kw):
    """roughness_on needs the stems BEFORE the peak gain (the gain is taken from the roughened sum, GOOFER.py:1195-1217): the
    batch runs with normalize = 0 (gain 1) and `_finish` applies the reference's gain."""
    if kw.get("roughness_on"):
        params = params.copy()
        params["normalize"] = 0.0
    return params


def _finish(c, out, d_mask, n, sr, kw):
    """The tail of gf.synthesize.  Without roughness the batch call has produced everything.  With it: the roughness layer on
    the device (only `reconstruct` hears it), then peak and gain like the reference."""
    if not kw.get("roughness_on"):
        return tuple(out[k].cpu().numpy() for k in ("rec", "harm", "uv", "bre"))
    k_list = list(kw.get("rough_k_list", (2, 3, 4)))
    h_list = kw.get("rough_h_list")
    if h_list is None:                                        # default partial weights: 0.45, 0.28, 0.18, then x 0.6 per further partial
        base = (0.45, 0.28, 0.18)
        h_list = [base[i] if i < 3 else base[2] * 0.6 ** (i - 2) for i in range(len(k_list))]
    k_list, h_list = k_list[:len(h_list)], list(h_list)[:len(k_list)]          # zip() of the reference
    alpha = float(kw.get("rough_alpha", 0.6))
    d_f0 = c.tensor(c.debug_fetch("f0")[:n])                  # f0_interp as the synthesis left it (scaled, stretched, jittered)
    noises = []
    for idx in range(len(k_list)):                            # make_smooth_noise re-seeds the LEGACY global generator
        np.random.seed(1337 + idx)
        noises.append(np.random.randn(n).astype(np.float32).astype(np.float64))
    sig_n = max(1.0, (float(kw.get("rough_noise_smooth_ms", 120.0)) * 0.001 * sr) / 6.0)
    sig_a = max(1.0, (float(kw.get("rough_alpha_slew_ms", 120.0)) * 0.001 * sr) / 6.0)
    nz = c.gauss_rows_f64(c.tensor(np.stack(noises)), gaussian_taps(sig_n)) if noises else None
    a_track = (d_mask.float() * np.float32(alpha)).double().reshape(1, n)      # alpha * vmask in fp32, filtered in fp64
    a_slew = c.gauss_rows_f64(a_track, gaussian_taps(sig_a)).reshape(n).float()
    rough = c.vocal_roughness(out["harm"], d_f0, d_mask, nz, k_list, h_list, float(kw.get("rough_noise_amp", 0.6)),
                              float(kw.get("rough_hp_fc", 320.0)), a_slew)
    harm, uv, bre = (out[k].cpu().numpy() for k in ("harm", "uv", "bre"))
    combined = rough.cpu().numpy() + uv + bre
    peak = float(np.max(np.abs(combined)) + 1e-12)
    gain = (1.0 / peak) ** float(np.clip(kw.get("normalize", 1.0), 0.0, 1.0))
    return combined * np.float32(gain), harm * np.float32(gain), uv * np.float32(gain), bre * np.float32(gain)


def _synthesize_stretched(c, d_env, f0, mask, F, params, sr, hop, phi, seed, kw):
    """gf.synthesize with stretch_factor != 1 (GOOFER.py:1019-1067): the warped envelope and the blurred noise envelope
    are made first, then both, f0 (already scaled by pitch_shift) and the mask are resampled along time, and the
    synth runs on the stretched features with its in-kernel warps and blur switched off."""
    factor = float(kw["stretch_factor"])
    f_shift = [kw.get("F%d_shift" % i, 1.0) for i in (1, 2, 3, 4)]
    fs = float(kw.get("formant_shift", 1.0))
    env_n = c.gauss_bins(d_env, gaussian_taps(1.75))
    env_h = d_env
    if any(v != 1.0 for v in f_shift) or fs != 1.0:
        env_h = c.warp_bins(d_env, c.tensor(F), f_shift if any(v != 1.0 for v in f_shift) else None, fs)
    f0 = (f0 * np.float32(kw.get("pitch_shift", 1.0))).astype(np.float32) if kw.get("pitch_shift", 1.0) != 1.0 else f0
    d_f0, d_mask = c.tensor(f0), c.tensor(mask)
    s0, s1 = kw.get("start_sec"), kw.get("end_sec")
    if s0 is not None and s1 is not None:
        a, b = int(s0 * sr), int(s1 * sr)
        fa, fb = int((s0 * sr) / hop), int((s1 * sr) / hop)
    else:
        a, b, fa, fb = 0, None, 0, None
    # f0_interp is a float64 array from here on in the reference: the jitter's product and the sub-harmonic phase trackers work on
    # it (their float32 versions land one event in ~10^5 a sample off, which a soak run of random keyword sets found)
    f0_64 = _stretch64(f0, a, b, factor) if (kw.get("f0_jitter") or kw.get("add_subharm")) else None
    d_f0, d_mask = _stretch(c, d_f0, a, b, factor), _stretch(c, d_mask, a, b, factor)
    env_h, env_n = _stretch(c, env_h, fa, fb, factor), _stretch(c, env_n, fa, fb, factor)
    n = int(d_f0.numel())
    d_f0_64 = c.tensor(f0_64) if f0_64 is not None and f0_64.size == n else None
    if n == 0:
        z = np.zeros(0, dtype=np.float32)
        return z, z.copy(), z.copy(), z.copy()
    params = _roughness_params(params, kw).copy()
    params["pitch_shift"], params["formant_shift"], params["f_shift"] = 1.0, 1.0, [1.0, 1.0, 1.0, 1.0]
    d_phi = None
    if phi is not None:
        d_phi = c.rows_from(np.asarray(phi, dtype=np.float32).T)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    noise_f0 = c.tensor(np.random.randn(n)) if kw.get("f0_jitter") else None
    noise_sub = c.tensor(np.random.randn(n)) if kw.get("add_subharm") and kw.get("subharm_f0_jitter", 0) > 0.0 else None
    vib = bool(kw.get("volume_jitter") and kw.get("volume_vibrato"))
    noise_vol = (c.tensor(np.random.randn(n)), c.tensor(np.random.randn(n))) if kw.get("volume_jitter") and not vib else None
    out = c.synth_batch(env_h, [env_h.shape[0]], d_f0, d_mask, [n], params, formants=None, phi=d_phi, seed=seed,
                        transition_sigma=float(kw.get("noise_transition_smoothness", 100)), want_mix=False,
                        noise_f0=noise_f0, noise_vol=noise_vol, f0_jitter_speed=float(kw.get("f0_jitter_speed", 100)),
                        vol_jitter_speed=float(kw.get("volume_jitter_speed", 150)), subharm=subharm_from_kwargs(kw),
                        volume_vibrato=vib, env_noise=env_n, noise_subharm=noise_sub, f0_64=d_f0_64)
    return _finish(c, out, d_mask, n, sr, kw)


def synthesize(env_spec, f0_interp, voicing_mask, y, sr, n_fft=1024, hop_length=256, glottal_smoothing=False,
               stretch_factor=1.0, start_sec=None, end_sec=None, apply_brightness=True, normalize=1.0, uv_strength=0.75,
               breath_strength=0.1, noise_transition_smoothness=100, pitch_shift=1.0, formant_shift=1.0, f0_jitter=False,
               f0_jitter_speed=100, f0_jitter_strength=1.5, volume_jitter=False, volume_vibrato=False, volume_jitter_speed=150,
               volume_jitter_strength_harm=50, volume_jitter_strength_breath=100, add_subharm=False, subharm_semitones=-12,
               subharm_weight=0.5, subharm_vibrato=False, cut_subharm_below_f0=True, subharm_vibrato_rate=6.0,
               subharm_vibrato_depth=0.1, subharm_f0_jitter=0, subharm_vibrato_delay=0.1, F1_shift=1.0, F2_shift=1.0,
               F3_shift=1.0, F4_shift=1.0, formants=None, roughness_on=False, rough_k_list=(2, 3, 4), rough_h_list=None,
               rough_alpha=0.6, rough_hp_fc=320.0, rough_noise_amp=0.6, rough_noise_smooth_ms=120.0, rough_alpha_slew_ms=120.0,
               *, phi=None, seed=None, ctx=None):
    """gf.synthesize for one note on the GPU -> (reconstruct, harmonic, aper_uv, aper_bre), fp32.

    The positional order and keyword set are the reference's (GOOFER.py:971-983; ``glottal_smoothing`` is accepted and
    unused there too); an unknown keyword raises TypeError like it does there.  Three keyword-ONLY additions:
    ``phi`` ``[bins, T]`` injects the aperiodic branch's random phases (parity runs); otherwise the device draws them
    from Philox keyed by ``seed`` (a fresh key per call when None, like the reference's unseeded generator); ``ctx``
    picks the device context."""
    kw = {k: v for k, v in locals().items() if k not in ("env_spec", "f0_interp", "voicing_mask", "y", "sr", "n_fft", "hop_length",
                                                         "phi", "seed", "ctx")}
    c = _ctx(sr, n_fft, hop_length, ctx)
    if isinstance(env_spec, dict) and env_spec.get("mode") == "knots":
        env_spec = decode_env_from_knots(env_spec, ctx=c)
        c.plan(sr, n_fft, hop_length)
    env = np.asarray(env_spec, dtype=np.float32)
    n = len(y)
    f0 = np.asarray(f0_interp, dtype=np.float32)
    mask = np.asarray(voicing_mask, dtype=np.float32)
    if n == 0:
        z = np.zeros(0, dtype=np.float32)
        return z, z.copy(), z.copy(), z.copy()
    T_env = env.shape[1]
    fm = formants_to_int_keys(kw.get("formants"))
    F = np.stack([_fit(fm[i], T_env) for i in (1, 2, 3, 4)], axis=1)           # [T_env, 4] fp64
    params = _roughness_params(note_params_from_kwargs(1, **kw), kw)
    d_env = c.rows_from(env.T)
    if kw.get("stretch_factor", 1.0) != 1.0:
        return _synthesize_stretched(c, d_env, f0, mask, F, params, sr, hop_length, phi, seed, kw)
    d_phi = None
    if phi is not None:
        d_phi = c.rows_from(np.asarray(phi, dtype=np.float32).T)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    # jitter flags draw from the legacy global np.random stream in the reference's order: f0, harm volume, breath volume
    noise_f0 = c.tensor(np.random.randn(n)) if kw.get("f0_jitter") else None
    noise_sub = c.tensor(np.random.randn(n)) if kw.get("add_subharm") and kw.get("subharm_f0_jitter", 0) > 0.0 else None
    vib = bool(kw.get("volume_jitter") and kw.get("volume_vibrato"))          # the sinusoid variant draws nothing
    noise_vol = (c.tensor(np.random.randn(n)), c.tensor(np.random.randn(n))) if kw.get("volume_jitter") and not vib else None
    d_mask = c.tensor(mask[:n])
    out = c.synth_batch(d_env, [T_env], c.tensor(f0[:n]), d_mask, [n], params, formants=c.tensor(F),
                        phi=d_phi, seed=seed, transition_sigma=float(kw.get("noise_transition_smoothness", 100)),
                        want_mix=False, noise_f0=noise_f0, noise_vol=noise_vol,
                        f0_jitter_speed=float(kw.get("f0_jitter_speed", 100)), vol_jitter_speed=float(kw.get("volume_jitter_speed", 150)),
                        subharm=subharm_from_kwargs(kw), volume_vibrato=vib, noise_subharm=noise_sub)
    return _finish(c, out, d_mask, n, sr, kw)
