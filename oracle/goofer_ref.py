"""CPU oracle for the GOOFER DSP core — TEST INFRASTRUCTURE, not product code.

A numpy (+ small C, see native.c) restatement of the reference's analyse/synth path, written
from SURVEY.md §8(a) and a reading of ``/root/reference/GOOFER.py``; every function cites the
reference lines it follows.  Pinned against the golden vectors in ``tests/golden`` (which were
produced by running the reference itself, see ``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``goofer_amd``) never does.

Conventions kept from the reference: spectra and envelopes are ``[bins, frames]`` C-order,
storage dtype fp32 with fp64 intermediates exactly where numpy promotion puts them.

Known residual vs the *numba* reference (cannot be observed here, numba is absent): inside
``pulse_train`` numba types the period ``T`` as fp64; the stub-imported reference that made the
golden vectors computes the pulse *shape* in fp32 (numpy-2 scalar promotion).  This oracle follows
numba (fp64); shapes agree with the fixtures to ~1e-6, onsets exactly.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

F32 = np.float32
F16 = np.float16

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _native():
    """ctypes handle on oracle/_native.so, or None when it has not been built."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_native.so")
        if not os.path.exists(path):
            try:
                from . import build as _b
                _b.build()
            except Exception:
                _LIB = False
                return None
        lib = ctypes.CDLL(path)
        i64, f64, ptr = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
        lib.ola_f32.argtypes = [ptr, ptr, i64, i64, i64, ptr]
        lib.ola_f32_rows.argtypes = [ptr, ptr, i64, i64, i64, ptr]
        lib.pulse_train_f32.argtypes = [ptr, i64, f64, f64, f64, f64, ptr, ptr, ptr, i64]
        lib.pulse_train_f32.restype = i64
        lib.onepole_cascade.argtypes = [ptr, ptr, i64, ctypes.c_int, ctypes.c_int]
        _LIB = lib
    return _LIB or None


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------
# per-(sr, n_fft) tables                                                    GOOFER.py:12-46,585-595
# ---------------------------------------------------------------------------------------------
def sqrt_hann(n_fft: int) -> np.ndarray:
    """sqrt of the *symmetric* Hann window, fp32 (GOOFER.py:16)."""
    return np.hanning(n_fft).astype(F32) ** 0.5


def bin_freqs(sr, n_fft: int) -> np.ndarray:
    """rfft bin centre frequencies, fp32 ``[bins]`` (GOOFER.py:24)."""
    return np.fft.rfftfreq(n_fft, 1.0 / sr).astype(F32)


def boost_curve(n_fft: int) -> np.ndarray:
    """Linear 1..100 spectral boost over the bins (GOOFER.py:33)."""
    return np.linspace(1, 100, n_fft // 2 + 1, dtype=F32)


def ramp_gain(n_bins: int, sr, lo_hz: float, hi_hz: float, gain_db: float) -> np.ndarray:
    """1 below lo_hz, linear ramp to 10^(dB/20) at hi_hz, flat above (GOOFER.py:585-595)."""
    f = np.linspace(0, sr / 2, n_bins)
    g = np.ones_like(f)
    a, b = np.searchsorted(f, lo_hz), np.searchsorted(f, hi_hz)
    top = 10 ** (gain_db / 20)
    g[a:b] = 1 + np.linspace(0, 1, b - a) * (top - 1)
    g[b:] = top
    return g


def brightness_curves(sr, n_fft: int):
    """(harmonic, breath) brightness curves, fp32 ``[bins]`` (GOOFER.py:42-43)."""
    nb = n_fft // 2 + 1
    return (ramp_gain(nb, sr, 2000, 3500, 3.0).astype(F32), ramp_gain(nb, sr, 3500, 5000, 20.0).astype(F32))


# ---------------------------------------------------------------------------------------------
# small numeric primitives                                                  GOOFER.py:170-285
# ---------------------------------------------------------------------------------------------
def rms(x) -> float:
    return float(np.sqrt(np.mean(np.square(x)) + 1e-12))  # GOOFER.py:171


class LinInterp:
    """Linear interpolant with linear extrapolation or constant fill (GOOFER.py:173-239).

    Edge slopes carry the reference's ``+1e-10`` in the denominator.  ``y`` keeps its dtype for
    the edge-slope subtraction (an fp32 ``y`` subtracts in fp32 — numpy scalar rules)."""

    def __init__(self, x, y, fill="extrapolate"):
        self.x, self.y, self.fill = np.asarray(x), np.asarray(y), fill
        if self.x.size == 0:
            raise ValueError("x cannot be empty")
        if self.x.size > 1:
            x_, y_ = self.x, self.y
            self.sl = (y_[1] - y_[0]) / (x_[1] - x_[0] + 1e-10)
            self.sr = (y_[-1] - y_[-2]) / (x_[-1] - x_[-2] + 1e-10)

    def __call__(self, q):
        q = np.asarray(q)
        x, y = self.x, self.y
        if x.size == 1:
            if self.fill == "extrapolate":
                return np.full_like(q, y[0], dtype=y.dtype)
            out = np.full_like(q, float(self.fill))
            out[np.isclose(q, x[0])] = y[0]
            return out
        if self.fill != "extrapolate":
            inside = (q >= x[0]) & (q <= x[-1])
            out = np.empty_like(q)
            if inside.any():
                out[inside] = np.interp(q[inside], x, y)
            out[~inside] = float(self.fill)
            return out
        out = np.interp(q, x, y)
        lo, hi = q < x[0], q > x[-1]
        if lo.any():
            out[lo] = y[0] + self.sl * (q[lo] - x[0])
        if hi.any():
            out[hi] = y[-1] + self.sr * (q[hi] - x[-1])
        return out


def gauss_taps(sigma: float, truncate: float = 4.0):
    """Normalised fp64 Gaussian FIR, radius int(truncate*sigma+0.5) (GOOFER.py:247-252)."""
    r = int(truncate * sigma + 0.5)
    if r <= 0:
        return None, 0
    t = np.arange(-r, r + 1)
    k = np.exp(-0.5 * (t / sigma) ** 2)
    return k / k.sum(), r


def gauss1d(a, sigma: float, axis: int = -1) -> np.ndarray:
    """Gaussian FIR along ``axis`` with numpy 'reflect' (no edge repeat) padding; fp64 (or
    complex128) result whatever the input dtype (GOOFER.py:241-261)."""
    a = np.asarray(a)
    if a.size == 0 or a.shape[axis] == 0 or sigma <= 0.0:
        return a.copy()
    k, r = gauss_taps(sigma)
    if k is None:
        return a.copy()
    m = np.moveaxis(a, axis, -1)
    pad = np.pad(m, [(0, 0)] * (m.ndim - 1) + [(r, r)], mode="reflect")
    n = m.shape[-1]
    if m.ndim == 1 and k.size > 64:
        out = np.convolve(pad, k, mode="valid")
    else:
        out = np.zeros(m.shape, dtype=np.result_type(pad.dtype, np.float64))
        for j in range(k.size):  # symmetric taps: correlation == convolution
            out += k[j] * pad[..., j:j + n]
    return np.moveaxis(out, -1, axis)


def gauss2d(a, sigma):
    """Separable: (s0 along axis 0, s1 along axis 1); zero sigma skips (GOOFER.py:263-285)."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("gauss2d expects a 2D array")
    if a.size == 0:
        return a.copy()
    s0, s1 = (max(float(s), 0.0) for s in sigma) if isinstance(sigma, (list, tuple)) else (max(float(sigma), 0.0),) * 2
    out = a
    if s0 > 0.0:
        out = gauss1d(out, s0, axis=0)
    if s1 > 0.0:
        out = gauss1d(out, s1, axis=1)
    return out


# ---------------------------------------------------------------------------------------------
# envelope knot codec                                                       GOOFER.py:74-168
# ---------------------------------------------------------------------------------------------
def mel_knots(sr, n_fft: int, K: int):
    """(bin freqs fp32, mel-spaced knot Hz fp32) (GOOFER.py:77-82)."""
    top = 2595.0 * np.log10(1.0 + (sr / 2.0) / 700.0)
    mel = np.linspace(2595.0 * np.log10(1.0), top, K, dtype=F32)
    hz = (700.0 * (10 ** (mel / 2595.0) - 1.0)).astype(F32)
    return np.fft.rfftfreq(n_fft, 1.0 / sr).astype(F32), hz


def lerp_plan(freqs, hz_knots):
    """Per-bin (left knot index, w0, w1) of the 2-tap lerp (GOOFER.py:86-90); fp32 weights."""
    K = len(hz_knots)
    idx = np.clip(np.searchsorted(hz_knots, freqs, side="right") - 1, 0, K - 2)
    x0, x1 = hz_knots[idx], hz_knots[idx + 1]
    w1 = (freqs - x0) / np.maximum(x1 - x0, 1e-12)
    return idx, (1.0 - w1).astype(F32), w1.astype(F32)


def lerp_matrix(freqs, hz_knots) -> np.ndarray:
    """Dense ``[bins, K]`` form with 2 non-zeros per row (GOOFER.py:84-95)."""
    idx, w0, w1 = lerp_plan(freqs, hz_knots)
    W = np.zeros((len(freqs), len(hz_knots)), dtype=F32)
    rows = np.arange(len(freqs))
    W[rows, idx] = w0
    W[rows, idx + 1] = w1
    return W


def decode_env_from_knots(pack) -> np.ndarray:
    """env = exp(W @ log_knots), fp32 ``[bins, T]`` (GOOFER.py:149-168)."""
    assert pack["mode"] == "knots"
    vals = np.asarray(pack["knot_vals_log"]).astype(F32)
    hz = np.asarray(pack["hz_knots"]).astype(F32)
    n_fft, sr, n_bins = int(pack["n_fft"]), int(pack["sr"]), int(pack["n_bins"])
    W = lerp_matrix(np.fft.rfftfreq(n_fft, 1.0 / sr).astype(F32), hz)
    env = np.exp(W @ vals).astype(F32)
    return env[:n_bins] if env.shape[0] != n_bins else env


def compress_env_to_knots(env_spec, sr, n_fft, eps=1e-2, K_start=32, K_step=16, K_max=192, smooth_sigma_bins=0.5):
    """Search the smallest K in K_start..K_max whose mel-knot lerp reproduces the (sigma=0.5
    blurred) envelope to < eps max relative error on <=256 probe frames; knots are *sampled* at
    the nearest bin, stored as log in fp16 (GOOFER.py:97-147)."""
    env = np.asarray(env_spec, dtype=F32)
    if smooth_sigma_bins > 0:
        env = gauss1d(env, smooth_sigma_bins, axis=0)
    log_env = np.log(np.maximum(env, 1e-8)).astype(F32)
    n_bins, T = log_env.shape
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr).astype(F32)
    res = sr / n_fft
    probe = np.linspace(0, T - 1, min(256, T), dtype=int)
    env_probe = env[:, probe]
    chosen = None
    for K in list(range(K_start, K_max + 1, K_step)) + [None]:
        last = K is None
        _, hz = mel_knots(sr, n_fft, K_max if last else K)
        at = np.clip(np.round(hz / res).astype(int), 0, n_bins - 1)
        vals = log_env[at, :]
        if not last:
            rec = lerp_matrix(freqs, hz) @ vals[:, probe]
            err = np.max(np.abs(np.exp(rec) - env_probe) / (env_probe + 1e-8))
            if not err < eps:
                continue
        chosen = {"mode": "knots", "knot_vals_log": vals.astype(F16), "hz_knots": hz.astype(F32),
                  "n_bins": int(n_bins), "n_fft": int(n_fft), "sr": int(sr)}
        break
    return chosen


# ---------------------------------------------------------------------------------------------
# .goofy feature files                                                      GOOFER.py:48-70,287-339
# ---------------------------------------------------------------------------------------------
def formants_int_keys(d) -> dict:
    """Keep F1..F4 under int keys ('F2' -> 2); missing tracks become zeros(1) (GOOFER.py:48-62)."""
    out = {}
    if isinstance(d, dict):
        for k, v in d.items():
            if isinstance(k, str) and k.upper().startswith("F"):
                try:
                    k = int(k[1:])
                except Exception:
                    continue
            if isinstance(k, int) and 1 <= k <= 4:
                out[k] = np.asarray(v)
    for i in (1, 2, 3, 4):
        out.setdefault(i, np.zeros(1, dtype=np.float64))
    return out


def fit_length(x, T: int) -> np.ndarray:
    """fp64 copy, edge-padded or truncated to T; empty -> zeros (GOOFER.py:64-70)."""
    x = np.asarray(x, dtype=np.float64)
    if x.size >= T:
        return x[:T]
    return np.zeros(T) if x.size == 0 else np.pad(x, (0, T - x.size), mode="edge")


def save_features(path, features, f0, vmask, formants, sr, y_len):
    """npz_compressed written through an open handle so the name keeps its .goofy suffix."""
    common = dict(f0_interp=np.asarray(f0).astype(F16), voicing_mask=np.asarray(vmask).astype(F16),
                  formants=formants_int_keys(formants), sr=np.array([sr], dtype=np.int32),
                  y_len=np.array([y_len], dtype=np.int64))
    with open(path, "wb") as fh:
        if isinstance(features, dict) and features.get("mode") == "knots":
            np.savez_compressed(fh, mode=np.array(["knots"]), knot_vals_log=features["knot_vals_log"],
                                hz_knots=features["hz_knots"],
                                n_bins=np.array([features["n_bins"]], dtype=np.int32),
                                n_fft=np.array([features["n_fft"]], dtype=np.int32),
                                env_sr=np.array([features["sr"]], dtype=np.int32), **common)
        else:
            dense = np.asarray(features, dtype=F16)
            np.savez_compressed(fh, mode=np.array(["full"]), env_spec=dense,
                                n_fft=np.array([dense.shape[0] * 2 - 2], dtype=np.int32), **common)


def load_features(path):
    z = np.load(path, allow_pickle=True)
    if str(z["mode"][0]) == "knots":
        env = {"mode": "knots", "knot_vals_log": z["knot_vals_log"], "hz_knots": z["hz_knots"],
               "n_bins": int(z["n_bins"][0]), "n_fft": int(z["n_fft"][0]), "sr": int(z["env_sr"][0])}
    else:
        env = np.asarray(z["env_spec"], dtype=F32)
    return (env, np.asarray(z["f0_interp"], dtype=F32), np.asarray(z["voicing_mask"], dtype=F32),
            formants_int_keys(z["formants"].item()), int(z["sr"][0]), int(z["y_len"][0]))


# ---------------------------------------------------------------------------------------------
# STFT / ISTFT / OLA                                                        GOOFER.py:355-413
# ---------------------------------------------------------------------------------------------
def padded_signal(x, n_fft: int) -> np.ndarray:
    """Reflect-pad n_fft/2 each side ('edge' when len < 2), then edge-extend to n_fft
    (GOOFER.py:358-362)."""
    x = np.asarray(x, dtype=F32)
    h = n_fft // 2
    xp = np.pad(x, h, mode="reflect" if len(x) >= 2 else "edge")
    if len(xp) < n_fft:
        xp = np.pad(xp, (0, n_fft - len(xp)), mode="edge")
    return xp


def stft(x, n_fft=2048, hop_length=512, window=None) -> np.ndarray:
    """complex64 ``[bins, T]`` with T = 1 + (len_padded - n_fft)//hop (GOOFER.py:355-370)."""
    if window is None:
        window = np.hanning(n_fft) ** 0.5
    xp = padded_signal(x, n_fft)
    T = max(1, 1 + (len(xp) - n_fft) // hop_length)
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = xp[idx]
    frames *= np.asarray(window)[:, None]   # in place: stays fp32 even for an fp64 window
    return np.fft.rfft(frames, axis=0)


def overlap_add(frames, window, hop: int, expected_len: int) -> np.ndarray:
    """Windowed OLA normalised by the summed squared window where it exceeds 1e-9; fp32,
    frame-major accumulation order (GOOFER.py:372-390).  ``frames`` is ``[n_fft, T]``."""
    frames = np.ascontiguousarray(frames, dtype=F32)
    window = np.ascontiguousarray(window, dtype=F32)
    n_fft, T = frames.shape
    y = np.zeros(expected_len, dtype=F32)
    lib = _native()
    if lib is not None:
        lib.ola_f32(_p(frames), _p(window), n_fft, T, hop, _p(y))
        return y
    wsum = np.zeros(expected_len, dtype=F32)
    w2 = window * window
    for i in range(T):
        s = i * hop
        y[s:s + n_fft] += frames[:, i] * window
        wsum[s:s + n_fft] += w2
    ok = wsum > 1e-9
    y[ok] /= wsum[ok]
    return y


def istft(S, hop_length=512, window=None, length=None) -> np.ndarray:
    """irfft -> OLA -> drop n_fft/2 each side -> zero-pad / truncate to ``length``
    (GOOFER.py:392-413)."""
    n_fft = (S.shape[0] - 1) * 2
    window = np.hanning(n_fft).astype(F32) ** 0.5 if window is None else np.asarray(window, dtype=F32)
    frames = np.fft.irfft(np.asarray(S, dtype=np.complex64), axis=0, n=n_fft).astype(F32)
    h = n_fft // 2
    full = n_fft + hop_length * (frames.shape[1] - 1)
    y = overlap_add(frames, window, hop_length, full)[h:full - h]
    if length is not None:
        y = np.pad(y, (0, length - y.shape[0])) if y.shape[0] < length else y[:length]
    return y


# ---------------------------------------------------------------------------------------------
# glottal source                                                            GOOFER.py:437-554
# ---------------------------------------------------------------------------------------------
def lf_pulse(T, Ra=0.01, Rg=1.47, Rk=0.34, sr=44100) -> np.ndarray:
    """One LF-style pulse of round(sr*T) (>=3) samples on an fp32 time grid, peak-normalised
    (GOOFER.py:437-471, smoothing=False path)."""
    n = int(round(sr * T))
    n = max(n, 3)
    t = np.linspace(0, T, n, endpoint=False, dtype=F32)
    Tp = Ra * T
    Tc = Tp + Rk * (T - Tp)
    p = np.zeros(n, dtype=F32)
    rise = t < Tp
    if rise.any():
        p[rise] = np.sin(np.pi * t[rise] / (2 * Tp)) ** 2
    fall = (t >= Tp) & (t < Tc)
    if fall.any():
        tau = (t[fall] - Tp) / (Tc - Tp)
        p[fall] = np.exp(-Rg * tau) * np.cos(np.pi * tau / 2)
    m = np.max(np.abs(p))
    if m > 0:
        p /= m
    return p


def pulse_shape(T0: int, T: float, Ra: float, Rg: float, Rk: float) -> np.ndarray:
    """The T0-sample shape the pulse train caches for period T (GOOFER.py:508-528), fp64 math,
    fp32 storage, normalised by its fp32 peak."""
    j = np.arange(T0, dtype=np.float64)
    ti = (j * T) / T0
    Tp = Ra * T
    Tc = Tp + Rk * (T - Tp)
    v = np.zeros(T0)
    a = ti < Tp
    b = (~a) & (ti < Tc)
    v[a] = np.sin(np.pi * ti[a] / (2.0 * Tp + 1e-12)) ** 2
    tau = (ti[b] - Tp) / (Tc - Tp + 1e-12)
    v[b] = np.exp(-Rg * tau) * np.cos(np.pi * tau / 2.0)
    buf = v.astype(F32)
    m = float(np.max(np.abs(buf))) if T0 else 0.0
    if m > 0.0:
        buf = (buf.astype(np.float64) / m).astype(F32)
    return buf


def pulse_train(f0, sr, Ra=0.02, Rg=1.7, Rk=0.8, return_onsets=False):
    """Phase-accumulator pulse placement (GOOFER.py:473-554).

    ``total_phase += f0[i]/sr`` strictly sequentially in fp64 with true division; at every integer
    crossing a T0 = clip(round_half_even(sr / last_valid_f0), 3, 8192) sample pulse is *added*
    starting at i (clipped at the end).  A 5-slot shape cache keyed by T0 (slot 0 overwritten when
    full) means a T0's shape is the one computed for the period at its first (or re-cached) use.
    """
    f0 = np.ascontiguousarray(f0, dtype=F32)
    n = f0.size
    out = np.zeros(n, dtype=F32)
    lib = _native()
    if lib is not None:
        cap = n + 16
        oi = np.zeros(cap, dtype=np.int64)
        ot = np.zeros(cap, dtype=np.int64)
        cnt = lib.pulse_train_f32(_p(f0), n, float(sr), Ra, Rg, Rk, _p(out), _p(oi), _p(ot), cap)
        cnt = min(int(cnt), cap)
        return (out, oi[:cnt].copy(), ot[:cnt].copy()) if return_onsets else out
    sr = float(sr)
    phase, nxt, last = 0.0, 1.0, 160.0
    keys, bank = [], []
    oi, ot = [], []
    for i in range(n):
        v = float(f0[i])
        if f0[i] > F32(1e-6):
            last = v
        phase += v / sr
        while phase >= nxt:
            T = 1.0 / max(last, 1e-6)
            T0 = min(max(int(round(sr * T)), 3), 8192)
            if T0 in keys:
                c = keys.index(T0)
            else:
                shape = pulse_shape(T0, T, Ra, Rg, Rk)
                if len(keys) < 5:
                    keys.append(T0)
                    bank.append(shape)
                    c = len(keys) - 1
                else:
                    keys[0], bank[0], c = T0, shape, 0
            end = min(i + T0, n)
            out[i:end] += bank[c][:end - i]
            oi.append(i)
            ot.append(T0)
            nxt += 1.0
    if return_onsets:
        return out, np.array(oi, dtype=np.int64), np.array(ot, dtype=np.int64)
    return out


# ---------------------------------------------------------------------------------------------
# mask smoothing, time stretch                                              GOOFER.py:556-569,597-616
# ---------------------------------------------------------------------------------------------
def smooth_mask(mask, sigma=100, ds=4) -> np.ndarray:
    """Decimate by ds, Gaussian sigma/ds (>=1), linear upsample on fp32 linspace(0,1) grids."""
    mask = np.asarray(mask)
    short = (mask[::ds] if ds > 1 else mask).astype(F32)
    sm = gauss1d(short, max(1.0, sigma / max(1, ds)))
    if ds <= 1:
        return sm.astype(F32)
    xo = np.linspace(0.0, 1.0, num=sm.size, dtype=F32)
    xn = np.linspace(0.0, 1.0, num=mask.size, dtype=F32)
    return LinInterp(xo, sm)(xn).astype(F32)


def stretch_feature(feature, stretch):
    """Resample the last axis to int(len*stretch) points on normalised coordinates."""
    feature = np.asarray(feature)
    if stretch == 1.0:
        return feature.copy()
    n_new = int(feature.shape[-1] * stretch)
    xo = np.linspace(0, 1, feature.shape[-1])
    xn = np.linspace(0, 1, n_new)
    if feature.ndim == 1:
        return LinInterp(xo, feature)(xn)
    if feature.ndim == 2:
        return np.stack([LinInterp(xo, row)(xn) for row in feature], axis=0)
    raise ValueError("1D or 2D only")


# ---------------------------------------------------------------------------------------------
# envelope warps along the bin axis                                         GOOFER.py:618-635,805-875
# ---------------------------------------------------------------------------------------------
def shift_formants(env, ratio, sr) -> np.ndarray:
    """env(f) <- env(clip(f/ratio, 0, sr/2)), linear in bins, output in env's dtype."""
    env = np.asarray(env)
    f = np.linspace(0, sr / 2, env.shape[0])
    q = np.clip(f / ratio, 0, sr / 2)
    out = np.zeros_like(env)
    for t in range(env.shape[1]):
        out[:, t] = LinInterp(f, env[:, t])(q)
    return out


def match_env_frames(env, T: int):
    if env.shape[1] > T:
        return env[:, :T]
    if env.shape[1] < T:
        return np.pad(env, ((0, 0), (0, T - env.shape[1])), mode="edge")
    return env


def warp_env_by_formants(env, orig, shifted, sr) -> np.ndarray:
    """Piecewise-linear frequency map through anchors (0,0), (shifted_i -> orig_i for valid
    formants), (sr/2, sr/2); anchors are NOT sorted (np.interp on possibly non-monotone xp, exactly
    as the reference does).  Valid: 50 < orig < sr/2 and shifted > 50 (GOOFER.py:840-875)."""
    env = np.asarray(env)
    nyq = sr / 2.0
    f = np.linspace(0.0, nyq, env.shape[0])
    out = np.zeros_like(env)
    for t in range(env.shape[1]):
        src, dst = [0.0], [0.0]
        for i in range(4):
            fo, fs = orig[i, t], shifted[i, t]
            if fo > 50.0 and fo < nyq and fs > 50.0:
                src.append(fo)
                dst.append(fs)
        src.append(nyq)
        dst.append(nyq)
        wf = LinInterp(np.array(dst), np.array(src))(f)
        out[:, t] = LinInterp(f, env[:, t])(wf)
    return out


# ---------------------------------------------------------------------------------------------
# jitter / sub-harmonic oscillators (legacy global RNG)                     GOOFER.py:638-766
# ---------------------------------------------------------------------------------------------
def smooth_unit_noise(n: int, sr, speed: float, noise=None) -> np.ndarray:
    """randn -> Gaussian sigma = sr/(6*speed) -> divide by max(|x| + 1e-6).  ``noise`` injects the
    draw for parity; None takes it from the legacy global RNG like the reference."""
    z = np.random.randn(n) if noise is None else np.asarray(noise, dtype=np.float64)
    z = gauss1d(z, sr / (speed * 6))
    return z / np.max(np.abs(z) + 1e-6)


def f0_jitter_curve(n, sr, speed=40.0, strength=0.04, noise=None):
    return 1.0 + smooth_unit_noise(n, sr, speed, noise) * strength          # GOOFER.py:662-670


def volume_jitter_curve(n, sr, speed=6.0, strength=0.1, vibrato=False, noise=None):
    """GOOFER.py:638-660 with seed=None: vibrato is a zero-phase sinusoid with a 0.1 s fade-in,
    clipped to [0.5, 1.5]; otherwise smoothed noise."""
    if vibrato:
        t = np.arange(n) / sr
        z = np.sin(2 * np.pi * speed * t + 0)
        fade = int(0.1 * sr)
        if fade < n:
            z[:fade] *= np.linspace(0, 1, fade)
        return np.clip(1.0 + z * strength, 0.5, 1.5)
    return 1.0 + smooth_unit_noise(n, sr, speed, noise) * strength


def subharm_vibrato(f0, sr, rate=6.0, depth=0.1, delay=0.1):
    """Multiply voiced f0 by 1 + depth*sin(2 pi rate t) with a linear fade-in (GOOFER.py:748-766)."""
    f0 = np.asarray(f0)
    v = np.sin(2 * np.pi * rate * (np.arange(len(f0)) / sr) + 0)
    k = int(delay * sr)
    if k < len(v):
        v[:k] *= np.linspace(0, 1, k)
    out = f0.copy()
    on = f0 > 0
    out[on] = out[on] * (1 + v[on] * depth)
    return out


def subharm_layer(f0, sr, weight=0.5, semitones=-12, vmask=None):
    """Extra pulse layer at f0*2^(st/12): one phase tracker that wraps by subtracting 1; LF pulses
    (Rk=1) cached by a 2-decimal Hz key; masked, max-normalised, weighted (GOOFER.py:672-736)."""
    f0 = np.asarray(f0, dtype=np.float64)
    vm = (f0 > 0).astype(np.float64) if vmask is None else np.asarray(vmask, dtype=np.float64)
    semis = np.atleast_1d(np.asarray(semitones, dtype=np.float64))
    ratios = 2.0 ** (semis / 12.0)
    track = np.zeros(len(ratios))
    last = 160.0
    events = []
    for i in range(len(f0)):
        if vm[i] <= 0 or f0[i] <= 0:
            continue
        last = f0[i]
        for j, r in enumerate(ratios):
            sub = last * r
            if sub < 1e-2:
                continue
            track[j] += sub / sr
            if track[j] >= 1.0:
                events.append((i, sub, r))
                track[j] -= 1.0
    out = np.zeros_like(f0)
    cache = {}
    for i, sub, r in events:
        key = f"{sub:.2f}_sub{r:.3f}"
        if key not in cache:
            cache[key] = lf_pulse(1.0 / sub, Ra=0.02, Rg=1.7, Rk=1, sr=sr).astype(np.float64)
        p = cache[key]
        e = min(len(out), i + len(p))
        out[i:e] += p[:e - i]
    out *= vm
    m = np.max(np.abs(out)) if len(out) else 0.0
    if m > 1e-6:
        out /= m
    return out * weight


# ---------------------------------------------------------------------------------------------
# drivers                                                                   GOOFER.py:940-1220
# ---------------------------------------------------------------------------------------------
def envelope_of(y, sr, n_fft=1024, hop_length=256):
    """Analysis half that does not need Praat: |STFT| + 1e-8 -> sigma=2 bin blur -> knot encode
    (GOOFER.py:942-946, 968)."""
    S = stft(y, n_fft=n_fft, hop_length=hop_length, window=sqrt_hann(n_fft))
    env = gauss1d(np.abs(S) + 1e-8, 2.0, axis=0)
    return env, compress_env_to_knots(env, sr=sr, n_fft=n_fft)


def frame_picks(x, hop: int, T: int) -> np.ndarray:
    """x[::hop] edge-padded / truncated to T (GOOFER.py:1104-1106, 1132-1136)."""
    v = x[::hop]
    if v.size < T:
        v = np.pad(v, (0, T - v.size), mode="edge")
    return v[:T]


def highpass_mask(freqs, f0_frames) -> np.ndarray:
    """sigma((f - f0_frame)/5) with the argument clipped to +-60, fp32 (GOOFER.py:1110-1111)."""
    z = np.clip((freqs.reshape(-1, 1) - f0_frames.reshape(1, -1)) / 5, -60, 60)
    return 1.0 / (1.0 + np.exp(-z))


def _voiced_brighten(S, voiced_frames, curve):
    """On frames whose picked mask > 0: multiply by the brightness curve then 5-tap (sigma 0.5)
    blur along bins; written back into the complex64 spectrum (GOOFER.py:1138-1144)."""
    cols = np.nonzero(voiced_frames > 0)[0]
    if cols.size:
        blk = S[:, cols] * curve[:, None]
        S[:, cols] = gauss2d(blk, (0.5, 0))
    return S


def smooth_noise(length, sr, smooth_ms, seed):
    """make_smooth_noise (GOOFER.py:893-899): the LEGACY global generator is re-seeded, the fp32 draw is convolved with an fp64
    Gaussian (fp64 result)."""
    np.random.seed(seed)
    n = np.random.randn(length).astype(F32)
    return gauss1d(n, max(1.0, (smooth_ms * 0.001 * sr) / 6.0))


def one_pole_highpass(x, sr, fc):
    """GOOFER.py:878-892: python-float (fp64) state, fp32 stores."""
    if fc <= 0:
        return np.zeros_like(x)
    rc = 1.0 / (2.0 * np.pi * fc)
    a = rc / (rc + 1.0 / sr)
    y = np.zeros_like(x, dtype=F32)
    px = py = 0.0
    for i in range(len(x)):
        xn = float(x[i])
        yn = a * (py + xn - px)
        y[i] = yn
        px, py = xn, yn
    return y


def vocal_roughness(y, f0, vmask, sr, k_list=(2, 3, 4), h_list=None, alpha=0.6, hp_fc=300.0, noise_amp=0.6,
                    noise_smooth_ms=120.0, alpha_slew_ms=120.0):
    """apply_vocal_roughness (GOOFER.py:901-940): sub-harmonic amplitude modulation at f0/k with noisy rates, the
    difference high-passed and faded in by the slewed voicing mask."""
    y = np.asarray(y, dtype=F32)
    f0 = np.asarray(f0, dtype=F32)
    vmask = np.asarray(vmask, dtype=F32)
    N = len(y)
    if h_list is None:
        h_list = [0.45, 0.28, 0.18][:len(k_list)]
        if len(h_list) < len(k_list):
            extra = len(k_list) - len(h_list)
            h_list += [h_list[-1] * 0.6 ** i for i in range(1, extra + 1)]
    mod_sum = np.zeros(N, dtype=F32)
    for idx, (k, hk) in enumerate(zip(k_list, h_list)):
        nz = smooth_noise(N, sr, noise_smooth_ms, seed=1337 + idx)
        f_mod = (f0 / float(k)) * (1.0 + noise_amp * nz)
        f_mod = np.maximum(f_mod, 0.0) * vmask
        phase = 2.0 * np.pi * np.cumsum(f_mod) / float(sr)
        mod_sum += hk * np.cos(phase).astype(F32)
    y_mod = y * (1.0 + mod_sum)
    y_sub = y_mod - y
    y_sub_hp = one_pole_highpass(y_sub, sr, hp_fc)
    alpha_track = alpha * vmask
    sigma = max(1.0, (alpha_slew_ms * 0.001 * sr) / 6.0)
    alpha_slewed = gauss1d(alpha_track, sigma).astype(F32)
    return y + alpha_slewed * y_sub_hp


def synthesize(env_spec, f0_interp, voicing_mask, y, sr, n_fft=1024, hop_length=256,
               stretch_factor=1.0, start_sec=None, end_sec=None, apply_brightness=True, normalize=1.0,
               uv_strength=0.75, breath_strength=0.1, noise_transition_smoothness=100,
               pitch_shift=1.0, formant_shift=1.0,
               f0_jitter=False, f0_jitter_speed=100, f0_jitter_strength=1.5,
               volume_jitter=False, volume_vibrato=False, volume_jitter_speed=150,
               volume_jitter_strength_harm=50, volume_jitter_strength_breath=100,
               add_subharm=False, subharm_semitones=-12, subharm_weight=0.5, subharm_vibrato=False,
               cut_subharm_below_f0=True, subharm_vibrato_rate=6.0, subharm_vibrato_depth=0.1,
               subharm_f0_jitter=0, subharm_vibrato_delay=0.1,
               F1_shift=1.0, F2_shift=1.0, F3_shift=1.0, F4_shift=1.0, formants=None,
               roughness_on=False, rough_k_list=(2, 3, 4), rough_h_list=None, rough_alpha=0.6, rough_hp_fc=320.0,
               rough_noise_amp=0.6, rough_noise_smooth_ms=120.0, rough_alpha_slew_ms=120.0,
               phi=None, rng=None, noise=None, return_parts=False, **_ignored):
    """Source-filter resynthesis (GOOFER.py:971-1220).

    ``phi`` ``[bins, T]`` pins the aperiodic branch's random phases (else ``rng`` or a fresh
    default_rng draws them, like the reference).  ``noise`` optionally injects the legacy-RNG draws
    as a dict: 'f0', 'sub', 'vol_harm', 'vol_breath'.  Roughness (dead from the CLI) is not
    restated.  Returns (reconstruct, harmonic, aper_uv, aper_bre), each fp32 ``[len(y)]``.
    """
    noise = noise or {}
    win = sqrt_hann(n_fft)
    if isinstance(env_spec, dict) and env_spec.get("mode") == "knots":
        env_spec = decode_env_from_knots(env_spec)
    env = np.asarray(env_spec, dtype=F32)
    f0 = np.array(f0_interp, dtype=F32)        # private copy (the reference aliases fp32 input)
    vm = np.asarray(voicing_mask, dtype=F32)
    n_out = len(y)

    env_noise = gauss1d(env, 1.75, axis=0)     # from the UN-warped envelope  (:993)
    f0 *= pitch_shift
    T_env = env.shape[1]
    fm = formants_int_keys(formants)
    F = np.stack([fit_length(fm[i], T_env) for i in (1, 2, 3, 4)], axis=0)

    ratios = [F1_shift, F2_shift, F3_shift, F4_shift]
    if any(r != 1.0 for r in ratios):
        env = warp_env_by_formants(env, F, F * np.asarray(ratios, dtype=np.float64)[:, None], sr)
    if formant_shift != 1.0:
        env = shift_formants(env, formant_shift, sr)

    if stretch_factor != 1.0:                  # :1019-1067 (not reachable from the sampler)
        if start_sec is not None and end_sec is not None:
            a, b = int(start_sec * sr), int(end_sec * sr)
            fa, fb = int((start_sec * sr) / hop_length), int((end_sec * sr) / hop_length)
            f0 = np.concatenate([f0[:a], stretch_feature(f0[a:b], stretch_factor), f0[b:]])
            vm = np.concatenate([vm[:a], stretch_feature(vm[a:b], stretch_factor), vm[b:]])
            env = np.concatenate([env[:, :fa], stretch_feature(env[:, fa:fb], stretch_factor), env[:, fb:]], axis=1)
            env_noise = np.concatenate([env_noise[:, :fa], stretch_feature(env_noise[:, fa:fb], stretch_factor),
                                        env_noise[:, fb:]], axis=1)
        else:
            f0 = stretch_feature(f0, stretch_factor)
            env = stretch_feature(env, stretch_factor)
            vm = stretch_feature(vm, stretch_factor)
            env_noise = stretch_feature(env_noise, stretch_factor)
        n_out = len(f0)

    if f0_jitter:
        jit = f0_jitter_curve(len(f0), sr, f0_jitter_speed, f0_jitter_strength, noise.get("f0"))
        f0 *= 1.0 + ((jit - 1.0) * vm)

    pulse = pulse_train(f0.astype(F32), sr, Ra=0.02, Rg=1.7, Rk=0.8).astype(F32)

    if add_subharm:
        fs = f0
        if subharm_f0_jitter > 0.0:
            sj = f0_jitter_curve(len(fs), sr, f0_jitter_speed, subharm_f0_jitter, noise.get("sub"))
            fs *= 1.0 + ((sj - 1.0) * vm)
        if subharm_vibrato:
            fs = subharm_vibrato_(fs, sr, subharm_vibrato_rate, subharm_vibrato_depth, subharm_vibrato_delay)
        pulse += subharm_layer(fs, sr, weight=subharm_weight, semitones=subharm_semitones, vmask=vm)

    S = stft(pulse, n_fft=n_fft, hop_length=hop_length, window=win)
    T = S.shape[1]
    freqs = bin_freqs(sr, n_fft)
    hp = highpass_mask(freqs, frame_picks(f0, hop_length, T))
    if cut_subharm_below_f0:
        S *= hp
    env = match_env_frames(env, T)
    mag = np.max(np.abs(S) + 1e-8)
    bright_h, bright_b = brightness_curves(sr, n_fft)
    S = (S / mag) * env
    S *= boost_curve(n_fft)[:, None]
    voiced = frame_picks(vm, hop_length, T)
    if apply_brightness:
        S = _voiced_brighten(S, voiced, bright_h)
    harmonic = istft(S, hop_length=hop_length, window=win, length=n_out)

    env_n = match_env_frames(env_noise, T).astype(F32)
    if phi is None:
        rng = rng or np.random.default_rng()
        phi = rng.uniform(0.0, 2.0 * np.pi, size=env_n.shape).astype(F32)
    U = np.cos(phi) + 1j * np.sin(phi)
    S_uv = U * env_n
    S_br = (U * env_n) * hp
    if apply_brightness:
        S_br = _voiced_brighten(S_br, voiced, bright_b)
    aper_b = istft(S_br, hop_length=hop_length, window=win, length=n_out)
    aper_u = istft(S_uv, hop_length=hop_length, window=win, length=n_out)

    ms = smooth_mask(vm, sigma=noise_transition_smoothness, ds=4)
    bre = aper_b * ms * breath_strength
    uv = aper_u * (1.0 - ms) * uv_strength

    if volume_jitter:
        jh = volume_jitter_curve(len(harmonic), sr, volume_jitter_speed, volume_jitter_strength_harm,
                                 volume_vibrato, noise.get("vol_harm"))
        jb = volume_jitter_curve(len(bre), sr, volume_jitter_speed, volume_jitter_strength_breath,
                                 volume_vibrato, noise.get("vol_breath"))
        vj = gauss1d(vm, 20)
        harmonic *= 1.0 + (jh - 1.0) * vj
        bre *= 1.0 + (jb - 1.0) * vj

    combined = harmonic + uv + bre
    if roughness_on:                                                   # GOOFER.py:1195-1206: only `combined` hears it
        combined = vocal_roughness(harmonic, f0, vm, sr, rough_k_list, rough_h_list, rough_alpha, rough_hp_fc,
                                   rough_noise_amp, rough_noise_smooth_ms, rough_alpha_slew_ms) + uv + bre
    peak = float(np.max(np.abs(combined)) + 1e-12)
    gain = (1.0 / peak) ** float(np.clip(normalize, 0.0, 1.0))
    harmonic *= gain
    uv *= gain
    bre *= gain
    rec = combined * gain
    if return_parts:
        return rec, harmonic, uv, bre, {"pulse": pulse, "mag": float(mag), "peak": peak, "mask_smooth": ms,
                                         "hp": hp, "env": env, "env_noise": env_n}
    return rec, harmonic, uv, bre


subharm_vibrato_ = subharm_vibrato
