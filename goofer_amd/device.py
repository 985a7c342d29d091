"""Device context: one libgoofer_hip handle per GPU, PyTorch-ROCm tensors as the buffer currency.

torch is plumbing here (allocation, streams, H2D/D2H); every number is produced by the HIP
kernels behind the C ABI.  Matrices are ``[frames, bins]`` row-major on the device — the transpose
of the reference's ``[bins, frames]`` — with fp32 rows padded to ``ld`` (a multiple of 4 floats).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


_ROW_ALIGN = int(__import__("os").environ.get("GOOFER_ROW_ALIGN", "4"))
if _ROW_ALIGN < 4 or _ROW_ALIGN & (_ROW_ALIGN - 1):
    raise ValueError("GOOFER_ROW_ALIGN must be a power of two >= 4 (floats per row are rounded up to it; the row kernels load 16 bytes at a time)")


def row_stride(n_bins: int) -> int:
    """floats per row of a [frames x bins] matrix (GOOFER_ROW_ALIGN: experiment knob, floats, a power of two >= 4)"""
    return (n_bins + _ROW_ALIGN - 1) & ~(_ROW_ALIGN - 1)


def spec_stride(n_bins: int) -> int:
    """complex64 slots per row of a [frames x bins] spectrum matrix: 128-byte aligned rows (csrc/common.h:spec_stride)."""
    return (n_bins + 15) & ~15


class GooferError(RuntimeError):
    pass


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """Owns a ``goofer_ctx`` for one device and the current (sr, n_fft, hop) plan."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise GooferError("no ROCm device visible: goofer_amd needs an MI355X (there is no CPU path)")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.lib.goofer_create(device, C.byref(h))
        if rc != 0:
            raise GooferError(f"goofer_create failed ({rc})")
        self.h = h
        self.geom = None

    def close(self):
        self._stem_scratch = None
        if getattr(self, "h", None):
            self.lib.goofer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise GooferError(f"libgoofer_hip error {rc}: {self.lib.goofer_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def tensor(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def plan(self, sr: int, n_fft: int, hop: int):
        g = (int(sr), int(n_fft), int(hop))
        if self.geom != g:
            self._check(self.lib.goofer_plan(self.h, *g))
            self.geom = g
            self.lf = (0.02, 1.7, 0.8)
        return self

    def pulse_model(self, Ra: float = 0.02, Rg: float = 1.7, Rk: float = 0.8):
        """gf.pulse_train_numba's Ra, Rg, Rk (GOOFER.py:474) for every pulse this handle makes from now on (the plan's tables
        are rebuilt; a new plan starts from the defaults again)."""
        lf = (float(Ra), float(Rg), float(Rk))
        if getattr(self, "lf", (0.02, 1.7, 0.8)) != lf:
            self._check(self.lib.goofer_pulse_model(self.h, *lf))
            self.lf = lf
        return self

    @property
    def n_bins(self):
        return self.geom[1] // 2 + 1

    def reserve(self, frames: int, samples: int, notes: int):
        self._check(self.lib.goofer_reserve(self.h, frames, samples, notes))

    def table(self, which: int) -> np.ndarray:
        out = np.zeros(8193, dtype=np.float32)
        n = self.lib.goofer_debug_table(self.h, which, out.ctypes.data_as(C.c_void_p), out.size)
        if n < 0:
            self._check(n)
        return out[:n]

    _DBG = {"frame_note": (0, np.int32), "row_src": (1, np.int64), "f0": (2, np.float32), "pulse": (3, np.float32),
            "S_harm": (4, np.complex64), "S_uv": (5, np.complex64), "S_breath": (6, np.complex64), "frames": (7, np.float32),
            "env_harm": (8, np.float32), "env_noise": (9, np.float32), "mask_short": (10, np.float64),
            "note_mag": (11, np.float32), "note_peak": (12, np.float32), "onset_cnt": (13, np.int32),
            "onset_idx": (14, np.int32), "frame_skip": (15, np.uint8)}

    def debug_fetch(self, name: str) -> np.ndarray:
        """Intermediate of the last synth_batch (onset_cnt / onset_idx: also of the last pulse_train) as a flat host array
        (tests / debugging only).  onset_idx: note k's onset samples start at sample_off[k] // 2 + 16 k."""
        which, dt = self._DBG[name]
        size = self.lib.goofer_debug_fetch(self.h, which, None, 0)
        if size < 0:
            raise GooferError(f"debug_fetch({name}) failed ({size})")
        out = np.empty(size // np.dtype(dt).itemsize, dtype=dt)
        self.lib.goofer_debug_fetch(self.h, which, out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def profile_begin(self, max_steps: int):
        self._check(self.lib.goofer_profile_begin(self.h, max_steps))

    def profile_stage_names(self) -> list:
        """Stage names by index for the pipeline the handle ran last (goofer_profile_stage_name_ex); '' = unused."""
        return [self.lib.goofer_profile_stage_name_ex(self.h, i).decode() for i in range(18)]

    def profile_only(self, stage=None):
        """Bracket one stage only from the next profile_begin on (option "prof_only": two event records per step instead of
        twenty); None: every stage again."""
        idx = -1 if stage is None else self.profile_stage_names().index(stage)
        self.set_option("prof_only", idx)

    def profile_end(self) -> dict:
        """{'steps': k, 'ms': {stage: summed milliseconds}} from HIP events on the launch stream."""
        ms = np.zeros(18, dtype=np.float64)
        k = self.lib.goofer_profile_end(self.h, ms.ctypes.data_as(C.c_void_p), 18)
        if k < 0:
            self._check(k)
        names = [self.lib.goofer_profile_stage_name_ex(self.h, i).decode() for i in range(18)]
        return {"steps": k, "ms": {nm: v for nm, v in zip(names, ms.tolist()) if nm}}

    def check(self):
        """Synchronise and raise what the asynchronous batch calls detected on the device (goofer_check)."""
        self._check(self.lib.goofer_check(self.h))

    def set_option(self, name: str, value: int):
        self._check(self.lib.goofer_set_option(self.h, name.encode(), int(value)))

    def pcm16(self, x, out=None):
        """The wav samples of fp32 audio on the device (goofer_pcm16: clip, * 32768, round half to even -> int16)."""
        out = torch.empty(x.numel(), dtype=torch.int16, device=x.device) if out is None else out
        self._check(self.lib.goofer_pcm16(self.h, _ptr(x), x.numel(), _ptr(out), self._stream()))
        return out

    def counter(self, name: str) -> int:
        """Cumulative device-side counter of the handle (goofer_counter): 'pulse_scanned_notes', 'pulse_fallback_notes'."""
        v = C.c_int64(0)
        self._check(self.lib.goofer_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    # -- CSR helpers ---------------------------------------------------------------------------
    def offsets(self, lengths):
        off = np.zeros(len(lengths) + 1, dtype=np.int64)
        np.cumsum(np.asarray(lengths, dtype=np.int64), out=off[1:])
        return off

    def frame_counts(self, sample_lengths, hop=None):
        hop = self.geom[2] if hop is None else int(hop)
        return (1 + np.asarray(sample_lengths, dtype=np.int64) // hop).tolist()

    # -- single-kernel entry points -------------------------------------------------------------
    def rfft_frames(self, x, sample_off, frame_off, total_frames: int, out=None):
        """x fp32 [N_total] -> complex64 [F_total, n_bins] (gf.stft per note)."""
        nb = self.n_bins
        if out is None:
            out = torch.empty((total_frames, spec_stride(nb)), dtype=torch.complex64, device=self.device)
        S, ldc = out, out.stride(0)
        self._check(self.lib.goofer_rfft_frames(self.h, _ptr(x), _ptr(sample_off), _ptr(frame_off), sample_off.numel() - 1,
                                                total_frames, _ptr(S), ldc, self._stream()))
        return S[:, :nb]

    def irfft_ola(self, S, sample_off, frame_off, total_samples: int):
        """complex64 [F_total, >=n_bins] (row stride taken from the tensor) -> fp32 [N_total] (gf.istft)."""
        assert S.dtype == torch.complex64 and S.stride(1) == 1
        y = torch.empty(total_samples, dtype=torch.float32, device=self.device)
        self._check(self.lib.goofer_irfft_ola(self.h, _ptr(S), S.stride(0), _ptr(sample_off), _ptr(frame_off),
                                              sample_off.numel() - 1, S.shape[0], total_samples, _ptr(y), self._stream()))
        return y

    def pulse_train(self, f0, sample_off):
        out = torch.empty_like(f0)
        self._check(self.lib.goofer_pulse_train(self.h, _ptr(f0), _ptr(sample_off), sample_off.numel() - 1, f0.numel(),
                                                _ptr(out), self._stream()))
        return out

    def gauss_bins(self, rows, taps: np.ndarray):
        """rows fp32 [R, ld-strided]; taps fp64 host array of odd length."""
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        out = self.rows_like(rows)
        n_bins = rows.shape[1]
        self._check(self.lib.goofer_gauss_bins(self.h, _ptr(rows), _ptr(out), rows.shape[0], n_bins, rows.stride(0),
                                               taps.ctypes.data_as(C.c_void_p), (taps.size - 1) // 2, self._stream()))
        return out

    def warp_bins(self, rows, formants=None, f_shift=None, ratio: float = 1.0):
        out = self.rows_like(rows)
        fs = None if f_shift is None else np.ascontiguousarray(f_shift, dtype=np.float64)
        self._check(self.lib.goofer_warp_bins(self.h, _ptr(rows), _ptr(out), rows.shape[0], rows.shape[1], rows.stride(0),
                                              _ptr(formants), fs.ctypes.data_as(C.c_void_p) if fs is not None else None,
                                              float(ratio), self._stream()))
        return out

    # -- analysis (GOOFER.py:942-946, 97-147) -----------------------------------------------------
    def mag_rows(self, S):
        """complex64 [R, >=n_bins] -> |S| + 1e-8 as fp32 rows."""
        R, nb = S.shape
        out = self.rows(R, nb)
        self._check(self.lib.goofer_mag_rows(self.h, _ptr(S), S.stride(0), R, nb, _ptr(out), out.stride(0), self._stream()))
        return out

    def gauss_bins_f64(self, rows, taps: np.ndarray):
        """fp32 rows -> fp64 rows (the reference's gaussian_filter1d returns float64)."""
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        R, nb = rows.shape
        ld64 = (nb + 1) & ~1
        out = torch.empty((R, ld64), dtype=torch.float64, device=self.device)[:, :nb]
        self._check(self.lib.goofer_gauss_bins_f64(self.h, _ptr(rows), rows.stride(0), _ptr(out), ld64, R, nb,
                                                   taps.ctypes.data_as(C.c_void_p), (taps.size - 1) // 2, self._stream()))
        return out

    def knot_fit_error(self, env64, probe_rows, knot_bin, hz_knots: np.ndarray) -> float:
        hz = np.ascontiguousarray(hz_knots, dtype=np.float32)
        err = C.c_double(0.0)
        self._check(self.lib.goofer_knot_fit_error(self.h, _ptr(env64), env64.stride(0), _ptr(probe_rows), probe_rows.numel(),
                                                   env64.shape[1], _ptr(knot_bin), hz.size, hz.ctypes.data_as(C.c_void_p),
                                                   C.byref(err), self._stream()))
        return float(err.value)

    def knot_gather(self, env64, knot_bin):
        R, K = env64.shape[0], knot_bin.numel()
        out = torch.empty((R, K), dtype=torch.float16, device=self.device)
        self._check(self.lib.goofer_knot_gather(self.h, _ptr(env64), env64.stride(0), R, _ptr(knot_bin), K, _ptr(out), self._stream()))
        return out

    def knot_decode(self, knots_f16, hz_knots: np.ndarray):
        """knots fp16 [rows, K] -> env fp32 [rows, n_bins] (view of an ld-strided buffer)."""
        hz = np.ascontiguousarray(hz_knots, dtype=np.float32)
        rows, K = knots_f16.shape
        nb = self.n_bins
        env = self.rows(rows, nb)
        self._check(self.lib.goofer_knot_decode(self.h, _ptr(knots_f16), K, hz.ctypes.data_as(C.c_void_p), rows, _ptr(env),
                                                nb, env.stride(0), self._stream()))
        return env

    def rows(self, n_rows: int, n_bins: int, dtype=torch.float32):
        """Uninitialised [n_rows, n_bins] view of an ld-strided device buffer."""
        ld = row_stride(n_bins)
        return torch.empty((n_rows, ld), dtype=dtype, device=self.device)[:, :n_bins]

    def rows_like(self, r):
        """Same shape AND same row stride as ``r`` (torch.empty_like would densify a strided view)."""
        return torch.empty((r.shape[0], r.stride(0)), dtype=r.dtype, device=self.device)[:, :r.shape[1]]

    def rows_from(self, a: np.ndarray):
        """Upload a host [n_rows, n_bins] fp32 matrix into an ld-strided buffer."""
        r = self.rows(a.shape[0], a.shape[1])
        r.copy_(torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)))
        return r

    def gauss_rows_f64(self, x, taps: np.ndarray):
        """gf.gaussian_filter1d along the last axis of an fp64 [rows, L] device tensor (each row its own reflect padding)."""
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        rows, L = x.shape
        off = self.tensor(np.arange(rows + 1, dtype=np.int64) * L)
        out = torch.empty_like(x)
        self._check(self.lib.goofer_gauss_rows_f64(self.h, _ptr(x), _ptr(off), rows, rows * L, taps.ctypes.data_as(C.c_void_p),
                                                   (taps.size - 1) // 2, _ptr(out), self._stream()))
        return out

    def vocal_roughness(self, y, f0, mask, noise_s, k_list, h_list, noise_amp: float, hp_fc: float, alpha_slewed, lengths=None):
        """apply_vocal_roughness (GOOFER.py:901-940) on fp32 device signals; ``noise_s`` fp64 [n_k, N] smoothed noises,
        ``alpha_slewed`` fp32 [N].  Returns the roughened signal (a new tensor)."""
        n_total = y.numel()
        lengths = [n_total] if lengths is None else list(lengths)
        off = self.tensor(self.offsets(lengths))
        k = np.ascontiguousarray(k_list, dtype=np.float64)
        h = np.ascontiguousarray(h_list, dtype=np.float64)
        out = torch.empty_like(y)
        self._check(self.lib.goofer_vocal_roughness(self.h, _ptr(y), _ptr(f0), _ptr(mask), _ptr(noise_s) if len(k) else None, len(k),
                                                    k.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), float(noise_amp),
                                                    float(hp_fc), _ptr(alpha_slewed), _ptr(off), len(lengths), n_total, _ptr(out),
                                                    self._stream()))
        out._keep = off
        return out

    def smooth_mask_ds(self, mask, lengths=None, sigma: float = 100.0, fast_interp: bool = False):
        """gf.smooth_mask_ds (GOOFER.py:556-569) of fp32 device masks (a ragged batch when ``lengths`` is given)."""
        n_total = mask.numel()
        lengths = [n_total] if lengths is None else [int(v) for v in lengths]
        # the kernels index the mask by these lengths: a mismatch would read and write out of bounds on the device
        if not (isinstance(mask, torch.Tensor) and mask.dtype == torch.float32 and mask.is_contiguous() and mask.device == self.device):
            raise ValueError("smooth_mask_ds expects a contiguous fp32 tensor on this context's device")
        if any(v < 0 for v in lengths) or sum(lengths) != n_total:
            raise ValueError(f"smooth_mask_ds: the lengths sum to {sum(lengths)}, the mask has {n_total} samples")
        off = self.tensor(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
        out = torch.empty(n_total, dtype=torch.float32, device=self.device)
        self._check(self.lib.goofer_smooth_mask_ds(self.h, _ptr(mask), _ptr(off), len(lengths), n_total, float(sigma),
                                                   int(bool(fast_interp)), _ptr(out), self._stream()))
        return out

    def stretch_rows(self, x, rows_out: int):
        """gf.stretch_feature along axis 0 (GOOFER.py:597-616): a 1-D fp32 tensor, or an ld-strided [rows, bins] view."""
        if x.dim() == 1:
            y = torch.empty(rows_out, dtype=torch.float32, device=self.device)
            self._check(self.lib.goofer_stretch_rows(self.h, _ptr(x), 1, x.numel(), _ptr(y), 1, rows_out, 1, self._stream()))
            return y
        y = self.rows(rows_out, x.shape[1])
        self._check(self.lib.goofer_stretch_rows(self.h, _ptr(x), x.stride(0), x.shape[0], _ptr(y), y.stride(0), rows_out, x.shape[1],
                                                 self._stream()))
        return y

    def onepole_cascade(self, x, f0, cutoff_factor: float, order: int = 4, btype: str = "lowpass", f0_mode: int = 0, lengths=None):
        """dynamic_butter_filter (SillySampler.py:95-174) on fp32 device signals: every note of ``lengths`` (default:
        the whole array as one note) is filtered with the same settings.  Returns a new tensor."""
        n_total = x.numel()
        lengths = [n_total] if lengths is None else list(lengths)
        off = self.offsets(lengths)
        jobs = np.zeros(len(lengths), dtype=_lib.ONEPOLE_JOB)
        jobs["src_off"] = jobs["dst_off"] = jobs["f0_off"] = off[:-1]
        jobs["n"], jobs["order"], jobs["highpass"] = lengths, order, int(btype != "lowpass")
        jobs["f0_mode"], jobs["cutoff_factor"] = f0_mode, cutoff_factor
        d_jobs = self.tensor(jobs.view(np.uint8))
        y = torch.empty_like(x)
        self._check(self.lib.goofer_onepole_cascade(self.h, _ptr(x), _ptr(y), _ptr(f0), _ptr(d_jobs), len(lengths), self._stream()))
        y._keep = d_jobs
        return y

    # -- the batch ------------------------------------------------------------------------------
    def device_offsets(self, env_lengths, sample_lengths, params: np.ndarray, put=None, hop=None):
        """Upload the CSR offsets and the per-note parameter array once (reused by every step of a resident batch)."""
        params = self._c_params(params)
        s_off = self.offsets(sample_lengths)
        f_off = self.offsets(self.frame_counts(sample_lengths, hop))
        e_off = self.offsets(env_lengths)
        put = put or self.tensor
        return {"s_off": s_off, "f_off": f_off, "e_off": e_off, "d_s": put(s_off), "d_f": put(f_off),
                "d_e": put(e_off), "d_par": put(params.view(np.uint8))}

    def _c_params(self, params: np.ndarray) -> np.ndarray:
        if params.dtype != _lib.NOTE_PARAMS or params.dtype.itemsize != _lib.NOTE_PARAMS.itemsize:
            # numpy re-packs structured dtypes on concatenate/promotion: force the C layout back
            fixed = np.zeros(params.shape, dtype=_lib.NOTE_PARAMS)
            for name in _lib.NOTE_PARAMS.names:
                fixed[name] = params[name]
            params = fixed
        return np.ascontiguousarray(params)

    def synth_batch(self, env, env_lengths, f0, mask, sample_lengths, params: np.ndarray, formants=None, phi=None,
                    seed: int = 0, transition_sigma: float = 100.0, want_rec=True, want_mix=True, offsets=None,
                    noise_f0=None, noise_vol=None, f0_jitter_speed: float = 100.0, vol_jitter_speed: float = 150.0,
                    subharm=None, volume_vibrato: bool = False, env_noise=None, mix_only: bool = False, noise_subharm=None,
                    assembly=None, f0_64=None):
        """Run goofer_synth_batch — or, given the ``assembly`` descriptor that produces this batch's f0 / mask / env,
        goofer_render_batch (assembly + synthesis as one call, the pulse chain forked as soon as f0 exists).

        ``subharm`` = dict(semitones, vibrato, rate, depth, delay) switches the sub-harmonic pulse layer on for the
        notes whose params.subharm_weight > 0 (gf.synthesize's add_subharm, GOOFER.py:1076-1097).

        env fp32 [R_total, n_bins] ld-strided device tensor; env_lengths rows per note;
        f0 / mask fp32 [N_total]; sample_lengths per note; params structured array (NOTE_PARAMS);
        formants fp64 [R_total, 4] or None; phi fp32 [F_total, n_bins] ld-strided or None.
        ``f0_64``: the same f0 as a float64 device tensor (goofer_batch.f0_64: what the reference holds behind its time stretch).
        Returns dict of device tensors harm / uv / bre / rec / mix, plus the CSR offsets.
        """
        nb = self.n_bins
        n = len(sample_lengths)
        o = offsets or self.device_offsets(env_lengths, sample_lengths, params)
        s_off, f_off, e_off = o["s_off"], o["f_off"], o["e_off"]
        d_s, d_f, d_e, d_par = o["d_s"], o["d_f"], o["d_e"], o["d_par"]
        N, F, R = int(s_off[-1]), int(f_off[-1]), int(e_off[-1])
        assert env.shape == (R, nb) and f0.numel() == N and mask.numel() == N and params.shape == (n,)
        if mix_only and want_mix:
            # the three stems are scratch of this call (they hold the stems before the peak gain afterwards): one block per handle,
            # grown as needed and re-used call after call in stream order, instead of 12 N bytes through the allocator per batch
            st = getattr(self, "_stem_scratch", None)
            Np = (N + 63) & ~63                                 # each stem starts on a 256-byte boundary (the finish pass's 16-byte accesses)
            if st is None or st.numel() < 3 * Np:
                st = self._stem_scratch = torch.empty(3 * max(Np, (int(1.25 * N) + 63) & ~63), dtype=torch.float32, device=self.device)
            out = {k: st[i * Np:i * Np + N] for i, k in enumerate(("harm", "uv", "bre"))}
        else:
            out = {k: torch.empty(N, dtype=torch.float32, device=self.device) for k in ("harm", "uv", "bre")}
        if want_rec:
            out["rec"] = torch.empty(N, dtype=torch.float32, device=self.device)
        if want_mix:
            out["mix"] = torch.empty(N, dtype=torch.float32, device=self.device)
        if phi is not None:
            assert phi.shape == (F, nb) and phi.stride(0) == env.stride(0)
        if env_noise is not None:                            # pre-blurred noise envelope rows (gf.synthesize's time stretch)
            assert env_noise.shape == env.shape and env_noise.stride(0) == env.stride(0)
        b = _lib.Batch(n_notes=n, n_bins=nb, ld=env.stride(0), mix_only=int(bool(mix_only and want_mix)), total_frames=F, total_samples=N, total_env_rows=R,
                       sample_off=d_s.data_ptr(), frame_off=d_f.data_ptr(), env_off=d_e.data_ptr(), env=env.data_ptr(),
                       formants=formants.data_ptr() if formants is not None else None, f0=f0.data_ptr(),
                       mask=mask.data_ptr(), phi=phi.data_ptr() if phi is not None else None,
                       env_noise=env_noise.data_ptr() if env_noise is not None else None, params=d_par.data_ptr(),
                       seed=seed, transition_sigma=float(transition_sigma),
                       noise_f0=noise_f0.data_ptr() if noise_f0 is not None else None,
                       noise_vol_h=noise_vol[0].data_ptr() if noise_vol is not None else None,
                       noise_vol_b=noise_vol[1].data_ptr() if noise_vol is not None else None,
                       noise_subharm=noise_subharm.data_ptr() if noise_subharm is not None else None,
                       f0_jitter_sigma=self.geom[0] / (f0_jitter_speed * 6), vol_jitter_sigma=self.geom[0] / (vol_jitter_speed * 6),
                       vol_jitter_speed=float(vol_jitter_speed), volume_vibrato=int(bool(volume_vibrato)),
                       subharm_ratio=_ratios(subharm)[0], subharm_more=(C.c_double * 15)(*_ratios(subharm)[1:16]),
                       subharm_vib_rate=float(subharm.get("rate", 6.0)) if subharm else 0.0,
                       subharm_vib_depth=float(subharm.get("depth", 0.1)) if subharm else 0.0,
                       subharm_vib_delay=float(subharm.get("delay", 0.1)) if subharm else 0.0,
                       subharm_vibrato=int(bool(subharm.get("vibrato", False))) if subharm else 0,
                       unit_pitch_shift=int(bool(np.all(params["pitch_shift"] == 1.0))),
                       no_warp=int(bool(np.all(params["f_shift"] == 1.0) and np.all(params["formant_shift"] == 1.0))),
                       harm=out["harm"].data_ptr(),
                       uv=out["uv"].data_ptr(), bre=out["bre"].data_ptr(),
                       rec=out["rec"].data_ptr() if want_rec else None, mix=out["mix"].data_ptr() if want_mix else None,
                       f0_64=f0_64.data_ptr() if f0_64 is not None else None)
        if f0_64 is not None:
            assert f0_64.dtype == torch.float64 and f0_64.numel() == N and f0_64.is_contiguous()
        if assembly is not None:
            self._check(self.lib.goofer_render_batch(self.h, C.byref(assembly), C.byref(b), self._stream()))
        else:
            self._check(self.lib.goofer_synth_batch(self.h, C.byref(b), self._stream()))
        out["_keep"] = (d_s, d_f, d_e, d_par, f0_64)   # keep device-side descriptors alive until the caller syncs
        out["sample_off"], out["frame_off"] = s_off, f_off
        return out


def _ratios(subharm):
    """2^(st/12) for up to sixteen sub-harmonic semitone offsets (a scalar or a list, like gf.add_subharms takes), 0-padded."""
    if not subharm:
        return [0.0] * 16
    st = np.atleast_1d(np.asarray(subharm["semitones"], dtype=np.float64))
    if st.size > 16:
        raise ValueError("at most sixteen sub-harmonic ratios per call")
    r = [float(2.0 ** (v / 12.0)) for v in st]
    return r + [0.0] * (16 - len(r))


_default = {}


def default_context(device: int = 0) -> Context:
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


def default_params(n: int) -> np.ndarray:
    """NOTE_PARAMS array with gf.synthesize's defaults (GOOFER.py:971-983) and a unity mix."""
    p = np.zeros(n, dtype=_lib.NOTE_PARAMS)
    p["pitch_shift"] = 1.0
    p["formant_shift"] = 1.0
    p["f_shift"] = 1.0
    p["uv_strength"] = 0.75
    p["breath_strength"] = 0.1
    p["normalize"] = 1.0
    p["apply_brightness"] = 1
    p["cut_below_f0"] = 1
    p["mix_harm"] = p["mix_breath"] = p["mix_unvoiced"] = p["volume"] = 1.0
    return p
