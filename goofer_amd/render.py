"""Batch renderer: UTAU note requests -> audio, entirely on the MI355X.

``Renderer.render(jobs)`` takes any number of (source features, 13-argument request) pairs, plans them on
the host (``sampler.plan_note``: scalars, index plans, few-hundred-value tracks), uploads the plans and
runs ``goofer_render_batch``: ``goofer_assemble_batch`` (SillySampler.resample up to the synthesize call) and
``goofer_synth_batch`` (gf.synthesize + V/B/U mix) as one call.  ``GooferResampler`` keeps the reference's
construct-to-render, one-note call surface (SillySampler.py:285-413) on top of it.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass

import os
import time
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
    # SourceArena registry key of this sample: (arena epoch token << 32) | index into the arena's per-sample tables, -1: in none.
    # An instance attribute once the sample is resident; a batch then finds its samples' table rows with one C-level pass over
    # the list (operator.attrgetter) instead of half a dozen Python loops over the sources.
    _reg_key = -1
    _DERIVED = ("_reg_key", "_knot_rows", "_tracks64")

    def __setattr__(self, name, value):
        # a feature array re-assigned after the sample was registered: the arena's table row (K, T, ylen, the raw addresses of the
        # track arrays) and the cached layouts describe the OLD arrays — forget them, the next batch registers the sample afresh
        object.__setattr__(self, name, value)
        if name in ("knots", "hz_knots", "mask", "formants", "sr", "ylen", "n_fft"):
            for k in Source._DERIVED:
                self.__dict__.pop(k, None)

    def __getstate__(self):
        # copy.copy / copy.deepcopy / pickle: a copy is another object with its own arrays — it must not inherit the original's
        # arena key (whose table row holds host addresses of the ORIGINAL's track arrays) nor its cached layouts
        return {k: v for k, v in self.__dict__.items() if k not in Source._DERIVED}

    def knot_rows(self) -> np.ndarray:
        """The knot table frame-major ([T, K] contiguous, flattened), the layout the device kernels index; made once per source
        (a voicebank sample is rendered many times)."""
        kr = getattr(self, "_knot_rows", None)
        if kr is None:
            kr = np.ascontiguousarray(self.knots.T).reshape(-1)
            object.__setattr__(self, "_knot_rows", kr)
        return kr

    def tracks64(self):
        """F1..F4 as the library's host planner takes them (``sampler.source_tracks64``; None: not the plain fp64 case)."""
        t = getattr(self, "_tracks64", False)
        if t is False:
            t = S.source_tracks64(self.formants)
            object.__setattr__(self, "_tracks64", t)
        return t

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
        # the layouts the upload and the planner take, made while the sample is loaded (once per voicebank sample) instead of
        # inside the first batch that renders it: 15 us per sample, 15 ms of a 1024-sample first batch
        src.knot_rows()
        src.tracks64()
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


class Staging:
    """What a batch uploads, laid out in ONE pinned host block and shipped by one H2D copy into a device block of the same
    size: the planner's rows (written there by the library itself, goofer_host_plan_into), the note plans, tables, pitch bends,
    parameters and offsets.  Sixteen small uploads from pageable memory were a millisecond of every batch, and the planner's
    freshly allocated result arrays cost more in page faults than in planning.  Blocks are re-used from batch to batch
    (``Renderer`` keeps the free ones); ``ready`` is the event behind the last copy out of the host block."""

    def __init__(self, device, nbytes: int):
        self.nbytes = int(nbytes)
        self.host = torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True)   # (not .pin_memory(): that is a second block and a copy)
        self.dev = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        self.np = self.host.numpy()
        self.used = 0
        self.ready = None

    def reset(self):
        if self.ready is not None:                             # the previous batch's copy out of the host block
            self.ready.synchronize()
            self.ready = None
        self.used = 0

    def reserve(self, shape, dtype):
        """(host numpy view, device tensor view) of a fresh 256-byte aligned piece."""
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        off = self.used
        end = off + ((nbytes + 255) & ~255)
        if end > self.nbytes:
            raise S.StagingFull(end)
        self.used = end
        h = self.np[off:off + nbytes].view(dtype).reshape(shape)
        tdt = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8,
               "int16": torch.int16}.get(dtype.name)
        d = self.dev[off:off + nbytes]
        if tdt is not None and tdt != torch.uint8:
            d = d.view(tdt)
        if tdt is not None and len(shape) > 1:
            d = d.view(*shape)
        return h, d

    def put(self, a, dtype=None):
        """Copy ``a`` into the block; the device view of it (valid once ``ship`` has run)."""
        a = np.asarray(a) if dtype is None else np.asarray(a, dtype=dtype)
        if a.dtype.fields is not None:                         # structured records travel as bytes
            a = np.ascontiguousarray(a).view(np.uint8)
        h, d = self.reserve(a.shape, a.dtype)
        h[...] = a
        return d

    def ship(self):
        if self.used:
            self.dev[:self.used].copy_(self.host[:self.used], non_blocking=True)
        self.ready = torch.cuda.Event()
        self.ready.record()


class _StagingLease:
    """Keeps a Staging block out of the free list for as long as a prepared batch refers to it."""

    def __init__(self, stg, pool):
        self.stg, self.pool = stg, pool

    def __del__(self):
        try:
            self.pool.append(self.stg)
        except Exception:                                      # noqa: BLE001 - interpreter shutdown
            pass


class SourceArena:
    """Voicebank samples resident in HBM: the fp16 knot tables (frame-major) and the voicing masks of every Source a Renderer has
    seen, back to back in two device arrays that only grow.  A render job touches a few hundred samples thousands of times; each
    is uploaded once and the notes' plans carry element offsets into the arrays (goofer_note_plan.knot_off / src_sample_off).
    288 GB of HBM holds any voicebank: ``budget_bytes`` only bounds a process that streams unrelated sources for days — past
    it, or once the bytes of Sources their owners have dropped (the server's LRU) exceed half of what is used, the arena
    starts over in fresh arrays (batches already prepared keep the old ones alive).  Default budget: a quarter of the device
    memory that is free when the arena is made, 48 GiB at most — several handles, ranks or a two-in-flight pair share a GPU."""

    def __init__(self, ctx: Context, budget_bytes: int | None = None):
        import threading
        if budget_bytes is None:
            try:
                free, _total = torch.cuda.mem_get_info(ctx.device)
                budget_bytes = min(48 << 30, max(1 << 30, free // 4))
            except Exception:                                  # noqa: BLE001 - no device query: the fixed default
                budget_bytes = 48 << 30
        self.ctx, self.budget = ctx, int(budget_bytes)
        self.lock = threading.RLock()                          # (lookup places missing samples under it)
        self._reset()

    def _reset(self):
        # every arena state gets its own 31-bit random token: a Source that carries another arena's key (another Renderer, an
        # earlier epoch, a pickled copy from another process) never matches by accident
        self.token = int.from_bytes(os.urandom(4), "little") >> 1
        # per-sample tables, indexed by the low half of Source._reg_key (rows are never re-used inside an epoch)
        self.n_reg = 0
        cap = 1024
        self.t = {"koff": np.zeros(cap, np.int64), "soff": np.zeros(cap, np.int64), "K": np.zeros(cap, np.int64),
                  "T": np.zeros(cap, np.int64), "ylen": np.zeros(cap, np.int64), "sr": np.zeros(cap, np.int64),
                  "n_fft": np.zeros(cap, np.int64), "lerp": np.full(cap, -1, np.int64), "trk_ok": np.zeros(cap, np.bool_),
                  "trk_ptr": np.zeros((cap, 4), np.uint64), "trk_len": np.zeros((cap, 4), np.int32)}
        self.lerp_ids, self.lerp_tabs = {}, []                 # (sr, n_fft, K, hz bytes) -> index; the 2-tap plans themselves
        self.where = {}                                        # id(Source) -> (weak reference, knot_off, sample_off)
        self.knots = torch.empty(0, dtype=torch.int16, device=self.ctx.device)
        self.mask = torch.empty(0, dtype=torch.float32, device=self.ctx.device)
        self.k_used = self.m_used = 0
        self.dead = [0]                                        # bytes of dropped Sources still held (a list: the weakref callbacks add to it)

    def _grown(self, t, used, need):
        if used + need <= t.numel():
            return t
        # doubling, but never past what the budget leaves for this array (the old array lives on in prepared batches for a while)
        cap = max(used + need, self.budget // (2 * t.element_size()))
        new = torch.empty(max(min(2 * t.numel(), cap), used + need, 1 << 20), dtype=t.dtype, device=t.device)
        new[:used].copy_(t[:used])
        return new

    @staticmethod
    def _upload(stg, pieces, dst, at, threads: int = 8):
        """``pieces`` (host arrays of dst's dtype) back to back into dst[at:], through the pinned block of ``stg``: the block's two
        halves take turns — the library packs the next pieces into one half on ``threads`` threads (goofer_host_pack, outside
        the interpreter lock) while the DMA engine ships the other.  One thread filling the whole block and waiting for its copy
        moved 200 MB of voicing masks in 25 ms; the PCIe copy alone is 8."""
        lib = _lib.load()
        item = dst.element_size()
        half = (stg.nbytes // 2) // 256 * 256
        room = half // item
        if room <= 0:
            raise ValueError("staging block too small to upload through")
        stream = torch.cuda.current_stream(dst.device)
        done_ev = [None, None]
        base = stg.host.data_ptr()
        # the pieces cut into runs of at most `room` elements: (pointer, bytes) lists, one per half-block
        runs, cur, fill = [], [], 0
        keep = []
        for p in pieces:
            p = np.ascontiguousarray(p).reshape(-1)
            keep.append(p)
            ptr, left = p.ctypes.data, p.size
            while left:
                take = min(left, room - fill)
                cur.append((ptr, take * item))
                ptr += take * item
                left -= take
                fill += take
                if fill == room:
                    runs.append((cur, fill))
                    cur, fill = [], 0
        if fill:
            runs.append((cur, fill))
        for j, (run, count) in enumerate(runs):
            b = j & 1
            if done_ev[b] is not None:
                done_ev[b].synchronize()                       # the copy out of this half two runs ago
            n = len(run)
            srcs = (C.c_void_p * n)(*[r[0] for r in run])
            sizes = (C.c_int64 * n)(*[r[1] for r in run])
            got = lib.goofer_host_pack(srcs, sizes, n, C.c_void_p(base + b * half), half, threads)
            if got != count * item:
                raise RuntimeError("goofer_host_pack: %d" % got)
            dst[at:at + count].copy_(stg.host[b * half:b * half + count * item].view(dst.dtype), non_blocking=True)
            done_ev[b] = torch.cuda.Event()
            done_ev[b].record(stream)
            at += count
        stream.synchronize()                                   # the block is the batch's own from here on
        del keep

    def place(self, sources, stg=None):
        """(knot_off, sample_off) int64 arrays for ``sources``, uploading the ones not resident yet; also the two device
        arrays those offsets index (to be kept alive with the batch).  ``stg``: an (empty) Staging block to upload through."""
        with self.lock:
            fresh = [sc for sc in {id(sc): sc for sc in sources}.values() if id(sc) not in self.where]
            if fresh:
                nk, nm = sum(sc.knots.size for sc in fresh), sum(sc.ylen for sc in fresh)
                used = 2 * self.k_used + 4 * self.m_used
                if (used + 2 * nk + 4 * nm > self.budget or (self.dead[0] > (64 << 20) and 2 * self.dead[0] > used)) and self.where:
                    self._reset()
                    fresh = list({id(sc): sc for sc in sources}.values())
                    nk, nm = sum(sc.knots.size for sc in fresh), sum(sc.ylen for sc in fresh)
                self.knots = self._grown(self.knots, self.k_used, nk)
                self.mask = self._grown(self.mask, self.m_used, nm)
                if stg is not None:
                    # through the batch's pinned staging block (still empty at this point of prepare), a block's worth at a time:
                    # the copy into pinned memory is what costs (the DMA runs at PCIe speed), a pageable source costs a second
                    # copy inside the driver and np.concatenate a third
                    self._upload(stg, [sc.knot_rows().view(np.int16) for sc in fresh], self.knots, self.k_used)
                    self._upload(stg, [sc.mask[:sc.ylen] for sc in fresh], self.mask, self.m_used)
                else:
                    kc = np.concatenate([sc.knot_rows() for sc in fresh]).view(np.int16)
                    mc = np.concatenate([sc.mask[:sc.ylen] for sc in fresh]).astype(np.float32, copy=False)
                    self.knots[self.k_used:self.k_used + nk].copy_(torch.from_numpy(kc))
                    self.mask[self.m_used:self.m_used + nm].copy_(torch.from_numpy(mc))
                torch.cuda.current_stream(self.ctx.device).synchronize()   # batches on other streams (PipelinedRenderer) read these arrays
                for sc in fresh:
                    # weakly: a Source its owner has dropped (the server's LRU of 512) must not stay alive in host memory here;
                    # its entry goes with it (its device bytes stay until the arena starts over), so a recycled id cannot alias it
                    where, dead, nbytes = self.where, self.dead, 2 * sc.knots.size + 4 * sc.ylen

                    def gone(_r, key=id(sc), where=where, dead=dead, nbytes=nbytes):
                        if where.pop(key, None) is not None:
                            dead[0] += nbytes
                    ref = weakref.ref(sc, gone)
                    self.where[id(sc)] = (ref, self.k_used, self.m_used, self._register(sc, self.k_used, self.m_used))
                    self.k_used += sc.knots.size
                    self.m_used += sc.ylen
            at = [self.where[id(sc)] for sc in sources]
            return (np.array([a[1] for a in at], dtype=np.int64), np.array([a[2] for a in at], dtype=np.int64), self.knots, self.mask)

    def _register(self, sc, koff, soff):
        """A row of the per-sample tables for a sample that has just been placed (under the lock)."""
        u = self.n_reg
        if u >= self.t["K"].shape[0]:
            for k, v in self.t.items():
                grown = np.zeros((2 * v.shape[0],) + v.shape[1:], dtype=v.dtype)
                if k == "lerp":
                    grown[:] = -1
                grown[:v.shape[0]] = v
                self.t[k] = grown
        t = self.t
        t["koff"][u], t["soff"][u], t["K"][u], t["T"][u] = koff, soff, sc.knots.shape[0], sc.knots.shape[1]
        t["ylen"][u], t["sr"][u], t["n_fft"][u] = sc.ylen, sc.sr, sc.n_fft
        if sc.hz_knots is not None:                            # (dense sources: no lerp plan, -1)
            key = (sc.sr, sc.n_fft, sc.knots.shape[0], sc.hz_knots.tobytes())
            g = self.lerp_ids.get(key)
            if g is None:
                g = self.lerp_ids[key] = len(self.lerp_tabs)
                self.lerp_tabs.append(_lerp_plan(sc.sr, sc.n_fft, sc.hz_knots))
            t["lerp"][u] = g
        tr = sc.tracks64()
        t["trk_ok"][u] = tr is not None
        if tr is not None:
            t["trk_ptr"][u], t["trk_len"][u] = tr.ptrs, tr.lens
        object.__setattr__(sc, "_reg_key", (self.token << 32) | u)
        self.n_reg = u + 1
        return u

    def lookup(self, sources, stg=None):
        """Rows of the per-sample tables for ``sources`` (an int64 array, one per element) plus the tables and device arrays to
        index with them — samples not resident yet are placed (uploaded) first.  One C-level pass over the list when every
        sample is resident, which is every batch of a job after a voicebank sample's first."""
        import operator
        n = len(sources)
        get = operator.attrgetter("_reg_key")
        with self.lock:
            # The attribute is a HINT: a sample carries the key of one arena, and one that lives in several (two Renderers, one
            # PipelinedRenderer per GPU on threads) is re-stamped by whichever saw it last.  A key is trusted only when it
            # carries this epoch's token (rows are never re-used inside an epoch); every other row comes from this arena's own
            # table under its lock, never from a second read of the shared attribute.
            keys = np.fromiter(map(get, sources), dtype=np.int64, count=n)
            token = self.token
            miss = np.nonzero((keys >> 32) != token)[0] if n else ()
            if len(miss):
                ms = [sources[i] for i in miss]
                self.place(ms, stg)                            # uploads the ones that are not resident (may start a new epoch)
                if self.token != token:
                    # a new epoch: the rows behind the keys that matched are gone with the old one — place the whole batch here
                    self.place(list(sources), stg)
                    miss, ms = np.arange(n), sources
                rows = np.fromiter((self.where[id(sc)][3] for sc in ms), dtype=np.int64, count=len(ms))
                keys[miss] = (self.token << 32) | rows
                for sc, k in zip(ms, keys[miss].tolist()):
                    object.__setattr__(sc, "_reg_key", k)
            if n and not bool(((keys >> 32) == self.token).all()):
                raise RuntimeError("SourceArena.lookup: a sample's row belongs to another arena or epoch")
            return keys & 0xFFFFFFFF, dict(self.t), list(self.lerp_tabs), self.knots, self.mask


class Renderer:
    def __init__(self, ctx: Context | None = None, hop: int = S.HOP):
        self.ctx = ctx or default_context()
        self.hop = hop
        self.sources = SourceArena(self.ctx)
        self._stagings = []                                    # free Staging blocks (see prepare)
        # host threads of the library's planner (0: up to eight).  One process per GPU shares the host with the other ranks of
        # its node: under torchrun (LOCAL_WORLD_SIZE) each rank takes its share of the cores, at most eight
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
        self.plan_threads = 0 if lw <= 1 else max(1, min(8, (os.cpu_count() or 8) // lw))

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
                     "planned": prep["planned"], "stems": out}
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
        ctx, post, rb = self.ctx, prep["post"].copy(), prep["requests"]
        n = rb.n
        extra = {}
        su = [int(i) for i in np.nonzero(rb.col["subharm_gain"] > 0.0)[0]]
        sj = [int(i) for i in np.nonzero(rb.col["growl_mix"] > 0.0)[0]]
        sa = [int(i) for i in np.nonzero(rb.col["aperiodic_mix"] > 0.0)[0]]
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

    def prepare(self, jobs, phi_seeds=None, note_ids=None, trim_rows: bool = True, device_calls: bool = True):
        """Plan every note on the host and make the batch resident in HBM (plans, tables, sources).
        ``device_calls=False`` (PipelinedRenderer: this runs on a worker thread while the handle is rendering another batch):
        nothing here touches the library handle or waits for the device — the caller plans / reserves on the handle
        (``prep["geometry"]``) before it runs the batch.
        ``jobs``: a list of (Source, Request) pairs, or the pair (list of Sources, sampler.RequestBatch) — the requests of the
        batch as columns, what ``sampler.decode_request_batch`` makes of the argument strings without a Python object per note.
        ``note_ids`` key the on-device noise phases (default: the position in ``jobs``), so a note can render to the
        same bits whatever batch or rank it lands in.
        ``trim_rows``: assemble only the envelope rows synthesize can reach.  The reference's L0 loop hands over more
        envelope frames than the note has STFT frames (every cross-faded repeat is appended again, SillySampler.py:657-672)
        and ``gf.synthesize`` cuts the envelope to ``1 + n // hop`` frames (GOOFER.py:1115-1119): the rows behind that are
        never read — a fifth of the rows of a one-second note.  ``False`` keeps them (tests that compare the whole envelope).

        Host cost: the per-note decisions run in the library's host planner (a batch per call, csrc/planner.hip) and everything
        here is column arithmetic over the batch — no per-note Python except a handful of attribute reads."""
        need = 0
        while True:
            stg = None
            while self._stagings and stg is None:
                cand = self._stagings.pop()
                if cand.nbytes >= need:
                    stg = cand
            if stg is None:
                stg = Staging(self.ctx.device, max(need, 48 << 20))
            stg.reset()
            try:
                return self._prepare(stg, jobs, phi_seeds, note_ids, trim_rows, device_calls)
            except S.StagingFull as e:
                need = max(int(stg.nbytes * 3 // 2), 2 * int(e.args[0]) if e.args[0] > stg.nbytes else 0, stg.nbytes + (16 << 20))
                del stg

    def _prepare(self, stg, jobs, phi_seeds, note_ids, trim_rows, device_calls=True):
        tr = getattr(self, "trace_prepare", None)              # a list: (label, perf_counter) marks of the host phases (scripts/prepare_phases.py)
        if tr is None:
            _T = lambda label: None
        else:
            _T = lambda label: tr.append((label, time.perf_counter()))
        _T("start")
        ctx = self.ctx
        if isinstance(jobs, tuple) and len(jobs) == 2 and isinstance(jobs[1], S.RequestBatch):
            srcs, rb = list(jobs[0]), jobs[1]
            if len(srcs) != rb.n:
                raise ValueError("one Source per request")
        else:
            srcs = [j[0] for j in jobs]
            rb = S.RequestBatch.from_requests([j[1] for j in jobs])
        sr, n_fft = srcs[0].sr, srcs[0].n_fft
        n = rb.n
        c = rb.col
        # the samples' rows in the arena's tables (a voicebank sample rendered by several notes is resident once: same Source object)
        rows, T_, g_lerp_tabs, d_knots, d_mask_src = self.sources.lookup(srcs, stg)   # resident in HBM; new samples are uploaded here
        _T("uniq")
        uu, src_ix = np.unique(rows, return_inverse=True)
        u_K, u_T, u_ylen, u_koff, u_soff = T_["K"][uu], T_["T"][uu], T_["ylen"][uu], T_["koff"][uu], T_["soff"][uu]
        if (T_["sr"][uu] != sr).any() or (T_["n_fft"][uu] != n_fft).any():
            raise ValueError("one batch must share sr / n_fft")
        if device_calls:
            ctx.plan(sr, n_fft, self.hop)
        B = n_fft // 2 + 1
        ld = row_stride(B)
        bad = np.nonzero(u_T != 1 + u_ylen // self.hop)[0]     # frames of the analysis STFT (GOOFER.py:355-370)
        if bad.size:
            k = int(bad[0])
            raise ValueError(f"bad features: envelope has {int(u_T[k])} frames, y_len {int(u_ylen[k])} at hop {self.hop} implies "
                             f"{1 + int(u_ylen[k]) // self.hop}")
        _T("place")
        # 2-tap lerp plans of the batch: the arena's plans its samples use, renumbered from 0
        g = T_["lerp"][uu]
        dense = g < 0
        if dense.any() and (u_K[dense] != B).any():            # dense source: rows are the envelope, no lerp plan
            raise ValueError("dense envelope has %d bins, the plan has %d" % (int(u_K[dense][u_K[dense] != B][0]), B))
        used = np.unique(g[~dense])
        remap = np.full(len(g_lerp_tabs) + 1, -1, dtype=np.int64)
        remap[used] = np.arange(used.size)
        u_lerp = np.where(dense, -1, remap[np.maximum(g, 0)])
        lerp_tabs = [g_lerp_tabs[int(i)] for i in used]

        _T("lerp")
        # -- the notes' plans: cut points, frame taps, sample counts, formant tracks (SillySampler.py:449-833)
        pb = None
        if T_["trk_ok"][uu].all():
            tracks = srcs                                      # (what keeps the track arrays alive while the planner reads them)
            rec = S.plan_records(rb, sr, u_ylen[src_ix], u_T[src_ix], None, track_ptrs=T_["trk_ptr"][rows], track_lens=T_["trk_len"][rows],
                                 skip_unused_fst=True)
            # the rows go straight into the staging block: whatever the block has left after ~2 MiB for the small pieces
            geo_h, _ = stg.reserve(n, _lib.PLAN_GEOMETRY)
            cap = max(0, (stg.nbytes - stg.used - (2 << 20) - 600 * n - 8 * int(rb.bend.size)) // 96 - 8)
            (ti_h, ti_d), (tw_h, tw_d), (fo_h, fo_d), (fs_h, fs_d) = (stg.reserve((cap, 4), np.int32), stg.reserve((cap, 4), np.float64),
                                                                      stg.reserve((cap, 4), np.float64), stg.reserve((cap, 4), np.float32))
            try:
                pb = S.plan_native_into(rec, self.hop, trim_rows, geo_h, cap, ti_h, tw_h, fo_h, fs_h, keep=(tracks, rec),
                                        threads=self.plan_threads)
            except S.StagingFull as e:
                raise S.StagingFull(stg.nbytes + 96 * (int(e.args[0]) - cap) + (4 << 20)) from None
            if pb is not None:
                pb.device = {"tap_idx": ti_d[:pb.tap_idx.shape[0]], "tap_w": tw_d[:pb.tap_idx.shape[0]],
                             "formants": fo_d[:pb.tap_idx.shape[0]], "fst": fs_d[:pb.tap_idx.shape[0]]}
        if pb is None:                                         # odd formant dicts, or a note the reference refuses (raises here)
            pb = S.plans_to_arrays(S.plan_notes([(rb.request(i), sc.sr, sc.ylen, sc.knots.shape[1], sc.formants)
                                                 for i, sc in enumerate(srcs)], self.hop), self.hop, trim_rows)
        _T("planner")
        geo = pb.geo
        if (geo["n_out"] <= 0).any():
            raise ValueError("a note assembles to zero samples")
        env_lens = geo["n_out_rows"].astype(np.int64)
        lens = geo["n_out"].astype(np.int64)
        n_edit = (geo["row_hi"] - geo["row_lo"]).astype(np.int64)
        csum0 = lambda v: np.concatenate([[0], np.cumsum(v)])
        env_off, sample_off, edit_off = csum0(env_lens), csum0(lens), csum0(n_edit)
        t_off, o_off, e_off = int(env_off[-1]), int(sample_off[-1]), int(edit_off[-1])

        _T("geo")
        # -- request scalars as columns
        c_rev, c_be, c_es, c_fw, c_fv, c_pm, c_tempo, c_fhz = (c["reverse"], c["brightness_env"], c["env_shape"], c["formant_width"],
                                                                c["force_voiced"], c["pitch_m"], c["tempo"], c["fry_hz"])
        c_pd, c_fs, c_norm, c_hm, c_bm, c_um, c_vol = (c["pitch_dyn"], c["formant_shift"], c["normalize"], c["harmonic_mix"],
                                                        c["breathiness_mix"], c["unvoiced_mix"], c["volume"])
        c_f0j = np.where(c["f0_jitter"] != 0, c["f0_jitter_strength"], 0.0)
        c_vj = np.where(c["volume_jitter"] != 0, c["volume_jitter_strength"], 0.0)
        c_sub = np.where(c["add_subharm"] != 0, c["subharm_weight"], 0.0)
        c_su, c_sj, c_sa, c_sd, c_st = c["subharm_gain"], c["growl_mix"], c["aperiodic_mix"], c["sd_strength"], c["tension"]
        c_tc = rb.t_cents
        bend_off = rb.bend_off
        n_bend = np.diff(bend_off)

        def table_ids(col, off_value, make):
            """Index of every note's table among the tables of the distinct values of ``col`` (-1 where col == off_value)."""
            on = col != off_value
            ids = np.full(n, -1, dtype=np.int64)
            if not on.any():
                return ids, []
            vals, inv = np.unique(col[on], return_inverse=True)
            ids[on] = inv
            return ids, [make(float(v)) for v in vals]

        tilt_id, tilt_tabs = table_ids(c_be, 1.0, lambda v: _tilt(sr, B, v))
        fw_id, fw_tabs = table_ids(c_fw, 0.0, lambda v: _fw_plan(B, v))
        es_id, es_tabs = table_ids(c_es, 0.0, lambda v: S.gauss_taps((1.0 + 6.0 * abs(v)) if v < 0.0 else (0.8 + 4.0 * abs(v))))
        es_toff = csum0([t.size for t in es_tabs])
        es_rad = np.array([(t.size - 1) // 2 for t in es_tabs], dtype=np.int64)
        es_on = es_id >= 0

        _T("tables")
        P = np.zeros(n, dtype=_lib.NOTE_PLAN)
        P["knot_off"], P["K"], P["lerp_plan"], P["n_src_rows"] = u_koff[src_ix], u_K[src_ix], u_lerp[src_ix], u_T[src_ix]
        P["reverse"], P["tilt"], P["fw_plan"] = c_rev, tilt_id, fw_id
        P["es_mode"] = np.where(es_on, np.where(c_es < 0.0, 1, 2), 0)
        P["es_amount"] = np.where(es_on, 5 * np.abs(c_es), 0.0)
        P["es_taps_off"] = np.where(es_on, es_toff[np.maximum(es_id, 0)], 0) if es_tabs else 0
        P["es_radius"] = np.where(es_on, es_rad[np.maximum(es_id, 0)], 0) if es_tabs else 0
        P["row_lo"], P["n_edit"], P["edit_off"] = geo["row_lo"], n_edit, edit_off[:-1]
        P["tap_off"], P["env_off"], P["n_out_rows"], P["env_f64"] = env_off[:-1], env_off[:-1], env_lens, geo["env_f64"]
        P["fst"] = rb.formant_strength
        P["src_sample_off"], P["ylen"], P["out_sample_off"] = u_soff[src_ix], u_ylen[src_ix], sample_off[:-1]
        for k in ("n_out", "n_pre", "s_pre", "s_tail", "tail_len", "want_samples", "n_before_vel", "vel_active", "vel_factor", "pre_new",
                  "fry_dir", "fry_const_lo", "fry_const_hi", "fry_glide_lo", "fry_glide_hi", "fry_a", "fry_b", "fry_fade"):
            P[k] = geo[k]
        P["force_voiced"], P["bend_off"], P["n_bend"], P["pitch_m"] = c_fv, bend_off[:-1], n_bend, c_pm
        P["pitch_t"] = c_tc / 100.0
        P["tick_dt"] = 60.0 / (c_tempo * 96.0)
        P["fry_hz"], P["pd_on"], P["pd_base"] = c_fhz, c_pd != 0.0, c_pm + c_tc / 100.0
        # pitch curve per tick in MIDI semitones: bend / 100 + pitch_m (+ t / 100 where the flag is set)   SillySampler.py:838-846
        bend = rb.bend.astype(np.float64) / 100.0 + np.repeat(c_pm, n_bend)
        if c_tc.any():
            sel = np.repeat(c_tc != 0.0, n_bend)
            bend[sel] = bend[sel] + np.repeat(c_tc / 100.0, n_bend)[sel]

        _T("P")
        def cat_tab(tabs, k, dtype):
            return stg.put(np.concatenate([t[k] for t in tabs]), dtype) if tabs else None

        on_dev = getattr(pb, "device", None)                   # the library planner wrote the rows into the staging block
        d = dict(
            notes=stg.put(P),
            knots=d_knots, mask_src=d_mask_src,
            lerp_idx=cat_tab(lerp_tabs, 0, np.int32), lerp_w0=cat_tab(lerp_tabs, 1, np.float32), lerp_w1=cat_tab(lerp_tabs, 2, np.float32),
            tilts=stg.put(np.concatenate(tilt_tabs)) if tilt_tabs else None,
            es_taps=stg.put(np.concatenate(es_tabs)) if es_tabs else None,
            fw_lo=cat_tab(fw_tabs, 0, np.int32), fw_hi=cat_tab(fw_tabs, 1, np.int32), fw_frac=cat_tab(fw_tabs, 2, np.float64),
            tap_idx=on_dev["tap_idx"] if on_dev else stg.put(pb.tap_idx), tap_w=on_dev["tap_w"] if on_dev else stg.put(pb.tap_w),
            fst_tracks=on_dev["fst"] if on_dev else stg.put(pb.fst),
            bend=stg.put(bend),
            lease=_StagingLease(stg, self._stagings),
        )
        _T("puts")
        d_formants = on_dev["formants"] if on_dev else stg.put(pb.formants)
        env = torch.empty((t_off, ld), dtype=torch.float32, device=ctx.device)[:, :B]   # (= ctx.rows, without asking the handle for B)
        f0 = torch.empty(o_off, dtype=torch.float32, device=ctx.device)
        mask = torch.empty(o_off, dtype=torch.float32, device=ctx.device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        any_pd = bool((c_pd != 0.0).any())
        any_fry = bool((geo["fry_b"] > geo["fry_a"]).any())
        bend_out = torch.zeros(o_off, dtype=torch.float32, device=ctx.device) if any_pd else None
        a = _lib.Assembly(n_notes=n, n_bins=B, ld=ld, sr=sr, max_K=int(u_K.max()),
                          total_edit_rows=e_off, total_out_rows=t_off, total_samples=o_off,
                          notes=ptr(d["notes"]), knots=ptr(d["knots"]), lerp_idx=ptr(d["lerp_idx"]), lerp_w0=ptr(d["lerp_w0"]),
                          lerp_w1=ptr(d["lerp_w1"]), tilts=ptr(d["tilts"]), es_taps=ptr(d["es_taps"]), fw_lo=ptr(d["fw_lo"]),
                          fw_hi=ptr(d["fw_hi"]), fw_frac=ptr(d["fw_frac"]), tap_idx=ptr(d["tap_idx"]), tap_w=ptr(d["tap_w"]),
                          fst_tracks=ptr(d["fst_tracks"]), mask_src=ptr(d["mask_src"]), bend=ptr(d["bend"]), edit_rows=None,
                          env_out=env.data_ptr(), f0_out=f0.data_ptr(), mask_out=mask.data_ptr(), bend_out=ptr(bend_out),
                          any_fry=int(any_fry))
        _T("alloc+assembly")
        # per-note synthesize parameters
        par = default_params(n)
        par["formant_shift"], par["normalize"] = c_fs, c_norm
        par["f_shift"] = rb.f_shift
        par["mix_harm"], par["mix_breath"], par["mix_unvoiced"], par["volume"] = c_hm, c_bm, c_um, c_vol
        nids = np.asarray(note_ids if note_ids is not None else range(n), dtype=np.uint64)   # Philox stream of the note: its id, not its batch position
        par["seed"] = np.stack([nids & np.uint64(0xFFFFFFFF), (nids >> np.uint64(32)) & np.uint64(0xFFFFFFFF)], axis=1)
        par["f0_jitter"], par["vol_jitter_harm"], par["vol_jitter_breath"], par["subharm_weight"] = c_f0j, c_vj, c_vj * 2, c_sub
        lens_l = lens.tolist()
        env_lens_l = env_lens.tolist()
        # sh / sr draws come from the legacy global np.random stream, note by note, in the reference's order
        # (f0 jitter, harmonic volume, breath volume: GOOFER.py:666, 653)
        noise_f0 = noise_vol = None
        any_f0j, any_vj = bool((c["f0_jitter"] != 0).any()), bool((c["volume_jitter"] != 0).any())
        if any_f0j or any_vj:
            nf, nh, nb = [], [], []
            for jf, jv, n_ in zip(c["f0_jitter"] != 0, c["volume_jitter"] != 0, lens_l):
                nf.append(np.random.randn(n_) if jf else np.zeros(n_))
                nh.append(np.random.randn(n_) if jv else np.zeros(n_))
                nb.append(np.random.randn(n_) if jv else np.zeros(n_))
            if any_f0j:
                noise_f0 = ctx.tensor(np.concatenate(nf))
            if any_vj:
                noise_vol = (ctx.tensor(np.concatenate(nh)), ctx.tensor(np.concatenate(nb)))
        _T("par")
        # sample-domain post chain: per-note table (offsets into the extra synth calls are filled in by run())
        post = np.zeros(n, dtype=_lib.POST_NOTE)
        post["su_off"] = post["sj_off"] = post["sa_off"] = -1
        post["su_gain"], post["sj_mix"], post["sa_mix"], post["sd_strength"], post["tension"], post["pitch_dyn"] = c_su, c_sj, c_sa, c_sd, c_st, c_pd
        post["fry_a"], post["fry_b"], post["fry_fade"] = geo["fry_a"], geo["fry_b"], geo["fry_fade"]
        growl = {}
        for i in np.nonzero(c_sj > 0.0)[0]:                    # 'sj': f0 * 0.5 * 2^N(0, mix^2), a fresh generator per call  :1063-1065
            rng = np.random.default_rng(phi_seeds[i]) if phi_seeds is not None else np.random.default_rng()
            growl[int(i)] = 0.5 * (2.0 ** rng.normal(loc=0.0, scale=float(c_sj[i]) ** 2, size=lens_l[i]))
        f0_growl = None
        if growl:
            # the layer's f0 is the fp64 pitch curve times the factor, rounded to fp32 once (SillySampler.py:1065): the
            # assembly kernel writes it next to f0 while it still holds the fp64 value
            mul = np.ones(o_off, dtype=np.float64)
            for i, gv in growl.items():
                mul[int(sample_off[i]):int(sample_off[i + 1])] = gv
            d["f0_mul"] = ctx.tensor(mul)
            f0_growl = torch.zeros(o_off, dtype=torch.float32, device=ctx.device)
            a.f0_mul, a.f0_mul_out = d["f0_mul"].data_ptr(), f0_growl.data_ptr()
        has_post = bool(((c_su > 0) | (c_sj > 0) | (c_sa > 0) | (c_sd > 0) | (c_st != 0) | (c_pd != 0)).any()) or any_fry
        phi = None
        if phi_seeds is not None:
            mats = []
            for n_, sd in zip(lens_l, phi_seeds):
                T = 1 + n_ // self.hop
                mats.append(np.random.default_rng(sd).uniform(0.0, 2.0 * np.pi, size=(B, T)).astype(np.float32).T)
            phi = ctx.rows_from(np.concatenate(mats))
        _T("post")
        offsets = ctx.device_offsets(env_lens_l, lens_l, par, put=stg.put, hop=self.hop)
        frames = int(offsets["f_off"][-1])
        _T("offsets")
        stg.ship()                                             # one H2D copy for everything above
        if device_calls:
            ctx.reserve(frames, o_off, n)
            torch.cuda.current_stream(ctx.device).synchronize()   # (this batch's uploads; other lanes' streams are not waited for)
        _T("ship+sync")
        return {"assembly": a, "keep": d, "env": env, "f0": f0, "mask": mask, "params": par, "lens": lens_l, "env_lens": env_lens_l,
                "noise_f0": noise_f0, "noise_vol": noise_vol, "subharm": bool((c_sub > 0).any()),
                "post": post if has_post else None, "growl": growl, "f0_growl": f0_growl, "bend_out": bend_out, "requests": rb, "sources": srcs, "geometry": (sr, n_fft, self.hop, frames, o_off, n),
                "formants": d_formants, "phi": phi, "planned": pb, "offsets": offsets,
                "sample_off": sample_off, "env_off": env_off, "frames": frames, "samples": o_off, "edit_rows": e_off}


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
    """Mono PCM16 wav (soundfile's default subtype, what the reference writes).  int16 input (samples already converted on
    the device, ``Context.pcm16``) is written as it is."""
    import wave
    x = np.asarray(x)
    if x.dtype == np.int16:
        pcm = x.astype("<i2", copy=False)
    else:
        pcm = np.clip(x.astype(np.float64), -1.0, 1.0 - 1.0 / 32768)
        pcm = np.round(pcm * 32768.0).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm.tobytes())


# ---------------------------------------------------------------------------------------------
# long jobs: two batches in flight
# ---------------------------------------------------------------------------------------------
class PipelinedRenderer:
    """A stream of batches with ``depth`` of them in flight on one GPU (SillySampler.py:1196-1224 is the entry this stands
    behind: the reference renders a note per request; a render job is thousands of them).

    ``depth`` lanes, each a ``Renderer`` with its own library handle (scratch arena, side stream) and its own HIP stream; the
    voicebank arena is shared.  Host threads decode and plan batches k + 1, k + 2 (``Renderer.prepare(device_calls=False)``:
    numpy + the library's planner, both outside the interpreter lock for most of their time) while batch k runs; the finished
    mix of batch k - 1 crosses PCIe on a copy stream into a pinned buffer of its lane under step k.  Per batch the job then
    costs max(host planning / workers, device step, D2H of the mix) instead of their sum.  ``depth + 1`` batches are in flight
    (round 5): batch k + 1 is queued on the device before batch k - 1 is home, two pinned buffers per lane taking turns, so the
    device does not idle while the host waits for audio — the steady state runs at the device step (1.9-2.0 ms per 1024 notes).

    ``coalesce`` (long jobs): that many consecutive batches of the caller are planned and rendered as ONE device batch and handed
    back one by one — the host's cost per batch is mostly per-call overhead of ~150 numpy operations under the interpreter lock,
    which the worker threads queue for (a ``prepare`` of 3.7 ms alone takes 6-10 ms beside three others); twice the notes per call
    is half of that per note.  The notes keep the Philox ids of their own batch.  The audio is bit-identical to the un-coalesced
    job's WHEN the merged batches take the same kernel path: the library picks some paths per device batch (a note with the 'sg'
    layer or the volume jitter moves its whole device batch to the one-kernel-per-step pipeline, a fry note turns the fused
    warp off), and those agree with the default path to fp32 rounding, not to the bit — so a note's last bits can depend on
    the batches it was merged with (never beyond the 1e-4 parity bound).  Batches of another geometry or request type
    (``Request`` objects / argument lists) start a new device batch; ``RequestBatch`` objects are not merged.  First-audio
    latency grows with ``coalesce``.  The default 1 keeps a server's latency and per-batch path choice.

    ``render_iter`` yields ``(mix, sample_off)`` per batch, in order: ``mix`` is a float32 numpy view of the lane's pinned
    buffer, valid until ``depth`` more batches have been taken from the iterator (copy what must live longer)."""

    def __init__(self, device: int = 0, hop: int = S.HOP, depth: int = 2, workers: int = 4, staging_bytes: int = 48 << 20,
                 freeze_gc: bool = True, coalesce: int = 1):
        from concurrent.futures import ThreadPoolExecutor
        self.coalesce = max(1, int(coalesce))
        staging_bytes = int(staging_bytes) * self.coalesce
        self.device = torch.device("cuda", device)
        self.freeze_gc = bool(freeze_gc)
        self.lanes = []
        arena = None
        depth, workers = max(1, depth), max(1, workers)
        # batches prepared but not finished at any time: depth + workers waiting or being planned, depth on the device — each
        # holds a staging block of its lane.  They are made here: pinning 48 MiB of host memory takes tens of milliseconds, which
        # belongs to the start of a job and not to whichever batch first finds its lane's free list empty.
        per_lane = (2 * depth + 1 + workers + depth - 1) // depth   # (depth + 1 on the device: see _render_iter)
        for _ in range(depth):
            r = Renderer(Context(device), hop=hop)
            if arena is None:
                arena = r.sources
            r.sources = arena
            r._stagings.extend(Staging(self.device, staging_bytes) for _ in range(per_lane))
            self.lanes.append({"r": r, "stream": torch.cuda.Stream(self.device), "host": None})
        self.copy_stream = torch.cuda.Stream(self.device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="goofer-prepare")
        self.workers = max(1, workers)
        self.trace = None                                      # a list: (what, batch, start, end) of the host-side phases (scripts/pipeline_job.py --trace)

    def close(self):
        self.pool.shutdown(wait=True)
        for ln in self.lanes:
            ln["r"].ctx.close()
        self.lanes = []

    def _prepare(self, k, group, note_ids):
        if self.trace is not None:
            t0 = time.perf_counter()
            try:
                return self._prepare_batch(k, group, note_ids)
            finally:
                self.trace.append(("prepare", k, t0, time.perf_counter()))
        return self._prepare_batch(k, group, note_ids)

    def _prepare_batch(self, k, group, note_ids):
        """group: [(index of the caller's batch, (sources, requests)), ...] — one device batch.  Returns (prep, note counts)."""
        ln = self.lanes[k % len(self.lanes)]
        if len(group) == 1:
            j, (srcs, reqs) = group[0]
            if not isinstance(reqs, S.RequestBatch):
                reqs = list(reqs)
                reqs = S.RequestBatch.from_requests(reqs) if (reqs and isinstance(reqs[0], S.Request)) else S.decode_request_batch(reqs)
            counts = [len(reqs)]
            ids = note_ids(j, len(reqs)) if note_ids else None
        else:
            srcs, reqs, counts, ids = [], [], [], []
            for j, (sj, rj) in group:
                rj = list(rj)
                srcs.extend(sj)
                reqs.extend(rj)
                counts.append(len(rj))
                ids.extend(note_ids(j, len(rj)) if note_ids else range(len(rj)))   # the ids the batch has when it is rendered alone
            reqs = S.RequestBatch.from_requests(reqs) if (reqs and isinstance(reqs[0], S.Request)) else S.decode_request_batch(reqs)
        with torch.cuda.stream(ln["stream"]):
            return ln["r"].prepare((srcs, reqs), note_ids=ids, device_calls=False), counts

    def _groups(self, batches):
        """The caller's batches, ``coalesce`` of them to a device batch (RequestBatch objects and the tail of the job: as they come)."""
        group, geom = [], None
        for j, b in enumerate(batches):
            if self.coalesce == 1 or isinstance(b[1], S.RequestBatch) or not len(b[0]):
                if group:
                    yield group
                    group = []
                yield [(j, b)]
                continue
            # a device batch shares sr / n_fft and one request type: batches of another geometry or kind start a new one
            g = (b[0][0].sr, b[0][0].n_fft, isinstance(b[1][0], S.Request) if len(b[1]) else None)
            if group and g != geom:
                yield group
                group = []
            geom = g
            group.append((j, b))
            if len(group) == self.coalesce:
                yield group
                group = []
        if group:
            yield group

    def render_iter(self, batches, seed: int = 0, note_ids=None, pcm16: bool = False):
        """``batches``: an iterable of (sources, requests) — requests as a ``RequestBatch``, a list of ``Request`` or a list of
        argument lists (decoded on the worker threads).  ``note_ids(k, n)``: the Philox ids of batch k's notes (default: the
        position in the batch).  ``pcm16``: yield the wav's int16 samples (converted on the device, goofer_pcm16: what
        ``write_wav`` computes on the host) — half the bytes over PCIe, which is what bounds a long job."""
        import collections
        import sys
        it = iter(enumerate(self._groups(batches)))
        ahead = collections.deque()                           # futures of prepared batches, in order
        # the worker threads and this one hand the interpreter lock over every 0.1 ms while a job runs (the default 5 ms makes a
        # thread that is ready to launch the next step wait for a planner's whole Python stretch)
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 1e-4))
        # ... and what is alive when the job starts (the voicebank's Source objects, the process's set-up) is moved out of the
        # collector's sight for its duration: a full collection that walks it is 10-100 ms in the middle of a 4 ms batch
        # (``freeze_gc=False``: the caller manages that, e.g. a server that has frozen its heap already)
        import gc
        frozen = self.freeze_gc and gc.get_freeze_count() == 0
        if frozen:
            gc.freeze()
        try:
            yield from self._render_iter(it, ahead, seed, note_ids, pcm16)
        finally:
            sys.setswitchinterval(old_interval)
            if frozen:
                gc.unfreeze()

    def _render_iter(self, it, ahead, seed, note_ids, pcm16):
        import collections

        def feed():
            while len(ahead) < len(self.lanes) + self.workers:
                nxt = next(it, None)
                if nxt is None:
                    return
                ahead.append(self.pool.submit(self._prepare, nxt[0], nxt[1], note_ids))

        flying = collections.deque()                           # (lane, event, prep, out, samples)
        k = 0
        feed()
        while ahead or flying:
            if ahead:
                if self.trace is not None:
                    tw = time.perf_counter()
                prep, counts = ahead.popleft().result()
                if self.trace is not None:
                    self.trace.append(("wait_prepared", k, tw, time.perf_counter()))
                    tw = time.perf_counter()
                feed()
                ln = self.lanes[k % len(self.lanes)]
                k += 1
                r = ln["r"]
                sr, n_fft, hop, frames, samples, n = prep["geometry"]
                marks = [] if self.trace is not None else None   # (sub-steps of the launch, for scripts/pipeline_job.py --trace)
                want = torch.int16 if pcm16 else torch.float32
                # two pinned buffers per lane, taking turns: the lane's next batch may be launched while this one's audio is still
                # on its way home (or being read by the caller)
                if ln["host"] is None:
                    ln["host"] = [None, None]
                slot = ((k - 1) // len(self.lanes)) & 1
                hb = ln["host"][slot]
                if hb is None or hb.numel() < samples or hb.dtype != want:
                    hb = ln["host"][slot] = torch.empty(max(samples, int(1.25 * samples)), dtype=want, pin_memory=True)
                with torch.cuda.stream(ln["stream"]):
                    r.ctx.plan(sr, n_fft, hop)                 # no-ops once the lane has seen the geometry / the sizes
                    r.ctx.reserve(frames, samples, n)
                    if marks is not None:
                        marks.append(time.perf_counter())
                    out = r.run(prep, seed=seed)
                    if marks is not None:
                        marks.append(time.perf_counter())
                    if pcm16:
                        out["pcm"] = r.ctx.pcm16(out["mix"])
                    done = torch.cuda.Event()
                    done.record()
                    if marks is not None:
                        marks.append(time.perf_counter())
                mix = out["pcm"] if pcm16 else out["mix"]
                mix.record_stream(self.copy_stream)
                # (on this ROCm an asynchronous call now and then holds its caller for as long as the work queued in front of it
                # takes, 6-8 ms every ten to twenty batches — this copy, or the next launch when the copy is queued from a thread of
                # its own or on the lane's stream, which therefore bought nothing, and neither did HSA_KERNARG_POOL_SIZE,
                # ROC_SIGNAL_POOL_SIZE or GPU_MAX_HW_QUEUES; the conversion kernel storing its int16 samples straight into the
                # pinned buffer instead of a copy: 3.8 ms per batch against 2.8: scripts/pipeline_job.py --trace)
                home = self._ship_home(hb[:samples], mix, done)
                if marks is not None:
                    marks.append(time.perf_counter())
                flying.append((hb, home, prep, out, samples, counts))
                if self.trace is not None:
                    self.trace.append(("launch", k - 1, tw, time.perf_counter(), tuple(marks)))
            # one more batch in flight than there are lanes: batch k + 1 is queued behind batch k on the device before batch k - 1 is
            # home, so the device never waits for the host's launch (with `lanes` in flight it idled ~0.5 ms per batch: the launch
            # came after the previous audio's 1.8 ms trip home)
            if flying and (len(flying) >= len(self.lanes) + 1 or not ahead):
                hb, home, prep, out, samples, counts = flying.popleft()
                if self.trace is not None:
                    tw = time.perf_counter()
                home.synchronize()
                if self.trace is not None:
                    self.trace.append(("wait_audio", -1, tw, time.perf_counter()))
                if len(counts) == 1:
                    yield hb[:samples].numpy(), prep["sample_off"]
                else:                                          # the caller's batches of a coalesced device batch, one by one
                    audio, off, a = hb[:samples].numpy(), prep["sample_off"], 0
                    for c in counts:
                        yield audio[off[a]:off[a + c]], off[a:a + c + 1] - off[a]
                        a += c
                del prep, out
        for ln in self.lanes:                                  # the device is idle now: what the asynchronous calls flagged
            ln["r"].ctx.check()

    def _ship_home(self, dst, src, done):
        """the D2H copy behind ``done`` on the copy stream; returns the event behind it"""
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(done)
            dst.copy_(src, non_blocking=True)
            home = torch.cuda.Event()
            home.record()
        return home

    def render_all(self, batches, seed: int = 0, note_ids=None):
        """Every note of every batch as its own float32 array (copies), batch by batch."""
        res = []
        for mix, off in self.render_iter(batches, seed=seed, note_ids=note_ids):
            res.append([mix[off[i]:off[i + 1]].copy() for i in range(len(off) - 1)])
        return res
