"""GPU: the batch pipeline of long jobs (goofer_amd.render.PipelinedRenderer: two batches in flight on two handles, host
planning on worker threads, the mix downloaded under the next step) against the one-batch-at-a-time Renderer, bit for bit;
and the columnar request path (sampler.decode_request_batch -> Renderer.prepare) against the per-note Request objects."""
import numpy as np
import pytest

from goofer_amd import synthetic as syn

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _batches(config, sizes, first=0):
    from goofer_amd.render import Source
    out, i = [], first
    for n in sizes:
        srcs, args = [], []
        for _ in range(n):
            src, req, _ = syn.config_note(config, i)
            srcs.append(Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]))
            args.append(syn.request_args(req))
            i += 1
        out.append((srcs, args))
    return out


@pytest.mark.parametrize("config,sizes", [(3, [40, 7, 64, 1, 33, 20]), (4, [50, 50, 50])])
def test_pipelined_batches_equal_single_batches(config, sizes):
    from goofer_amd import sampler as S
    from goofer_amd.device import Context
    from goofer_amd.render import PipelinedRenderer, Renderer
    geo = syn.config_geometry(config)
    batches = _batches(config, sizes)
    ctx = Context(0)
    try:
        r = Renderer(ctx, hop=geo["hop"])
        want = []
        for srcs, args in batches:
            reqs = S.decode_requests(args)
            want.append(r.render(list(zip(srcs, reqs)), seed=11))
    finally:
        ctx.close()
    # (coalesce: two / four of the caller's batches as one device batch, handed back one by one with the ids of their own batch)
    for depth, workers, coalesce in ((2, 2, 1), (1, 1, 1), (3, 1, 1), (2, 3, 2), (2, 2, 4)):
        p = PipelinedRenderer(0, hop=geo["hop"], depth=depth, workers=workers, coalesce=coalesce)
        try:
            got = p.render_all(batches, seed=11)
        finally:
            p.close()
        assert len(got) == len(want)
        for b, (g, w) in enumerate(zip(got, want)):
            assert len(g) == len(w)
            for i, (a, c) in enumerate(zip(g, w)):
                assert np.array_equal(a, c), (depth, coalesce, b, i)


def test_prepare_from_columns_equals_prepare_from_requests():
    """(sources, RequestBatch) — the argument strings decoded straight to columns — against the list of (Source, Request)
    pairs: the same plans, the same render, for the whole flag vocabulary."""
    from goofer_amd import sampler as S
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    rng = np.random.default_rng(77)
    srcs, args = [], []
    for i in range(48):
        src, req, _ = syn.config_note(3, i)
        srcs.append(Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]))
        flags = syn.random_flags(rng)
        for bad in ("sh", "sr", "sj"):                          # the legacy-RNG / fresh-generator draws differ from call to call
            flags = __import__("re").sub(bad + r"-?\d+", "", flags)
        args.append(syn.request_args(syn.make_request(3000 + i, flags, length_ms=int(rng.integers(80, 1500)))))
    ctx = Context(0)
    try:
        r = Renderer(ctx)
        a = r.render(list(zip(srcs, S.decode_requests(args))), seed=5)
        prep = r.prepare((srcs, S.decode_request_batch(args)))
        out = r.run(prep, seed=5)
        ctx.check()
        mix = out["mix"].cpu().numpy()
        off = prep["sample_off"]
        for i in range(len(srcs)):
            assert np.array_equal(mix[off[i]:off[i + 1]], a[i]), i
    finally:
        ctx.close()


def test_device_pcm16_equals_the_wav_writers_arithmetic():
    """goofer_pcm16 (the finished audio converted on the device, two bytes per sample over PCIe) = write_wav's clip / scale /
    round-half-even on the host, sample for sample — incl. values past full scale, exact halves and ragged lengths; and the
    pipelined renderer's pcm16 stream = the conversion of its fp32 stream."""
    from goofer_amd.device import Context
    from goofer_amd.render import PipelinedRenderer
    ctx = Context(0)
    try:
        rng = np.random.default_rng(3)
        for n in (1, 7, 8, 9, 4099, 100003):
            x = (rng.standard_normal(n) * 0.6).astype(np.float32)
            x[::17] = (rng.integers(-32768, 32768, size=x[::17].shape) + 0.5) / 32768.0      # exact halves: ties to even
            x[::29] = rng.choice([1.0, -1.0, 1.5, -2.0, 0.99998474, 0.9999999], size=x[::29].shape)
            want = np.round(np.clip(x.astype(np.float64), -1.0, 1.0 - 1.0 / 32768) * 32768.0).astype(np.int16)
            for off in (0, 1):                                  # 16-byte aligned and not
                t = torch.zeros(n + off, dtype=torch.float32, device="cuda")
                t[off:] = torch.from_numpy(x).cuda()
                got = ctx.pcm16(t[off:]).cpu().numpy()
                assert np.array_equal(got, want), (n, off)
    finally:
        ctx.close()
    batches = _batches(3, [24, 5, 31])
    p = PipelinedRenderer(0, depth=2, workers=2)
    try:
        f32 = [(m.copy(), o.copy()) for m, o in p.render_iter(batches, seed=4)]
        i16 = [(m.copy(), o.copy()) for m, o in p.render_iter(batches, seed=4, pcm16=True)]
    finally:
        p.close()
    for (a, oa), (b, ob) in zip(f32, i16):
        assert b.dtype == np.int16 and np.array_equal(oa, ob)
        assert np.array_equal(b, np.round(np.clip(a.astype(np.float64), -1.0, 1.0 - 1.0 / 32768) * 32768.0).astype(np.int16))


def test_two_arenas_sharing_sources_from_two_threads():
    """Source._reg_key holds the key of ONE arena.  Two Renderers (own handles, own arenas) render the same Source
    objects from two threads with a 10 us switch interval: every lookup must return rows of ITS arena (checked inside
    SourceArena.lookup) and every render must equal the single-threaded one bit for bit.  A copied Source registers afresh."""
    import copy
    import sys
    import threading
    from goofer_amd import sampler as S
    from goofer_amd import synthetic as syn
    from goofer_amd.device import Context
    from goofer_amd.render import Renderer, Source
    srcs, reqs = [], []
    for i in range(24):
        src, req, _ = syn.config_note(3, i)
        srcs.append(Source.from_pack(src["env_pack"], src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]))
        reqs.append(S.decode_request(*syn.request_args(req)))
    jobs = list(zip(srcs, reqs))
    ctxs = [Context(0), Context(0)]
    try:
        rs = [Renderer(c) for c in ctxs]
        seeds = list(range(100, 124))                          # injected phases: a note's noise does not depend on its position
        ref = [m.copy() for m in rs[0].render(jobs, phi_seeds=seeds)]
        errs = []
        where = {id(sc): j for j, sc in enumerate(srcs)}

        def work(r, k):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for it in range(12):
                        rot = (it + k) % 5                      # different orders: different rows per arena
                        sub, sd = jobs[rot:] + jobs[:rot], seeds[rot:] + seeds[:rot]
                        outs = r.render(sub, phi_seeds=sd)
                        for (sc, _), o in zip(sub, outs):
                            if not np.array_equal(o, ref[where[id(sc)]]):
                                errs.append((k, it))
            except Exception as e:                            # noqa: BLE001
                errs.append(repr(e))

        old = sys.getswitchinterval()
        sys.setswitchinterval(1e-5)
        try:
            ts = [threading.Thread(target=work, args=(rs[k], k)) for k in range(2)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        finally:
            sys.setswitchinterval(old)
        assert not errs, errs[:3]
        # a copy is another sample: it does not inherit the original's arena row
        c2 = copy.copy(srcs[0])
        assert c2._reg_key == -1 and srcs[0]._reg_key != -1
        d2 = copy.deepcopy(srcs[1])
        assert d2._reg_key == -1
        (o,) = rs[0].render([(d2, reqs[1])], phi_seeds=[seeds[1]])
        assert np.array_equal(o, ref[1])
        srcs[2].mask = srcs[2].mask.copy()                     # re-assigned feature array: the registration is forgotten
        assert srcs[2]._reg_key == -1
    finally:
        for c in ctxs:
            c.close()
