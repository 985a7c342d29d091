"""The C-ABI library's host planner (csrc/planner.hip: goofer_host_plan_notes, goofer_host_decode_bends) against the numpy
planner of goofer_amd/sampler.py — which the reference's 53 index-plan fixtures and the product tests pin — bit for bit:
cut points, frame taps and weights, sample counts, fry ranges, the formant tracks handed to synthesize and the repaired +
smoothed tracks of the formant-strength gain.  Pure host code: no GPU."""
import numpy as np
import pytest

from goofer_amd import sampler as S
from conftest import golden


def _same(A, B):
    for name in A.geo.dtype.names:
        if name == "reserved":
            continue
        assert np.array_equal(A.geo[name], B.geo[name]), name
    for nm in ("tap_idx", "tap_w", "formants", "fst"):
        a, b = getattr(A, nm), getattr(B, nm)
        assert a.shape == b.shape and a.dtype == b.dtype, nm
        assert np.array_equal(a, b, equal_nan=True), nm


def _native(jobs, hop, trim):
    tracks = [S.source_tracks64(j[4]) for j in jobs]
    assert all(t is not None for t in tracks)
    rec = S.plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
    return S.plan_native(rec, hop, trim, keep=(tracks, rec))


def _random_jobs(seed, n, hop=256, sr=44100):
    rng = np.random.default_rng(seed)
    jobs = []
    for i in range(n):
        ylen = int(rng.integers(hop * 3, sr * 2))
        T = 1 + ylen // hop
        forms = {}
        for k in (1, 2, 3, 4):
            tr = (400.0 * k + 900.0 * rng.random(T)).astype(np.float64)
            mode = rng.integers(0, 8)
            if mode == 0:                                     # out-of-range stretches and NaNs: the repair path
                a = int(rng.integers(0, T))
                tr[a:a + int(rng.integers(1, 30))] = [0.0, np.nan, 1e6][int(rng.integers(0, 3))]
            elif mode == 1:
                tr[:int(rng.integers(1, 12))] = 0.0           # bad from the start: float32 extrapolation on the left
            elif mode == 2:
                tr[-int(rng.integers(1, 12)):] = np.inf
            elif mode == 3 and k == 4:
                tr[:] = 0.0                                   # nothing good: 300 Hz
            elif mode == 4:
                tr[:] = 0.0
                tr[int(rng.integers(0, T))] = 700.0 * k       # a single good value
            forms[k] = tr
        if rng.random() < 0.2:
            forms[5] = 4500.0 + rng.random(T)                 # F5 rides along unused
        if rng.random() < 0.1:
            forms[2] = forms[2][:max(1, T - int(rng.integers(1, 40)))]     # tracks of unequal length
        flags = "L%d" % int(rng.integers(0, 3))
        if rng.random() < 0.3:
            flags += "R1"
        if rng.random() < 0.3:
            flags += "vf%dvl%d" % (int(rng.integers(-100, 101)), int(rng.integers(0, 101)))
        dur = ylen / sr * 1000.0
        off = float(rng.integers(0, int(dur * 0.4)))
        cons = float(rng.integers(0, int(dur * 0.4))) if rng.random() < 0.85 else 0.0
        cutoff = float(rng.integers(-int(dur * 0.5), int(dur * 0.3)))
        length = float(rng.integers(20, 2500))
        vel = [100, 100, 60, 140, 0, 200, 99][int(rng.integers(0, 7))]
        req = S.decode_request("C4", str(vel), flags, str(off), str(length), str(cons), str(cutoff), "100", "0", "!120", "AA")
        jobs.append((req, sr, ylen, T, forms))
    return jobs


def _plannable(jobs, hop):
    keep = []
    for j in jobs:
        try:
            S._GEO_CACHE.clear()
            S.plan_notes([j], hop)
            keep.append(j)
        except (ZeroDivisionError, ValueError):
            pass
    return keep


@pytest.mark.parametrize("seed,hop", [(1, 256), (2, 256), (3, 96), (4, 512)])
def test_native_plans_equal_numpy_plans(seed, hop):
    jobs = _plannable(_random_jobs(seed, 160, hop=hop, sr=44100 if hop != 96 else 96000), hop)
    assert len(jobs) > 100
    modes = {j[0].loop_mode for j in jobs}
    assert modes == {"concat", "avg", "stretch"}
    for trim in (False, True):
        S._GEO_CACHE.clear()
        A = S.plans_to_arrays(S.plan_notes(jobs, hop), hop, trim)
        B = _native(jobs, hop, trim)
        assert B is not None
        _same(A, B)
    assert A.geo["vel_active"].any() and (A.geo["fry_b"] > A.geo["fry_a"]).any() and A.geo["env_f64"].any()


def test_native_plans_on_the_reference_index_plan_cases():
    """The argument sets of the reference's 53 index-plan fixtures (every slicing variant, L0/L1/L2, reverse, velocity): the cases
    the reference renders come out equal to the numpy planner's, the cases it refuses make plan_native step aside."""
    from test_product_sampler import probe_source
    g = golden("index_plans")
    env, f0, mask, forms, sr, n, T = probe_source()
    forms = {k: np.asarray(v, dtype=np.float64) for k, v in forms.items()}
    good, refused = [], []
    for tag in g["names"]:
        req = S.decode_request(*[str(a) for a in g[f"{tag}_args"]])
        (refused if f"{tag}_error" in g.files else good).append((req, sr, n, T, forms))
    assert len(good) >= 45 and refused
    for trim in (False, True):
        _same(S.plans_to_arrays(S.plan_notes(good, 256), 256, trim), _native(good, 256, trim))
    for j in refused:
        assert _native([j], 256, False) is None
        with pytest.raises((ZeroDivisionError, ValueError)):
            S.plan_notes_arrays([j], 256)


def test_thread_count_changes_nothing():
    jobs = _plannable(_random_jobs(9, 300), 256)
    tracks = [S.source_tracks64(j[4]) for j in jobs]
    rec = S.plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
    one = S.plan_native(rec, 256, True, keep=(tracks, rec), threads=1)
    many = S.plan_native(rec, 256, True, keep=(tracks, rec), threads=5)
    _same(one, many)


def test_unused_strength_columns_are_left_zero():
    """skip_unused_fst: the smoothed track of a formant whose 'fst' strength is off for the note is not computed (its column
    stays 0, the assembly never reads it); every other value of the plan is what the full plan holds."""
    jobs = _plannable(_random_jobs(11, 200), 256)
    rng = np.random.default_rng(4)
    flagged = []
    for j in jobs:                                            # the strengths: none, one, some, all
        fl = "".join("fst%s%d" % (c, int(rng.integers(-80, 80))) for c in "abcd" if rng.random() < 0.4)
        r = j[0]
        req = S.decode_request("C4", "100", fl, "0", "500", "0", "0", "100", "0", "!120", "AA")
        flagged.append((S.Request(**{**r.__dict__, "formant_strength": req.formant_strength}),) + tuple(j[1:]))
    tracks = [S.source_tracks64(j[4]) for j in flagged]
    a = [j[0] for j in flagged], [j[1] for j in flagged], [j[2] for j in flagged], [j[3] for j in flagged], tracks
    full = S.plan_native(S.plan_records(*a), 256, True, keep=tracks)
    rec = S.plan_records(*a, skip_unused_fst=True)
    lean = S.plan_native(rec, 256, True, keep=(tracks, rec))
    off = np.array([[abs(v) < 1e-6 for v in j[0].formant_strength] for j in flagged])
    assert off.any() and (~off).any() and off.all(axis=1).any() and np.array_equal(rec["fst_skip"] != 0, off)
    for nm in ("tap_idx", "tap_w", "formants"):
        assert np.array_equal(getattr(full, nm), getattr(lean, nm), equal_nan=True), nm
    for name in full.geo.dtype.names:
        assert np.array_equal(full.geo[name], lean.geo[name]), name
    rows_off = np.repeat(off, full.geo["n_out_rows"], axis=0)
    assert np.array_equal(lean.fst[~rows_off], full.fst[~rows_off]) and not lean.fst[rows_off].any() and full.fst[rows_off].any()


def test_a_block_too_small_for_the_batch_says_how_many_rows_it_needs():
    from goofer_amd import _lib
    jobs = _plannable(_random_jobs(12, 64), 256)
    tracks = [S.source_tracks64(j[4]) for j in jobs]
    rec = S.plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
    want = S.plan_native(rec, 256, True, keep=(tracks, rec))
    rows = want.tap_idx.shape[0]
    geo = np.zeros(len(jobs), dtype=_lib.PLAN_GEOMETRY)
    mk = lambda cap: (np.full((cap, 4), -7, np.int32), np.zeros((cap, 4)), np.zeros((cap, 4)), np.zeros((cap, 4), np.float32))
    small = mk(rows - 1)
    with pytest.raises(S.StagingFull) as e:
        S.plan_native_into(rec, 256, True, geo, rows - 1, *small, keep=(tracks, rec))
    assert e.value.args[0] == rows and (small[0] == -7).all()          # sized, nothing written
    fit = mk(rows)
    got = S.plan_native_into(rec, 256, True, geo, rows, *fit, keep=(tracks, rec))
    _same(want, got)


def test_an_empty_stretched_track_is_refused_like_the_reference():
    """Soak case 240370 (round 6): 'L2', a 5 ms note (one frame wanted), no consonant frames, a 98-frame tail.  The reference cuts
    the envelope's tail to the frame it wants (SillySampler.py:631-636) but resamples the formant tracks (:721-726) to
    int(98 * (1 / 98)) = 0 frames, and np.pad(mode='edge') then refuses the empty track (:755-760): the render fails with a
    ValueError.  Both planners rendered it; now the numpy planner raises np.pad's error and the native one steps aside."""
    from goofer_amd import synthetic as syn
    from oracle import sampler_ref as SR
    src = syn.make_source(95000 + 240370, seconds=0.5711272864751321)
    args = ("C2", "0", "t-12L2sa25", "1", "5", "1", "0", "100", "0", "!120", "/+/+/+#9#AAAA#3#gA")
    assert int(98 * (1 / 98.0)) == 0
    feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()}, src["sr"], src["y_len"])
    with pytest.raises(ValueError, match="empty axis"):
        SR.render(feats, SR.decode_request(*args), seed=1)
    req, T = S.decode_request(*args), 1 + src["y_len"] // 256
    S._GEO_CACHE.clear()
    with pytest.raises(ValueError, match="empty axis"):
        S.plan_notes([(req, src["sr"], src["y_len"], T, src["formants"])], 256)
    assert _native([(req, src["sr"], src["y_len"], T, src["formants"])], 256, True) is None
    ok = S.decode_request(*(args[:4] + ("10",) + args[5:]))               # two frames wanted: int(98 * (2 / 98)) = 2
    S._GEO_CACHE.clear()
    _same(S.plans_to_arrays(S.plan_notes([(ok, src["sr"], src["y_len"], T, src["formants"])], 256), 256, True),
          _native([(ok, src["sr"], src["y_len"], T, src["formants"])], 256, True))


def test_odd_sources_take_the_numpy_planner():
    jobs = _plannable(_random_jobs(5, 6), 256)
    odd = [(j[0], j[1], j[2], j[3], {k: v.astype(np.float32) for k, v in j[4].items()}) for j in jobs]
    assert S.source_tracks64(odd[0][4]) is None
    S._GEO_CACHE.clear()
    _same(S.plans_to_arrays(S.plan_notes(odd, 256), 256, True), S.plan_notes_arrays(odd, 256, True))
    assert S.source_tracks64({0: [1.0], 1: [1.0], 2: [1.0], 3: [1.0], 4: [1.0]}) is None
    assert S.source_tracks64({1: [1.0], 2: [1.0], 3: [1.0]}) is None


def test_batch_decode_of_pitch_strings():
    rng = np.random.default_rng(3)
    alpha = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"
    texts = ["AA", "AA#5#AF#3#/+", "//#200#", "Ab#3", "4f"]
    for _ in range(200):
        t = ""
        for _ in range(int(rng.integers(1, 6))):
            t += "".join(alpha[int(c)] for c in rng.integers(0, 64, size=2 * int(rng.integers(1, 9))))
            if rng.random() < 0.6:
                t += "#%d#" % int(rng.integers(0, 40))
        texts.append(t)
    got = S.pitch_strings_to_cents(texts)
    for t, a in zip(texts, got):
        b = S.pitch_string_to_cents(t)
        assert a.dtype == b.dtype == np.float32 and np.array_equal(a, b), t
    # malformed strings: the batch answers / raises what the one-by-one decoder does
    for bad in ("A", "AA#x#", "#3#AA", "A?", "", "AA##"):
        try:
            want = S.pitch_string_to_cents(bad)
        except Exception as e:                                # noqa: BLE001
            with pytest.raises(type(e)):
                S.pitch_strings_to_cents(["AA", bad])
        else:
            assert np.array_equal(S.pitch_strings_to_cents(["AA", bad])[1], want)
    reqs = S.decode_requests([("C4", "100", "L1g-10", "30", "700", "80", "40", "100", "0", "!120", "AA#5#AF#3#/+"), ("A#3", "80")])
    one = S.decode_request("C4", "100", "L1g-10", "30", "700", "80", "40", "100", "0", "!120", "AA#5#AF#3#/+")
    assert reqs[0].loop_mode == "avg" and np.array_equal(reqs[0].bend, one.bend) and reqs[0].formant_shift == one.formant_shift
    assert reqs[1].pitch_m == S.note_to_midi("A#3") and reqs[1].length == 1.0


def test_native_planner_steps_aside_for_times_it_cannot_hold():
    """A request whose times are not finite, or whose sample counts would leave int32 (the geometry fields), is not planned by
    the library (undefined double -> int64 casts, wrapped counts, allocations that throw inside a worker thread): plan_native
    answers None and the numpy planner gets the note, as for the cases the reference refuses."""
    import ctypes as C
    jobs = _plannable(_random_jobs(11, 12), 256)
    tracks = [S.source_tracks64(j[4]) for j in jobs]
    rec = S.plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
    assert S.plan_native(rec, 256, True, keep=(tracks, rec)) is not None
    for field, value in (("length", float("nan")), ("length", float("inf")), ("offset", float("nan")), ("cutoff", -float("inf")),
                         ("consonant", float("inf")), ("length", 60000.0), ("offset", 1e12), ("cutoff", -1e15), ("consonant", 1e300),
                         ("vel_factor", float("inf")), ("vel_factor", float("nan"))):
        bad = rec.copy()
        bad[field][5] = value
        assert S.plan_native(bad, 256, True, keep=(tracks, bad)) is None, (field, value)
        h = C.c_void_p(123)
        taps = np.ascontiguousarray(S._gauss_taps_cached(4.0))
        rc = S._host_lib().goofer_host_plan_notes(bad.ctypes.data, bad.shape[0], 256, 1, taps.ctypes.data, (taps.size - 1) // 2, 3, C.byref(h))
        assert rc == -1 and not h.value                       # GOOFER_EINVAL, no handle left behind


def test_request_columns_equal_request_objects():
    """sampler.decode_request_batch (argument strings -> columns: numeric strings through the library's strtod parser, flag
    strings and note names once per distinct string, pitch strings in one decoder call) against the per-note decode_request."""
    from goofer_amd import synthetic as syn
    rng = np.random.default_rng(1)
    args = [syn.request_args(syn.config_note(c, i)[1]) for c in (3, 4, 5) for i in range(60)]
    for i in range(200):
        args.append(syn.request_args(syn.make_request(2000 + i, syn.random_flags(rng), length_ms=int(rng.integers(50, 2000)))))
    # literals the strict native parser hands back to float(): spaces, underscores, exponents, signs, several '!'
    args.append(("C4", " 100 ", "g10", "1_0", "1e3", "5.5", "-3", "+100", "0", "!!120.5", "AA#3#"))
    args.append(("A#3", "80"))                                  # defaults
    args.append(("F#2", "1e2", "", ".5", "5.", "-0", "0.0", "1E2", "00", "!060", "AA"))
    A = S.RequestBatch.from_requests(S.decode_requests(args))
    B = S.decode_request_batch(args)
    for f in S._SCALAR_FIELDS:
        assert np.array_equal(A.col[f], B.col[f]), f
    for f in ("f_shift", "formant_strength", "loop_code", "t_cents", "bend", "bend_off"):
        a, b = getattr(A, f), getattr(B, f)
        assert a.dtype == b.dtype and np.array_equal(a, b), f
    one = S.decode_request(*args[7])
    back = B.request(7)
    for f in S._SCALAR_FIELDS:
        assert getattr(one, f) == getattr(back, f), f
    assert np.array_equal(one.bend, back.bend) and one.f_shift == back.f_shift and one.loop_mode == back.loop_mode
    for bad in (("C4", "abc"), ("H9", "100"), ("C4", "100", "g"), ("C4", "100", "", "nope")):
        with pytest.raises((ValueError, TypeError)):
            S.decode_request(*bad)
        with pytest.raises((ValueError, TypeError)):
            S.decode_request_batch([("A#3", "80"), bad])
    assert len(S.decode_request_batch([])) == 0


def test_plans_written_into_the_callers_arrays():
    """goofer_host_plan_into (the planner writing into pinned staging memory of the caller) = goofer_host_plan_notes; a
    capacity that is too small is reported with the row count, nothing half-written is handed out."""
    from goofer_amd import _lib
    jobs = _plannable(_random_jobs(21, 200), 256)
    tracks = [S.source_tracks64(j[4]) for j in jobs]
    rec = S.plan_records([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs], tracks)
    ref = S.plan_native(rec, 256, True, keep=(tracks, rec))
    rows = ref.tap_idx.shape[0]
    cap = rows + 100
    geo = np.zeros(len(jobs), dtype=_lib.PLAN_GEOMETRY)
    ti, tw = np.full((cap, 4), -7, np.int32), np.zeros((cap, 4))
    fo, fs = np.zeros((cap, 4)), np.zeros((cap, 4), np.float32)
    got = S.plan_native_into(rec, 256, True, geo, cap, ti, tw, fo, fs, keep=(tracks, rec))
    _same(ref, got)
    assert (ti[rows:] == -7).all()
    with pytest.raises(S.StagingFull) as e:
        S.plan_native_into(rec, 256, True, geo, rows - 1, ti, tw, fo, fs, keep=(tracks, rec))
    assert e.value.args[0] == rows


def test_host_pack_is_a_concatenation_on_any_thread_count():
    """goofer_host_pack (the pinned-staging gather of the voicebank upload, render.SourceArena): pieces of any size, incl.
    empty ones, land back to back; a block that is too small is refused."""
    import ctypes as C
    from goofer_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    sizes = [0, 1, 5, 4 << 20, 0, 3_000_001, 17, 2_500_000, 1]
    pieces = [rng.integers(0, 255, n, dtype=np.uint8) for n in sizes]
    want = np.concatenate(pieces)
    n = len(pieces)
    srcs = (C.c_void_p * n)(*[p.ctypes.data if p.size else None for p in pieces])
    nbytes = (C.c_int64 * n)(*[p.nbytes for p in pieces])
    for threads in (1, 2, 5, 8, 64):
        dst = np.full(want.size + 64, 0xEE, np.uint8)
        got = lib.goofer_host_pack(srcs, nbytes, n, C.c_void_p(dst.ctypes.data), dst.nbytes, threads)
        assert got == want.size
        assert np.array_equal(dst[:want.size], want) and np.all(dst[want.size:] == 0xEE), threads
    dst = np.zeros(16, np.uint8)
    assert lib.goofer_host_pack(srcs, nbytes, n, C.c_void_p(dst.ctypes.data), dst.nbytes, 4) < 0
    assert lib.goofer_host_pack(None, None, 0, None, 0, 4) == 0
