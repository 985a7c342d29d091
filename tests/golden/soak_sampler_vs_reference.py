#!/usr/bin/env python3
"""The oracle's full sampler render (oracle/sampler_ref.render) against the REFERENCE's GooferResampler on the random requests of
the GPU soak tests: even cases = tests/test_gpu_sampler.py::test_random_flag_combinations_vs_oracle's draw (every fourth source
hard, 'fstb' / 'fstd' included), odd cases = test_random_extreme_requests_vs_oracle's (every third source hard).  The GPU tests
compare the HIP path with the oracle on ~100 000 of these; this closes the loop for a sample: oracle == reference, including
which requests the reference refuses.

Runs ONLY in the build container (imports /root/reference through make_golden's stubs; writes nothing but the log on stdout).
Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/soak_sampler_vs_reference.py [first_case] [cases]"""
import os
import sys
import tempfile
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import logging
import make_golden as MG                      # installs the stubs, imports the reference
logging.disable(logging.CRITICAL)             # (the reference narrates every render)

from oracle import sampler_ref as SR          # noqa: E402
from goofer_amd import synthetic as syn       # noqa: E402
from conftest import rms_err                  # noqa: E402

KEYS = ("pitch", "velocity", "flags", "offset", "length", "consonant", "cutoff", "volume", "modulation", "tempo", "pitch_string")


def flag_case(case):
    rng = MG._orig_default_rng(9000 + case)
    src = (syn.make_hard_source if case % 4 == 3 else syn.make_source)(4000 + case, seconds=float(rng.uniform(0.3, 0.6)))
    flags = syn.random_flags(rng)
    pitch = ["A3", "C4", "E4", "G#4", "D5"][int(rng.integers(0, 5))]
    args = (pitch, str(int(rng.choice([60, 100, 140]))), flags, str(int(rng.integers(0, 60))), str(int(rng.integers(200, 700))),
            str(int(rng.integers(0, 120))), str(int(rng.choice([-200, 30, 80]))), str(int(rng.integers(50, 121))), "0",
            "!" + str(int(rng.choice([90, 120, 150]))), ["AA", "AA#5#AF#3#/+", "B7CPCV#2#Cb"][int(rng.integers(0, 3))])
    if rng.random() < 0.25:
        for name in ("fstb", "fstd"):
            if rng.random() < 0.6 and name not in flags:
                flags += "%s%d" % (name, int(rng.integers(-40, 41)))
        args = args[:2] + (flags,) + args[3:]
    return src, args, 800 + case, 77 + case


def extreme_case(case):
    rng = MG._orig_default_rng(90000 + case)
    src = (syn.make_hard_source if case % 3 == 2 else syn.make_source)(95000 + case, seconds=float(rng.uniform(0.15, 0.7)))
    flags = syn.random_flags(rng) if rng.random() < 0.7 else ""
    pitch = ["C2", "A2", "C4", "B5", "C7"][int(rng.integers(0, 5))]
    bend = ["AA", "AA#50#", "/+/+/+#9#AAAA#3#gA", "B7CPCV#2#Cb" * 6, "AAABACADAEAFAGAH" * 4][int(rng.integers(0, 5))]
    args = (pitch, str(int(rng.choice([0, 1, 100, 199, 200]))), flags, str(int(rng.choice([0, 1, 5, 30, 200]))),
            str(int(rng.choice([5, 12, 40, 120, 2500]))), str(int(rng.choice([0, 1, 40, 300]))),
            str(int(rng.choice([-400, -50, 0, 1, 50, 350]))), str(int(rng.choice([0, 1, 100, 200]))), "0",
            "!" + str(int(rng.choice([20, 60, 120, 480]))), bend)
    return src, args, 1900 + case, 277 + case


first = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst, worst_case, refused, differ, t0 = 0.0, None, 0, 0, time.time()
with tempfile.TemporaryDirectory() as tmp:
    for case in range(first, first + count):
        src, args, seed, legacy = (extreme_case if case % 2 else flag_case)(case)
        req = dict(zip(KEYS, args))
        try:
            ref, _, _, _ = MG._run_sampler(src, req, seed, tmp, legacy_seed=legacy)
            ref_err = None
        except Exception as e:
            ref, ref_err = None, e
            MG._WRITTEN.clear()
        feats = (src["env_pack"], src["f0"].copy(), src["mask"].copy(), {k: v.copy() for k, v in src["formants"].items()}, src["sr"], src["y_len"])
        np.random.seed(legacy)
        try:
            got = SR.render(feats, SR.decode_request(*args), seed=seed)
            got_err = None
        except Exception as e:
            got, got_err = None, e
        if ref_err is not None or got_err is not None:
            refused += 1
            same = (ref_err is None) == (got_err is None)
            differ += not same
            print("case %d %r: reference %s, oracle %s%s" % (case, args, type(ref_err).__name__, type(got_err).__name__, "" if same else "   <-- DIFFERENT"), flush=True)
            continue
        if ref.shape != got.shape:
            differ += 1
            print("case %d %r: shapes %s / %s   <-- DIFFERENT" % (case, args, ref.shape, got.shape), flush=True)
            continue
        e = rms_err(got, ref) / max(1.0, float(np.max(np.abs(ref))))
        if e > worst:
            worst, worst_case = e, case
        if e > 1e-5:
            print("case %d: %.3e  %r" % (case, e, args), flush=True)
print("%d cases from %d: %d refused by both, %d different; worst oracle-vs-reference error of the rest %.3e at case %s; %.0f s"
      % (count, first, refused - differ, differ, worst, worst_case, time.time() - t0))
