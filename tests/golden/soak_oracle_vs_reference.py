#!/usr/bin/env python3
"""The oracle's gf.synthesize against the REFERENCE's own, on the random keyword sets the GPU soak tests use
(tests/test_gpu_synth.py::_random_kwargs) — on the fixture's plain note and on a hard source per case.  The GPU tests compare the
HIP path with the oracle on tens of thousands of these; this closes the loop for a sample of them: oracle == reference.

Runs ONLY in the build container (imports /root/reference through make_golden's stubs; nothing is written but the log on stdout).
Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/soak_oracle_vs_reference.py [first_case] [cases]"""
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import make_golden as MG                      # installs the stubs, imports the reference as MG.gf

gf = MG.gf
from oracle import goofer_ref as R            # noqa: E402
from goofer_amd import synthetic as syn       # noqa: E402
import test_gpu_synth as T                    # noqa: E402  (the keyword generator)
from conftest import golden, rms_err          # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
c = T._case(golden("synthesize"), "plain")
worst, worst_case, t0 = 0.0, None, time.time()
combos = {"stretch+jitter": 0, "stretch+subharm": 0, "stretch+region": 0, "hard": 0}
for case in range(first, first + count):
    kw = T._random_kwargs(case)
    hard = case % 2 == 1
    if hard:
        src = syn.make_hard_source(9000 + case, seconds=0.3 + 0.05 * (case % 5))
        env, f0, mask, forms, sr, n = R.decode_env_from_knots(src["env_pack"]), src["f0"], src["mask"], src["formants"], src["sr"], src["y_len"]
    else:
        env, f0, mask, forms, sr, n = c["env"], c["f0"], c["mask"], c["formants"], c["sr"], c["n"]
    hop, n_fft = 256, 1024
    n_new = n
    if "stretch_factor" in kw:
        n_new = len(R.stretch_feature(f0, kw["stretch_factor"]))
        if "start_sec" in kw:
            a, b = int(kw["start_sec"] * sr), int(kw["end_sec"] * sr)
            n_new = a + int((b - a) * kw["stretch_factor"]) + (n - b)
    combos["stretch+jitter"] += "stretch_factor" in kw and bool(kw.get("f0_jitter"))
    combos["stretch+subharm"] += "stretch_factor" in kw and bool(kw.get("add_subharm"))
    combos["stretch+region"] += "start_sec" in kw
    combos["hard"] += hard
    seed = 5000 + case
    phi = MG._orig_default_rng(seed).uniform(0.0, 2.0 * np.pi, size=(env.shape[0], 1 + n_new // hop)).astype(np.float32)
    args = (env, np.asarray(f0, dtype=np.float64), mask, np.empty(n, bool), sr)
    MG._RNG_SEED[0] = seed
    np.random.seed(300 + case)
    try:
        ref = gf.synthesize(*args, n_fft=n_fft, hop_length=hop, formants={k: np.array(v) for k, v in forms.items()}, **kw)
        ref_err = None
    except Exception as e:                                      # (what the oracle must raise as well)
        ref, ref_err = None, e
    MG._RNG_SEED[0] = None
    np.random.seed(300 + case)
    try:
        got = R.synthesize(env, f0, mask, np.empty(n, bool), sr, n_fft=n_fft, hop_length=hop, formants=forms, phi=phi, **kw)
        got_err = None
    except Exception as e:
        got, got_err = None, e
    if ref_err is not None or got_err is not None:
        same = type(ref_err) is type(got_err)
        print("case %d: reference %r, oracle %r%s" % (case, ref_err, got_err, "" if same else "   <-- DIFFERENT"))
        if not same:
            worst, worst_case = float("inf"), case
        continue
    e = max(rms_err(a_, b_) / max(1.0, float(np.max(np.abs(b_)))) for a_, b_ in zip(got, ref))
    if e > worst:
        worst, worst_case = e, case
    if e > 1e-6:
        print("case %d: %.3e  %r" % (case, e, kw))
print("%d cases from %d (%s): worst oracle-vs-reference error %.3e at case %s; %.0f s" % (count, first, combos, worst, worst_case, time.time() - t0))
