"""Errors per output of one case of tests/test_gpu_synth.py::test_synthesize_random_kwargs_vs_oracle (soak triage): the case's
keyword set, then the same with one keyword group left out at a time, then under library options.
Usage (GPU box): python scripts/soak_kw_case.py <case> ["{...keywords instead of the case's own...}"] [--only]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden, rms_err
from goofer_amd import core
from goofer_amd.device import Context
from oracle import goofer_ref as R
import test_gpu_synth as T

case = int(sys.argv[1])
args_ = [a for a in sys.argv[2:] if not a.startswith("--")]
kw0 = eval(args_[0]) if args_ else T._random_kwargs(case)
g = golden("synthesize")
c = T._case(g, "plain")
ctx = Context(0)
args = (c["env"], c["f0"], c["mask"], np.empty(c["n"], bool), c["sr"])


def run(kw, label):
    phi = c["phi"]
    if "stretch_factor" in kw:
        n_new = len(R.stretch_feature(c["f0"], kw["stretch_factor"]))
        if "start_sec" in kw:
            a, b = int(kw["start_sec"] * c["sr"]), int(kw["end_sec"] * c["sr"])
            n_new = a + int((b - a) * kw["stretch_factor"]) + (len(c["f0"]) - b)
        phi = np.random.default_rng(case).uniform(0.0, 2.0 * np.pi, size=(c["env"].shape[0], 1 + n_new // c["hop"])).astype(np.float32)
    np.random.seed(300 + case)
    ref = R.synthesize(*args, n_fft=c["n_fft"], hop_length=c["hop"], formants=c["formants"], phi=phi, **kw)
    np.random.seed(300 + case)
    got = core.synthesize(*args, n_fft=c["n_fft"], hop_length=c["hop"], formants=c["formants"], phi=phi, ctx=ctx, **kw)
    print(label)
    for a, b, key in zip(got, ref, ("rec", "harm", "uv", "bre")):
        d = np.abs(a.astype(np.float64) - b)
        top = np.argsort(d)[-4:][::-1]
        print("  %-5s scaled rms err %.3e, max |d| %.3e, max |ref| %.3f; largest at %s; samples with |d| > 1e-3 max: %d, span %s" % (
            key, rms_err(a, b) / max(1.0, float(np.max(np.abs(b)))), float(d.max()), float(np.max(np.abs(b))),
            [(int(i), float(np.round(d[i], 6))) for i in top], int((d > 1e-3 * np.max(np.abs(b))).sum()),
            (int(np.nonzero(d > 1e-3 * np.max(np.abs(b)))[0][0]), int(np.nonzero(d > 1e-3 * np.max(np.abs(b)))[0][-1])) if (d > 1e-3 * np.max(np.abs(b))).any() else None))
    return got, ref


run(kw0, "the case: %r" % (kw0,))
if "--only" in sys.argv:
    sys.exit(0)
groups = {"stretch": ["stretch_factor", "start_sec", "end_sec"], "stretch region": ["start_sec", "end_sec"], "f0 jitter": ["f0_jitter", "f0_jitter_strength"],
          "volume jitter": ["volume_jitter", "volume_jitter_strength_harm", "volume_jitter_strength_breath", "volume_vibrato", "volume_jitter_speed"],
          "sub-harmonics": ["add_subharm", "subharm_weight", "subharm_semitones", "subharm_vibrato", "subharm_vibrato_rate", "subharm_vibrato_depth",
                            "subharm_vibrato_delay", "subharm_f0_jitter"],
          "cut_subharm_below_f0": ["cut_subharm_below_f0"], "formant shifts": ["formant_shift", "F1_shift", "F2_shift", "F3_shift", "F4_shift"],
          "pitch_shift": ["pitch_shift"]}
for name, keys in groups.items():
    if any(k in kw0 for k in keys):
        run({k: v for k, v in kw0.items() if k not in keys}, "without " + name)
if "subharm_semitones" in kw0:
    for st in kw0["subharm_semitones"]:
        run({**kw0, "subharm_semitones": [st]}, "one ratio: %d semitones" % st)
for opt, val in (("pulse_scan", 0), ("stems", 0), ("fused_ola", 0)):
    ctx.set_option(opt, val)
    run(kw0, "option %s = %d" % (opt, val))
    ctx.set_option(opt, 1)
