"""Errors per output of one case of tests/test_gpu_synth.py::test_synthesize_random_kwargs_vs_oracle (soak triage).
Usage (GPU box): python scripts/soak_kw_case.py <case> [key=value ...overrides]"""
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
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else None
g = golden("synthesize")
c = T._case(g, "plain")
ctx = Context(0)
args = (c["env"], c["f0"], c["mask"], np.empty(c["n"], bool), c["sr"])
for trial in ([kw] if kw is not None else []):
    np.random.seed(300 + case)
    ref = R.synthesize(*args, n_fft=c["n_fft"], hop_length=c["hop"], formants=c["formants"], phi=c["phi"], **trial)
    np.random.seed(300 + case)
    got = core.synthesize(*args, n_fft=c["n_fft"], hop_length=c["hop"], formants=c["formants"], phi=c["phi"], ctx=ctx, **trial)
    print(trial)
    for a, b, key in zip(got, ref, ("rec", "harm", "uv", "bre")):
        print("  %-5s rms err %.3e (scaled %.3e), max |d| %.3e at %d, max |ref| %.3f" % (key, rms_err(a, b), rms_err(a, b) / max(1.0, float(np.max(np.abs(b)))),
              float(np.max(np.abs(a - b))), int(np.argmax(np.abs(a - b))), float(np.max(np.abs(b)))))
