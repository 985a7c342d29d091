"""Time of the library's host planner (goofer_host_plan_notes) on the default workload's 1024 requests, by thread count.
Usage: python scripts/plan_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from goofer_amd import sampler as S, synthetic as syn
from goofer_amd.render import Source

raw = [syn.config_note(3, i) for i in range(1024)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
rb = S.decode_request_batch(args)
tracks = [sc.tracks64() for sc in srcs]
ylen = np.array([sc.ylen for sc in srcs])
T = np.array([sc.knots.shape[1] for sc in srcs])
rec = S.plan_records(rb, 44100, ylen, T, tracks)
for th in (1, 2, 4, 8, 16):
    best = 1e9
    for _ in range(9):
        t0 = time.perf_counter()
        pb = S.plan_native(rec, 256, True, keep=(tracks, rec), threads=th)
        best = min(best, time.perf_counter() - t0)
    print(th, "threads: %.2f ms" % (best * 1e3))
