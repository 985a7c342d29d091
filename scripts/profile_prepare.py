#!/usr/bin/env python3
"""Host profile of decode + Renderer.prepare for N notes of a BASELINE config (cProfile, needs the GPU for the uploads).

    python scripts/profile_prepare.py [config] [notes] [--prof]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goofer_amd import sampler as S  # noqa: E402
from goofer_amd import synthetic as syn  # noqa: E402
from goofer_amd.device import Context  # noqa: E402
from goofer_amd.render import Renderer, Source  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
notes = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = Context(0)
geo = syn.config_geometry(config)
ren = Renderer(ctx, hop=geo["hop"])
raw = [syn.config_note(config, i) for i in range(notes)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]


def once():
    t0 = time.perf_counter()
    reqs = S.decode_request_batch(args)
    t1 = time.perf_counter()
    prep = ren.prepare((srcs, reqs), note_ids=list(range(notes)))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), prep


for k in range(3):
    S._GEO_CACHE.clear()
    d, p, prep = once()
    print("config %d, %d notes, pass %d (cold geometry cache): decode %.1f ms, prepare %.1f ms, frames %d" % (config, notes, k, d, p, prep["frames"]))
d, p, prep = once()
print("warm geometry cache: decode %.1f ms, prepare %.1f ms" % (d, p))
if "--prof" in sys.argv:
    S._GEO_CACHE.clear()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        once()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(40)
