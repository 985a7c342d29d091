#!/usr/bin/env python3
"""cProfile of Renderer.prepare on a batch whose voicebank samples are not resident yet (fresh Source objects, fresh arena)."""
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
from goofer_amd.render import Renderer, Source, SourceArena  # noqa: E402

notes = 1024
ctx = Context(0)
ren = Renderer(ctx, hop=syn.config_geometry(3)["hop"])
raw = [syn.config_note(3, i) for i in range(notes)]
args = [syn.request_args(q) for _, q, _ in raw]
reqs = S.decode_request_batch(args)
mk = lambda: [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
p = ren.prepare((mk(), reqs), note_ids=list(range(notes)))
ren.run(p)
torch.cuda.synchronize()
del p
pr = cProfile.Profile()
for k in range(3):
    srcs = mk()
    ren.sources = SourceArena(ctx)
    t0 = time.perf_counter()
    pr.enable()
    p = ren.prepare((srcs, reqs), note_ids=list(range(notes)))
    pr.disable()
    print("fresh prepare %.1f ms" % (1e3 * (time.perf_counter() - t0)))
    del p
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
