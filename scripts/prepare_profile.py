"""cProfile of Renderer.prepare on the default workload (host cost of making a batch resident).
Usage (GPU box): python scripts/prepare_profile.py [config] [notes]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd import sampler as S
from goofer_amd import synthetic as syn
from goofer_amd.device import Context
from goofer_amd.render import Renderer, Source

config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
notes = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
geo = syn.config_geometry(config)
ctx = Context(0)
r = Renderer(ctx, hop=geo["hop"])
raw = [syn.config_note(config, i) for i in range(notes)]
reqs = [S.decode_request(*syn.request_args(q)) for _, q, _ in raw]
jobs = [(Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]), q) for (s, _, _), q in zip(raw, reqs)]
for _ in range(2):
    t0 = time.perf_counter(); r.prepare(jobs, note_ids=list(range(notes))); torch.cuda.synchronize(); print("prepare %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
r.prepare(jobs, note_ids=list(range(notes)))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
