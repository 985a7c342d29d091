"""cProfile of the host side of one batch, single thread: decode_request_batch, Renderer.prepare, Renderer.run (launch only).
Usage (GPU box): python scripts/host_profile.py [what=run|prepare|decode] [top]"""
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

what = sys.argv[1] if len(sys.argv) > 1 else "run"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ctx = Context(0)
ren = Renderer(ctx, hop=256)
raw = [syn.config_note(3, i) for i in range(1024)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
ids = list(range(1024))
reqs = S.decode_request_batch(args)
prep = ren.prepare((srcs, reqs), note_ids=ids)
for _ in range(3):
    ren.run(prep, seed=0)
torch.cuda.synchronize()


def one():
    if what == "run":
        ren.run(prep, seed=0)
    elif what == "prepare":
        ren.prepare((srcs, reqs), note_ids=ids)
    else:
        S.decode_request_batch(args)


R = 40
t0 = time.perf_counter()
for _ in range(R):
    one()
host = (time.perf_counter() - t0) / R * 1e3
torch.cuda.synchronize()
print("%s: %.3f ms of host time per call" % (what, host))
pr = cProfile.Profile()
pr.enable()
for _ in range(R):
    one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
