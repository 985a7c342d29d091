"""Host-side cost of making a batch resident (outside the timed steps of bench.py): decode the 13 arguments, plan, upload.
Usage (GPU box): python scripts/prepare_time.py [config] [notes]"""
import os
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
args = [syn.request_args(q) for _, q, _ in raw]
for rep in range(3):
    t0 = time.perf_counter()
    reqs = [S.decode_request(*a) for a in args]
    t1 = time.perf_counter()
    jobs = [(Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]), q) for (s, _, _), q in zip(raw, reqs)]
    t2 = time.perf_counter()
    prep = r.prepare(jobs, note_ids=list(range(notes)))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"config {config}, {notes} notes: decode {1e3 * (t1 - t0):.1f} ms, features {1e3 * (t2 - t1):.1f} ms, "
          f"prepare (plan + tables + upload) {1e3 * (t3 - t2):.1f} ms, total {1e3 * (t3 - t0):.1f} ms "
          f"= {notes / (t3 - t0):.0f} notes/s")
