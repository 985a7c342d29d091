"""Where Renderer.prepare's host time goes, phase by phase (single thread, the default 1024-note batch).
Usage (GPU box): python scripts/prepare_phases.py [config] [notes]"""
import collections
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
ctx = Context(0)
ren = Renderer(ctx, hop=syn.config_geometry(config)["hop"])
raw = [syn.config_note(config, i) for i in range(notes)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
for _ in range(3):
    ren.prepare((srcs, S.decode_request_batch(args)), note_ids=list(range(notes)))
acc = collections.OrderedDict()
dec = 0.0
R = 20
for _ in range(R):
    t0 = time.perf_counter()
    reqs = S.decode_request_batch(args)
    dec += time.perf_counter() - t0
    ren.trace_prepare = []
    ren.prepare((srcs, reqs), note_ids=list(range(notes)))
    tr = ren.trace_prepare
    for (a, ta), (b, tb) in zip(tr, tr[1:]):
        acc[b] = acc.get(b, 0.0) + (tb - ta)
print("decode_request_batch %.3f ms" % (dec / R * 1e3))
tot = 0.0
for k, v in acc.items():
    print("%-16s %.3f ms" % (k, v / R * 1e3))
    tot += v / R * 1e3
print("prepare total %.3f ms" % tot)
