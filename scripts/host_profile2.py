"""cProfile of the host side of one batch (decode, Renderer.prepare without device calls, Renderer.run): where the interpreter's
time goes.  Usage (GPU box): python scripts/host_profile2.py [notes]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd import sampler as S
from goofer_amd import synthetic as syn
from goofer_amd.device import Context
from goofer_amd.render import Renderer, Source

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = Context(0)
ren = Renderer(ctx, hop=256)
raw = [syn.config_note(3, i) for i in range(notes)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
ids = list(range(notes))
for _ in range(3):
    prep = ren.prepare((srcs, S.decode_request_batch(args)), note_ids=ids)
    ren.run(prep)
torch.cuda.synchronize()
for what in ("decode", "prepare", "run"):
    pr = cProfile.Profile()
    R = 30
    for _ in range(R):
        if what == "decode":
            pr.enable(); rb = S.decode_request_batch(args); pr.disable()
        elif what == "prepare":
            rb = S.decode_request_batch(args)
            pr.enable(); prep = ren.prepare((srcs, rb), note_ids=ids, device_calls=False); pr.disable()
            ctx.reserve(*prep["geometry"][3:])
        else:
            prep = ren.prepare((srcs, S.decode_request_batch(args)), note_ids=ids)
            pr.enable(); out = ren.run(prep); pr.disable()
            torch.cuda.synchronize()
    print("==== %s: per batch, ms (tottime / cumtime scaled by 1e3 / %d)" % (what, R))
    st = pstats.Stats(pr)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print("  %7.3f %7.3f  %5d  %s:%d %s" % (tt / R * 1e3, ct / R * 1e3, nc // R, os.path.basename(fn), line, name))
