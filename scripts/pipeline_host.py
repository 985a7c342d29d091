#!/usr/bin/env python3
"""Where a double-buffered host loop (prepare(k + 1) on a second thread beside run(k)) spends its time: per round the wall
time, the worker's prepare time and the main thread's run time.   python scripts/pipeline_host.py [rounds]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goofer_amd import sampler as S
from goofer_amd import synthetic as syn
from goofer_amd.device import Context
from goofer_amd.render import Renderer, Source

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = Context(0)
ren = Renderer(ctx)
raw = [syn.config_note(3, i) for i in range(1024)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
ids = list(range(1024))
T = {}


def prepare(tag=None):
    t0 = time.perf_counter()
    reqs = S.decode_requests(args)
    t1 = time.perf_counter()
    prep = ren.prepare(list(zip(srcs, reqs)), note_ids=ids)
    t2 = time.perf_counter()
    if tag is not None:
        T[tag] = (1e3 * (t1 - t0), 1e3 * (t2 - t1))
    return prep


prep = prepare()
host = torch.empty(prep["samples"], dtype=torch.float32).pin_memory()


def run(p):
    t0 = time.perf_counter()
    out = ren.run(p, seed=0)
    t1 = time.perf_counter()
    host.copy_(out["mix"], non_blocking=True)
    torch.cuda.synchronize()
    return 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)


for mode in ("serial", "threads", "threads, switch interval 1e-4"):
    if mode.endswith("1e-4"):
        sys.setswitchinterval(1e-4)
    prep = prepare()
    torch.cuda.synchronize()
    box = {}
    t0 = time.perf_counter()
    rows = []
    for k in range(rounds):
        a = time.perf_counter()
        if mode == "serial":
            r = run(prep)
            prep = prepare(k)
        else:
            th = threading.Thread(target=lambda: box.__setitem__("p", prepare(k)))
            th.start()
            r = run(prep)
            th.join()
            prep = box.pop("p")
        rows.append((1e3 * (time.perf_counter() - a), r, T.get(k)))
    dt = 1e3 * (time.perf_counter() - t0) / rounds
    print("%s: %.2f ms per batch" % (mode, dt))
    for w, r, p in rows[:4]:
        print("   round %.2f ms | run: enqueue %.2f wait+copy %.2f | prepare: decode %.2f plan+upload %.2f" % (w, r[0], r[1], p[0], p[1]))
