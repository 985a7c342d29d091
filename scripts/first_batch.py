#!/usr/bin/env python3
"""Where the first batch of a fresh process goes (bench.py host_inclusive.first_batch): timers with a device synchronisation
between the phases, and the caching allocator's device allocations per phase.

    python scripts/first_batch.py [notes]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goofer_amd import sampler as S  # noqa: E402
from goofer_amd import synthetic as syn  # noqa: E402
from goofer_amd.device import Context  # noqa: E402
from goofer_amd.render import Renderer, Source, SourceArena  # noqa: E402

notes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = Context(0)
geo = syn.config_geometry(3)
ren = Renderer(ctx, hop=geo["hop"])
raw = [syn.config_note(3, i) for i in range(notes)]
args = [syn.request_args(q) for _, q, _ in raw]


def sources():
    return [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]


def allocs():
    st = torch.cuda.memory_stats()
    return st.get("num_device_alloc", 0), st.get("reserved_bytes.all.current", 0) >> 20


srcs = sources()
host_mix = None
for k in range(4):
    if k == 2:                                                 # a second "first batch": fresh sources, fresh arena, warm allocator
        srcs = sources()
        ren.sources = SourceArena(ctx)
    a0 = allocs()
    t0 = time.perf_counter()
    reqs = S.decode_request_batch(args)
    t1 = time.perf_counter()
    prep = ren.prepare((srcs, reqs), note_ids=list(range(notes)))
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    a1 = allocs()
    out = ren.run(prep, seed=0)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    if host_mix is None:
        host_mix = torch.empty(out["mix"].numel(), dtype=torch.float32).pin_memory()
        t5 = time.perf_counter()
    host_mix.copy_(out["mix"], non_blocking=True)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    a2 = allocs()
    print("pass %d: decode %.1f  prepare %.1f (+%.1f until the device is idle)  run: enqueue %.1f, done %.1f  download %.1f ms | "
          "device allocations: prepare %d, run %d; reserved %d MiB" %
          (k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5),
           a1[0] - a0[0], a2[0] - a1[0], a2[1]))
    del prep, out
