"""Per-batch intervals of goofer_amd.render.PipelinedRenderer on the default workload (the same 1024 argument lists and
sources batch after batch).  Usage (GPU box): python scripts/pipeline_job.py [batches] [depth] [workers] [--extra]
--extra: another resident workload in the process first (what bench.py's variants leave behind)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from goofer_amd import synthetic as syn
from goofer_amd.render import PipelinedRenderer, Source

argv = [a for a in sys.argv[1:] if not a.startswith("--")]   # [batches] [depth] [workers] [--pcm16] [--trace] [--plan-threads=N] [--extra] [--nogc]
rounds = int(argv[0]) if len(argv) > 0 else 40
depth = int(argv[1]) if len(argv) > 1 else 2
workers = int(argv[2]) if len(argv) > 2 else 4
raw = [syn.config_note(3, i) for i in range(1024)]
args = [syn.request_args(q) for _, q, _ in raw]
srcs = [Source.from_pack(s["env_pack"], s["f0"], s["mask"], s["formants"], s["sr"], s["y_len"]) for s, _, _ in raw]
if "--extra" in sys.argv:
    from goofer_amd.device import Context
    from goofer_amd.workload import SamplerWorkload
    c0 = Context(0)
    keep = [SamplerWorkload(c0, 3, list(range(1024))), SamplerWorkload(c0, 3, list(range(1024)), unvoiced_share=0.3)]
    for w in keep:
        w.step()
    torch.cuda.synchronize()
    if "--two" in sys.argv:                                    # the two-in-flight variant of bench.py: a second handle, two more streams
        cb = Context(0)
        wb = SamplerWorkload(cb, 3, list(range(1024)))
        both = [keep[0], wb]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for i in range(40):
            with torch.cuda.stream(streams[i % 2]):
                both[i % 2].step()
        torch.cuda.synchronize()
        if "--keepb" not in sys.argv:
            del wb, both
            cb.close()
for a in sys.argv:
    if a.startswith("--switch="):                              # the interpreter's thread switch interval (bench.py: 1e-4)
        sys.setswitchinterval(float(a.split("=")[1]))
if "--nogc" in sys.argv:
    import gc
    gc.disable()
def _throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except OSError:
        return 0, 0, 0
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-", " torch threads:", torch.get_num_threads())
thr = [("setup", _throttled(), time.perf_counter())]
coalesce = 1
for a in sys.argv:
    if a.startswith("--coalesce="):
        coalesce = int(a.split("=")[1])
pipe = PipelinedRenderer(0, depth=depth, workers=workers, coalesce=coalesce)
for a in sys.argv:
    if a.startswith("--plan-threads="):
        for ln in pipe.lanes:
            ln["r"].plan_threads = int(a.split("=")[1])
if "--trace" in sys.argv:
    pipe.trace = []
phase_marks = []
if "--phases" in sys.argv:
    import threading

    class _Marks(list):
        def append(self, m):
            list.append(self, (m[0], m[1], threading.get_ident()))
    for ln in pipe.lanes:
        ln["r"].trace_prepare = _Marks()
        phase_marks.append(ln["r"].trace_prepare)
ids = list(range(1024))
stamps = []
allocs = []
t_prev = time.perf_counter()
for mix, off in pipe.render_iter(((srcs, args) for _ in range(rounds)), note_ids=lambda k, n: ids, pcm16="--pcm16" in sys.argv):
    now = time.perf_counter()
    stamps.append(1e3 * (now - t_prev))
    if len(stamps) in (rounds // 4, rounds // 2, 3 * rounds // 4):
        thr.append(("batch %d" % len(stamps), _throttled(), now))
    allocs.append(torch.cuda.memory_stats().get("num_device_alloc", 0))
    t_prev = now
trace = pipe.trace
pipe.close()
thr.append(("end", _throttled(), time.perf_counter()))
for (a, ta, wa), (b, tb, wb) in zip(thr, thr[1:]):
    print("%-10s -> %-10s: %5.2f s wall, %6.2f s of CPU, throttled %d times for %.2f s" % (a, b, wb - wa, (tb[2] - ta[2]) / 1e6, tb[0] - ta[0], (tb[1] - ta[1]) / 1e6))
print("intervals (ms):", " ".join("%.1f" % v for v in stamps))
print("device allocations by the caching allocator in the second half of the job: %d; reserved %d MiB" % (allocs[-1] - allocs[len(allocs) // 2], torch.cuda.memory_stats().get("reserved_bytes.all.current", 0) >> 20))
tail = stamps[len(stamps) // 2:]
print("second half: mean %.2f ms, median %.2f, max %.2f  -> %.1f M frames/s" % (np.mean(tail), np.median(tail), max(tail), 194560 / np.mean(tail) / 1e3))

if trace:
    t0 = min(e[2] for e in trace)
    for what in ("prepare", "wait_prepared", "launch", "wait_audio"):
        ev = [e for e in trace if e[0] == what]
        d = [1e3 * (e[3] - e[2]) for e in ev]
        print("%-14s n %3d  mean %.2f ms  median %.2f  max %.2f   per batch over the job: %s" % (what, len(d), np.mean(d), np.median(d), max(d), " ".join("%.1f" % v for v in d[:80])))
    ev = [e for e in trace if e[0] == "launch" and len(e) > 4 and len(e[4]) == 4][len(trace) // 8:]
    if ev:
        parts = np.array([[1e3 * (b - a) for a, b in zip((e[2],) + e[4], e[4])] for e in ev])
        print("launch sub-steps behind the warm-up, median / mean ms: plan+reserve %.2f / %.2f, run %.2f / %.2f, pcm16 + event %.2f / %.2f, copy home queued %.2f / %.2f"
              % tuple(v for k in range(4) for v in (np.median(parts[:, k]), parts[:, k].mean())))
    ev = [e for e in trace if e[0] == "launch" and len(e) > 4 and e[3] - e[2] > 4e-3]
    for e in ev[:12]:
        m = (e[2],) + e[4]
        print("slow launch of batch %d: plan/reserve %.2f, run %.2f, pcm16 + event %.2f, copy home queued %.2f ms" % ((e[1],) + tuple(1e3 * (b - a) for a, b in zip(m, m[1:]))))
        over = [(p[1], 1e3 * (max(p[2], e[2]) - e[2]), 1e3 * (min(p[3], e[3]) - e[2])) for p in trace if p[0] == "prepare" and p[3] > e[2] and p[2] < e[3]]
        print("   prepares running meanwhile (batch, from, to ms within the launch):", over)

if "--phases" in sys.argv:
    # Renderer.prepare's phases on the worker threads while the job runs (marks tagged with their thread)
    import collections
    acc, cnt = collections.OrderedDict(), 0
    for ln_marks in phase_marks:
        by_thread = collections.defaultdict(list)
        for label, t, ident in ln_marks:
            by_thread[ident].append((label, t))
        for seq in by_thread.values():
            for (a, ta), (b, tb) in zip(seq, seq[1:]):
                if b == "start":
                    cnt += 1
                    continue
                acc[b] = acc.get(b, 0.0) + (tb - ta)
    cnt = max(cnt, 1)
    print("prepare phases on the worker threads, mean ms over %d batches: %s" % (cnt, ", ".join("%s %.2f" % (k, 1e3 * v / cnt) for k, v in acc.items())))
