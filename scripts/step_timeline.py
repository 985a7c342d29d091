#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 --kernel-trace CSV: every kernel's start / end relative to the step's first kernel,
the idle time of the device (no kernel running) and the dependency gaps on the critical path.

    python scripts/step_timeline.py <..._kernel_trace.csv> [step-index-from-the-end]
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", "?")))
rows.sort()
# steps start at k_sample_assemble
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sample_assemble")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0 = starts[-which - 1]
i1 = starts[-which]
step = rows[i0:i1]
t0 = step[0][0]
print("step of %d kernels, %.3f ms from first start to last end; next step starts at %.3f ms" % (
    len(step), (max(r[1] for r in step) - t0) / 1e6, (rows[i1][0] - t0) / 1e6))
for s, e, name, st in step:
    print("%8.3f %8.3f  %7.3f ms  stream %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, name[:70]))
# idle: union of busy intervals
busy, cur_s, cur_e = 0, None, None
idle = []
for s, e, *_ in step:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        idle.append((cur_e - t0, s - cur_e))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("device busy (some kernel running) %.3f ms; idle gaps inside the step: %s" % (busy / 1e6, ", ".join("%.1f us at %.3f" % (d / 1e3, a / 1e6) for a, d in idle)))
print("gap to the next step's first kernel: %.1f us" % ((rows[i1][0] - max(r[1] for r in step)) / 1e3))
