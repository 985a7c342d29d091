#!/bin/bash
# the plain step of a workload under two builds of the library, alternating processes on one box
# usage: scripts/lib_ab.sh <libA.so> <libB.so> [rounds] [notes] [config]
A=$1; B=$2; R=${3:-3}; N=${4:-1024}; C=${5:-3}
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    GOOFER_HIP_LIB=$PWD/$L python - $N $C <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload
ctx = Context(0)
wl = SamplerWorkload(ctx, int(sys.argv[2]), list(range(int(sys.argv[1]))))
best = 1e9
for rep in range(4):
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(30):
        wl.step()
    t1.record()
    torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1) / 30)
print(os.path.basename(os.environ["GOOFER_HIP_LIB"]), "config %s, %s notes: %.3f ms" % (sys.argv[2], sys.argv[1], best))
PY
  done
done
