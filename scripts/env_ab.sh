#!/bin/bash
# the plain step of the default workload under several values of an environment variable, alternating processes on one box
# usage: scripts/env_ab.sh VAR "v1 v2 ..." [rounds] [notes] [config]
V=$1; VALS=$2; R=${3:-3}; N=${4:-1024}; C=${5:-3}
for r in $(seq 1 $R); do
  for x in $VALS; do
    env $V=$x python - $V $x $N $C <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from goofer_amd.device import Context
from goofer_amd.workload import SamplerWorkload
ctx = Context(0)
wl = SamplerWorkload(ctx, int(sys.argv[4]), list(range(int(sys.argv[3]))))
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
print("%s=%s %.3f ms" % (sys.argv[1], sys.argv[2], best))
PY
  done
done
