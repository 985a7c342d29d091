#!/bin/bash
# config 5 (96 kHz / 2048 / 96): kernels alone (serialised launches) + the job's bench line
tag=${1:-r05c5}; shift
mkdir -p gpurun_out/$tag
root=$PWD
args="--config 5 --job-notes 1024 --sub-batch 4096 --no-cpu-baseline --no-variants --no-host-inclusive"
python3 bench.py $args --steps 5 --warmup 2 "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
d=json.loads([x for x in open("gpurun_out/$tag/bench.json") if x.startswith("{")][-1])
print("value %.2f M frames/s, %.2f ms per pass" % (d["value"]/1e6, d["ms_per_step"]))
print({k: round(v,3) for k,v in d["stage_ms"].items() if v>0})
print("roofline:", d["roofline"]["kernel"], d["roofline"]["bound"], round(d["roofline"]["frac"],3))
PY
cd /tmp && export TMPDIR=/tmp
AMD_SERIALIZE_KERNEL=3 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$tag/alone -o r --output-format csv -- python3 $root/bench.py $args --steps 2 --warmup 1 "$@" > $root/gpurun_out/$tag/alone.log 2>&1
cd $root
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/alone/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-70s n=%4s avg=%9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
