#!/bin/bash
# rocprofv3 kernel-trace statistics of a short run of the default workload (serialised dispatches: each kernel alone).
# usage (GPU box, repo root): scripts/trace_stats.sh <tag> [stage_times args...]
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/trace_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out" -o r --output-format csv -- python3 "$root/scripts/stage_times.py" "$@" > "$out/run.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-62s n=%4s avg=%8.1f us  %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
