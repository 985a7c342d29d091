#!/bin/bash
# One rocprofv3 counter pass over a short run of the default workload; prints per-kernel averages.
# usage: scripts/pmc_pass.sh <tag> <counter> [<counter> ...]      (on the GPU box, from the repo root)
set -e
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
root=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace -d "$out" -o pmc --output-format csv -- python3 "$root/scripts/stage_times.py" 1024 2 > "$out/run.log" 2>&1
cd "$root"
python3 scripts/pmc_summary.py "$out"
