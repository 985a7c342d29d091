#!/bin/bash
# One rocprofv3 counter pass over any script of this repo; prints per-kernel averages.
# usage: scripts/pmc_script.sh <tag> <script.py> <script args, comma separated or ""> <counter> [<counter> ...]
set -e
tag=$1; script=$2; sargs=$3; shift 3
out=$PWD/gpurun_out/pmc_$tag
root=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace -d "$out" -o pmc --output-format csv -- python3 "$root/$script" ${sargs//,/ } > "$out/run.log" 2>&1
cd "$root"
python3 scripts/pmc_summary.py "$out"
