#!/bin/bash
# kernel timeline of a plain step (no profile events) + the dependency gaps on the caller's stream
tag=${1:-r05gaps}; shift
root=$PWD
out=$root/gpurun_out/trace_$tag
mkdir -p "$out"
python3 scripts/plain_steps.py 1024 30 3 "$@"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out" -o r --output-format csv -- python3 "$root/scripts/plain_steps.py" 1024 12 3 "$@" > "$out/run.log" 2>&1
cd "$root"
python3 scripts/step_timeline.py "$(find $out -name '*kernel_trace.csv' | head -1)" 4
