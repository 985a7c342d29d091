#!/bin/bash
# kernels alone + step time (no tests)
tag=${1:-r05tm}; shift
mkdir -p gpurun_out/$tag
bash scripts/r05_alone.sh ${tag}_alone "$@" > gpurun_out/$tag/alone.txt 2>&1; head -9 gpurun_out/$tag/alone.txt
python scripts/stage_times.py 1024 20 "$@" | tail -1
