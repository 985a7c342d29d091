#!/bin/bash
# kernel trace of the default step: timeline of one step (two streams) and the kernels alone (--serial)
tag=${1:-r05t}
shift
bash scripts/trace_stats.sh "$tag" 1024 6 "$@" > gpurun_out/${tag}_kernels.txt 2>&1 || { tail gpurun_out/${tag}_kernels.txt; exit 1; }
f=$(find gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/step_timeline.py "$f" > gpurun_out/${tag}_timeline.txt
bash scripts/trace_stats.sh "${tag}s" 1024 6 --serial "$@" > gpurun_out/${tag}_serial.txt 2>&1
cat gpurun_out/${tag}_timeline.txt; grep -v "at::native\|rocclr" gpurun_out/${tag}_serial.txt
