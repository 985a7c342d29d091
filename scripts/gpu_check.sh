#!/bin/bash
# GPU-box loop of one kernel iteration: tests (stop at the first failure), then the kernel timeline and the step time.
# usage: scripts/gpu_check.sh <tag> [pytest targets...]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
targets=${@:-tests/test_gpu_synth.py tests/test_gpu_sampler.py tests/test_gpu_fullsize.py}
timeout -k 10 420 python -m pytest $targets -m gpu -q -x > "$out/pytest.log" 2>&1
rc=$?
tail -3 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
bash scripts/trace_stats.sh "$tag" 1024 5 > "$out/kernels.txt" && python scripts/timeline.py gpurun_out/trace_"$tag" && python scripts/stage_times.py 1024 20 | tail -1
