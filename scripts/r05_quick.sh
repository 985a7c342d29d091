#!/bin/bash
# quick loop: the assembly-sensitive GPU tests, then the kernels alone and the step
tag=${1:-r05q}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_kernels.py tests/test_gpu_synth.py tests/test_gpu_fullsize.py tests/test_gpu_jobmode.py -m gpu -q -x > "$out/pytest.log" 2>&1
rc=$?
tail -3 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
bash scripts/r05_alone.sh ${tag}_alone > "$out/alone.txt" 2>&1; head -12 "$out/alone.txt"
python scripts/stage_times.py 1024 20 | tail -1
