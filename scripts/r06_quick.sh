#!/bin/bash
# round 6 quick loop: the named test files (stop at the first failure), then every kernel of the default step alone and the stage times
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -m pytest "$@" -m gpu -q -x > "$out/pytest.log" 2>&1
rc=$?
tail -4 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
bash scripts/r05_alone.sh ${tag}_alone > "$out/alone.txt" 2>&1; head -16 "$out/alone.txt"
python scripts/stage_times.py 1024 20 > "$out/stages.txt"; cat "$out/stages.txt"
