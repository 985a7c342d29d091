#!/bin/bash
# round 6 loop: the whole GPU suite (stop at the first failure), then every kernel of the default step alone and the step's stage times
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 ${2:-700} python -m pytest tests -m gpu -q -x > "$out/pytest.log" 2>&1
rc=$?
tail -3 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
bash scripts/r05_alone.sh ${tag}_alone > "$out/alone.txt" 2>&1; head -16 "$out/alone.txt"
python scripts/stage_times.py 1024 20 > "$out/stages.txt"; cat "$out/stages.txt"
