#!/bin/bash
# round-5 iteration loop on the GPU box: full GPU suite, value_f64 A/B (bit / error / stage times), bench line
tag=${1:-r05a}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > "$out/pytest.log" 2>&1
rc=$?
tail -5 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/ab_option.py value_f64 1 0 > "$out/ab_value_f64.txt" 2>&1 && cat "$out/ab_value_f64.txt" && \
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; tail -c 3000 "$out/bench.json"
