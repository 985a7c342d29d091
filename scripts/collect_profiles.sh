#!/bin/bash
# Evidence for profiles/: the bench line, a rocprofv3 kernel-trace summary of the same command, and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE — separate passes, no trace flags beside --kernel-trace) behind roofline.traffic.
# The traced / counted runs leave out the variants (skip_zero off, 30 % unvoiced) and the host-inclusive leg, so that the per-kernel
# averages are those of the default step.
# usage (GPU box, repo root): scripts/collect_profiles.sh <tag>
set -e
tag=$1
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json" | head -c 600; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r --output-format csv -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-host-inclusive --no-legs > "$out/trace.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/fetch" -o r --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-host-inclusive --no-legs > "$out/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/write" -o r --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-host-inclusive --no-legs > "$out/write.log" 2>&1
echo "write done"
cd "$root"
python3 scripts/pmc_traffic_json.py "$out" "$tag"
