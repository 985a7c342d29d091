#!/bin/bash
# Evidence for profiles/ of a fixed-job run (BASELINE configs 4 / 5): the bench line and the rocprofv3 kernel-trace summary of
# the same command.  usage (GPU box, repo root): scripts/collect_job_profile.sh <tag> <config> <job_notes> <sub_batch>
set -e
tag=$1; cfg=$2; notes=$3; sub=$4
root=$PWD
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles_$tag
mkdir -p "$out" "$dst"
python3 bench.py --config $cfg --job-notes $notes --sub-batch $sub --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o r --output-format csv -- python3 "$root/bench.py" --config $cfg --job-notes $notes --sub-batch $sub --steps 2 --warmup 1 --no-cpu-baseline > "$out/trace.log" 2>&1
cd "$root"
python3 - "$out" "$dst" "$tag" <<'PY'
import glob, json, shutil, sys
out, dst, tag = sys.argv[1:4]
line = open(out + "/bench.json").read().strip().splitlines()[-1]
json.dump(json.loads(line), open(f"{dst}/{tag}_bench.json", "w"), indent=1)
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
shutil.copy(st[0], f"{dst}/{tag}_kernel_stats.csv")
d = json.loads(line)
print(tag, d["ms_per_step"], d["value"], d["realtime_factor"])
PY
