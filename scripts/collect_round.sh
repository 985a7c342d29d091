#!/bin/bash
# Everything profiles/ holds for one milestone, from one GPU box: the default bench line + kernel stats + FETCH / WRITE passes
# (collect_profiles.sh), two SQ counter passes of the default step, and for the fixed jobs of BASELINE configs 4 and 5 the
# bench line, kernel stats, FETCH / WRITE passes and one SQ pass.  Counter passes run alone (--kernel-trace beside them only).
# usage (GPU box, repo root): scripts/collect_round.sh <tag>        -> gpurun_out/profiles_<tag>/
set -e
tag=$1
root=$PWD
dst=$root/gpurun_out/profiles_$tag
mkdir -p "$dst"
bash scripts/collect_profiles.sh "$tag"
{
  bash scripts/pmc_pass.sh "${tag}_sq1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM
  bash scripts/pmc_pass.sh "${tag}_sq2" SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS | grep -v "^# csrc"
} > "$dst/${tag}_sq_counters.txt"
echo "sq done"
for cfg in 4 5; do
  if [ $cfg = 4 ]; then notes=10000; else notes=1024; fi
  jt=${tag}_c${cfg}
  out=$root/gpurun_out/prof_$jt
  mkdir -p "$out"
  args="--config $cfg --job-notes $notes --sub-batch 4096 --no-cpu-baseline"
  python3 bench.py $args --steps 10 --warmup 4 > "$out/bench.json" 2> "$out/bench.err"
  cd /tmp && export TMPDIR=/tmp
  # (traced / counted runs without the two-in-flight variant: its overlapped launches would sit in the per-kernel averages)
  rocprofv3 --kernel-trace --stats -d "$out/trace" -o r --output-format csv -- python3 "$root/bench.py" $args --no-variants --steps 2 --warmup 1 > "$out/trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/fetch" -o r --output-format csv -- python3 "$root/bench.py" $args --no-variants --steps 1 --warmup 1 > "$out/fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/write" -o r --output-format csv -- python3 "$root/bench.py" $args --no-variants --steps 1 --warmup 1 > "$out/write.log" 2>&1
  cd "$root"
  python3 scripts/pmc_traffic_json.py "$out" "$jt"
  cp gpurun_out/profiles_$jt/* "$dst/"
  bash scripts/pmc_script.sh "${jt}_sq" bench.py "${args// /,},--no-variants,--steps,1,--warmup,1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY > "$dst/${jt}_sq_counters.txt"
  echo "config $cfg done"
done
cp gpurun_out/profiles_$tag/* "$dst/" 2>/dev/null || true
ls "$dst"
