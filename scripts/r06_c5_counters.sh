#!/bin/bash
# config 5 (96 kHz / n_fft 2048 / hop 96, 1024 notes): kernels alone + two SQ counter passes (issue mix; active / wait split)
tag=${1:-r06c5}
out=gpurun_out/$tag
mkdir -p "$out"
AMD_SERIALIZE_KERNEL=3 bash scripts/trace_stats.sh ${tag}_alone 1024 3 5 2>&1 | grep -v "at::native\|rocclr" > "$out/alone.txt"; head -14 "$out/alone.txt"
bash scripts/pmc_script.sh ${tag}_sq1 scripts/stage_times.py 1024,2,5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM > "$out/sq1.txt"
bash scripts/pmc_script.sh ${tag}_sq2 scripts/stage_times.py 1024,2,5 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS > "$out/sq2.txt"
grep "k_irfft_ola1\|k_harm_shape\|k_noise_spectra\|k_rfft_frames\|k_env_edit" "$out/sq1.txt" "$out/sq2.txt"
