#!/bin/bash
# one case of the keyword soak under a given build of the library: scripts/soak_case.sh <lib.so> <case>
GOOFER_HIP_LIB=$PWD/$1 GOOFER_FUZZ_FIRST=$2 GOOFER_FUZZ_CASES=4 python -m pytest tests/test_gpu_synth.py -m gpu -q -x -k "random_kwargs and $2" 2>&1 | grep -v "^$" | tail -4
