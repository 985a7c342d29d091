#!/bin/bash
# every kernel of the default step ALONE on the chip, in the step's real configuration (fused warp, both streams): the HIP
# runtime serialises the launches (AMD_SERIALIZE_KERNEL=3) under a rocprofv3 kernel trace
tag=${1:-r05alone}
shift
export AMD_SERIALIZE_KERNEL=3
bash scripts/trace_stats.sh "$tag" 1024 6 "$@" 2>&1 | grep -v "at::native\|rocclr"
