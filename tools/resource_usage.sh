#!/bin/bash
# VGPRs / spills / occupancy of the kernels of one csrc/*.hip file whose names match a pattern (hipcc remarks; no GPU).
#   tools/resource_usage.sh raster.hip seg_bwd_kernel
cd "$(dirname "$0")/../indirect_learning_pose-shape_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function $EXTRA -c "$1" -o /tmp/ru_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Spill: [1-9]|Occupancy|LDS Size" | \
  awk -v pat="$2" '/error/ {print} /Function Name/ {show = ($0 ~ pat); if (show) {sub(/.*Function Name: /, ""); sub(/ \[.*/, ""); printf "%s\n", $0}} !/Function Name/ && show {sub(/.*remark: +/, "   "); sub(/ \[.*/, ""); print}'
rm -f /tmp/ru_$$.o
