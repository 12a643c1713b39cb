#!/bin/bash
# Runs on the GPU box: SQ / memory PMC counters for the bench step (eager), one rocprofv3 pass per counter set.
# Usage: bash tools/pmc.sh TAG "SQ_WAVES SQ_INSTS_VALU ..." ["second set" ...]   -> gpurun_out/pmc_TAG_<i>/
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for SET in "$@"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-breakdown > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1; echo "pmc pass $i rc=$?") || exit 1
done
