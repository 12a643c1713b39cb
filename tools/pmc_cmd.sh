#!/bin/bash
# Like pmc.sh but for an arbitrary python script: bash tools/pmc_cmd.sh TAG "script.py args" "COUNTERS..." ["COUNTERS..."]
TAG=$1; CMD=$2; shift; shift
R=$GRAFT_REPO_ROOT
i=0
for SET in "$@"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/$CMD > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1; echo "pmc pass $i rc=$?") || exit 1
done
