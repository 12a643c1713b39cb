#!/bin/bash
# bash tools/ab_env.sh KERNEL_SUBSTRING[,..] "BATCHES" "ENV1" "ENV2" ...: rocprofv3 average of the named kernels of the eager
# bench step under each environment setting (e.g. "SMPLR_RASTER=1" "SMPLR_RASTER=2 SMPLR_RASTER_SHAPE=3"), per batch size.
# The program after `--` is python3 itself (never env / bash -c: see the GPU box's exec rule); settings are exported here.
cd "$GRAFT_REPO_ROOT"; K=$1; BS=$2; shift; shift
i=0
for e in "$@"; do
  i=$((i+1))
  for b in $BS; do
    rm -rf gpurun_out/abe_${i}_$b
    (cd /tmp && export TMPDIR=/tmp $e && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abe_${i}_$b -- python3 $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 5 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > /dev/null 2>&1)
    python3 -c "
import csv,glob
f=glob.glob('gpurun_out/abe_${i}_$b/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in '$K'.split(',')): print('[$e] B=$b', r['Name'][:44], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
  done
done
