#!/bin/bash
# bash tools/ab_kernel.sh KERNEL_SUBSTRING[,SUBSTRING...] lib1 lib2 ...: rocprofv3 average of the named kernels of the eager bench step
# under each library build (indirect_learning_pose-shape_amd/lib_<name>.so; "keep" = the product library), chosen through
# SMPLR_LIB_PATH - the product library is never overwritten.  BENCH_ARGS: extra bench.py arguments.
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; K=$1; shift
for v in "$@"; do
  rm -rf gpurun_out/abk_$v
  L=$GRAFT_REPO_ROOT/$PKG/lib_$v.so; [ "$v" = keep ] && L=
  (cd /tmp && export TMPDIR=/tmp SMPLR_LIB_PATH=$L && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abk_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg $BENCH_ARGS > /dev/null 2>&1)
  python3 -c "
import csv,glob
f=glob.glob('gpurun_out/abk_$v/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in '$K'.split(',')): print('$v', r['Name'][:40], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
done
