#!/bin/bash
# bash tools/ab_kernel_b.sh KERNEL_SUBSTRING[,..] "BATCHES" lib1 lib2 ...: like ab_kernel.sh at several batch sizes (eager bench
# step); libraries through SMPLR_LIB_PATH ("keep" = the product library).  BENCH_ARGS: extra bench.py arguments.
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; K=$1; BS=$2; shift; shift
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/$PKG/lib_$v.so; [ "$v" = keep ] && L=
  for b in $BS; do
    rm -rf gpurun_out/abk_${v}_$b
    (cd /tmp && export TMPDIR=/tmp SMPLR_LIB_PATH=$L && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abk_${v}_$b -- python3 $GRAFT_REPO_ROOT/bench.py --batch $b --steps 20 --warmup 5 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg $BENCH_ARGS > /dev/null 2>&1)
    python3 -c "
import csv,glob
f=glob.glob('gpurun_out/abk_${v}_$b/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in '$K'.split(',')): print('$v B=$b', r['Name'][:40], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
"
  done
done
