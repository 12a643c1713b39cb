#!/bin/bash
# Runs on the GPU box: the same PMC counter sets under two or more builds of the library (lib_<name>.so in the package
# directory), one rocprofv3 pass per (build, set); prints the per-launch average of every counter for kernels matching FILT.
#   LIBS="old new" BENCH_ARGS="--step fused_loss" bash tools/pmc_ab.sh FILT "SET 1" ["SET 2" ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PKG=indirect_learning_pose-shape_amd; FILT=$1; shift
# (libraries through SMPLR_LIB_PATH - "keep" = the product library, which is never overwritten)
for v in ${LIBS:-keep}; do
  L=$GRAFT_REPO_ROOT/$PKG/lib_$v.so; [ "$v" = keep ] && L=
  export SMPLR_LIB_PATH=$L
  i=0
  for SET in "$@"; do
    i=$((i+1)); D=gpurun_out/pmcab_${v}_$i; rm -rf $D
    timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg --min-warmup 0 $BENCH_ARGS > $D.log 2>&1 || { tail -20 $D.log; exit 1; }
    python3 - "$v" "$D" "$FILT" <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[3], r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-6s %-28s %14.0f  (n=%d)" % (sys.argv[1], k, sum(v) / len(v), len(v)))
PY
  done
done
