#!/bin/bash
# Runs on the GPU box: for each value of an env var, parity tests of the touched kernels + rocprof per-kernel averages.
# usage: variant_check.sh ENVVAR "v1 v2 ..." [pytest -k expr]
VAR=$1; VALS=$2; KEXPR=${3:-blend}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for v in $VALS; do
  export $VAR=$v
  timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $R/gpurun_out/pytest_$v.log 2>&1; echo "$VAR=$v pytest rc=$?"; tail -1 $R/gpurun_out/pytest_$v.log
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_var_$v -- python3 $R/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > $R/gpurun_out/rocprof_$v.log 2>&1; echo "rocprof rc=$?")
  python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_var_$v/*/*_kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=25:
        tot+=float(r['AverageNs'])/1e3
        print("  %-44s %8.1f" % (r['Name'][:44], float(r['AverageNs'])/1e3))
print("  sum %.1f us" % tot)
PY
  tail -1 $R/gpurun_out/rocprof_$v.log | cut -c1-130
done
