#!/bin/bash
# bash tools/ab_step_env.sh "BENCH ARGS" "ENV1" "ENV2" ...: the graph-replayed step (bench.py's headline method, no aux legs) under
# each environment setting, twice each, interleaved (same box, minutes apart): ms per step min / median / max of ten windows
cd "$GRAFT_REPO_ROOT"; A=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    (export $e; timeout -k 10 200 python bench.py $A --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']
print('[$e] rep $rep: %.1f meshes/s  ms %.4f  windows %.4f / %.4f / %.4f  graph1 %s' % (l['value'], l['ms_per_step'], w['min'], w['median'], w['max'], l.get('ms_per_step_graph1')))")
  done
done
