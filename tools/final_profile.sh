#!/bin/bash
# Runs on the GPU box: the round's reference measurements.  Output under gpurun_out/final_TAG/.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_$TAG
mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
# default bench.py command (graph launch, stage breakdown, cpu baseline)
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-160 $O/bench.json
# rocprofv3 kernel trace + stats of the same default command
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1; echo "rocprof(default) rc=$?")
# eager launch: one dispatch per kernel per step, clean per-kernel averages
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > $O/rocprof_eager.log 2>&1; echo "rocprof(eager) rc=$?")
# batch sweep (graph)
for b in 32 128 512 2048; do timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d' % $b, l['value'], l['ms_per_step'])"; done > $O/batch_sweep.txt; cat $O/batch_sweep.txt
