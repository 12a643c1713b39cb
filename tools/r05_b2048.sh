#!/bin/bash
# Runs on the GPU box: the evidence for the LARGE batch (every committed counter pass was B = 128).
#   bash tools/r05_b2048.sh TAG [BATCH]   -> gpurun_out/TAG/: eager kernel trace, FETCH / WRITE / TCC passes and the
#   graph-replayed step time at --batch BATCH (default 2048), all in ONE lease
TAG=${1:-r05_b2048}; B=${2:-2048}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R; export TMPDIR=/tmp
ARGS="--batch $B --no-cpu-baseline --no-breakdown --no-train-leg"
for i in 1 2; do timeout -k 10 200 python bench.py $ARGS --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print('B=$B', l['value'], l['ms_per_step'], w['min'], w['median'], w['max'], l['build_id'][:12])"; done > $O/step.txt; cat $O/step.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py $ARGS --steps 20 --warmup 5 --min-warmup 20 --mode eager > $O/rocprof_eager.log 2>&1; echo "rocprof(eager) rc=$?")
cp $(ls $O/prof_eager/*/*_kernel_stats.csv | head -1) $O/eager_kernel_stats.csv 2>/dev/null
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$i -- python3 $R/bench.py $ARGS --steps 3 --warmup 2 --min-warmup 4 --mode eager > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc=$?") || exit 1
done
find $O -name "*.db" -delete 2>/dev/null
python tools/kernel_table.py $O/eager_kernel_stats.csv $O/pmc_1 $O/pmc_2 $O/pmc_3 --batch $B > $O/kernel_table.txt; cat $O/kernel_table.txt
