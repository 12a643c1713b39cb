#!/bin/bash
# Runs on the GPU box (ONE gpurun call = one lease = one box): round 5's reference measurements -> gpurun_out/$TAG/
#   bash tools/r05_profile.sh TAG [main|variants|big]
# main (default): the default bench line, the eager kernel trace AND the headline step replayed from its graph, all on this
#   box (same_box.txt: sum of the seven kernels' averages against the step - VERDICT r04 #7), PMC passes (SQ x2, FETCH, WRITE,
#   TCC - each set in its own pass, kernel trace off), batch sweep, step variants' timings.
# variants: FETCH / WRITE passes + eager trace of the step variants, W = 64 / vs traces.
# big: B = 2048 evidence (tools/r05_b2048.sh) + the record sweep (tools/record_sweep.py).
TAG=${1:-r05}; MODE=${2:-main}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R; export TMPDIR=/tmp
EAGER="--steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg"
if [ "$MODE" = main ]; then
  timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py $EAGER > $O/rocprof_eager.log 2>&1; echo "rocprof(eager) rc=$?")
  cp $(ls $O/prof_eager/*/*_kernel_stats.csv | head -1) $O/eager_kernel_stats.csv 2>/dev/null
  # the same box's headline right after the trace: three short runs of the graph-replayed step
  for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print(l['ms_per_step'], w['min'], w['median'], w['max'], l['ms_per_step_graph1'], l['build_id'][:12])"; done > $O/headline_runs.txt
  python tools/same_box.py $O/eager_kernel_stats.csv $O/headline_runs.txt $O/bench.json > $O/same_box.txt
  # ... and the same kernels INSIDE the replayed graph (durations and the gap to the next dispatch), same box
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_graph -- python3 $R/bench.py --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg > $O/rocprof_graph.log 2>&1; echo "rocprof(graph) rc=$?")
  { echo "# the step's kernels inside the replayed HIP graph (last 2000 dispatches of a rocprofv3 --kernel-trace of the headline command, tools/graph_gaps.py):"; python tools/graph_gaps.py $O/prof_graph 2000; } >> $O/same_box.txt
  cat $O/same_box.txt
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1; echo "rocprof(default) rc=$?")
  cp $(ls $O/prof_default/*/*_kernel_stats.csv | head -1) $O/default_kernel_stats.csv 2>/dev/null
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_step_$i -- python3 $R/bench.py --steps 3 --warmup 2 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/pmc_step_$i.log 2>&1; echo "pmc step pass $i rc=$?") || exit 1
  done
  python tools/pmc_summary.py $O/pmc_step_1 $O/pmc_step_2 > $O/pmc_step_sq.txt
  for b in 32 128 512 2048; do timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d' % $b, l['value'], l['ms_per_step'])"; done > $O/batch_sweep.txt; cat $O/batch_sweep.txt
  for V in default seg_only fused_loss unfused_loss both_heads silhouette_only; do timeout -k 10 200 python bench.py --step $V --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print('$V', l['value'], l['ms_per_step'], w['min'], w['median'], w['max'])"; done > $O/step_variants.txt; cat $O/step_variants.txt
fi
if [ "$MODE" = variants ]; then
  for V in seg_only fused_loss unfused_loss both_heads silhouette_only; do
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --step $V $EAGER > $O/rocprof_$V.log 2>&1; echo "rocprof($V) rc=$?") || exit 1
    cp $(ls $O/prof_$V/*/*_kernel_stats.csv | head -1) $O/${V}_kernel_stats.csv 2>/dev/null
    j=0
    for SET in "FETCH_SIZE" "WRITE_SIZE"; do
      j=$((j+1))
      (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_${V}_$j -- python3 $R/bench.py --step $V --steps 3 --warmup 2 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/pmc_${V}_$j.log 2>&1; echo "pmc $V pass $j rc=$?") || exit 1
    done
  done
  for CFG in "w64 --wh 64" "vs5 --vertex-sampling 5" "vs2 --vertex-sampling 2"; do
    set -- $CFG; N=$1; shift
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -- python3 $R/bench.py "$@" $EAGER > $O/rocprof_$N.log 2>&1; echo "rocprof($N) rc=$?") || exit 1
    cp $(ls $O/prof_$N/*/*_kernel_stats.csv | head -1) $O/${N}_kernel_stats.csv 2>/dev/null
  done
fi
if [ "$MODE" = big ]; then
  bash tools/r05_b2048.sh $TAG 2048
  python tools/record_sweep.py > $O/record_sweep.txt 2>&1; grep -v amdgpu.ids $O/record_sweep.txt
fi
find $O -name "*.db" -delete 2>/dev/null; du -sh $O
