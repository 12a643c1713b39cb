#!/bin/bash
# Runs on the GPU box (one gpurun call): round 4's reference measurements -> gpurun_out/$TAG/
#   bash tools/r04_profile.sh TAG [quick|traffic|main|variants]   ("main" then "variants" under the SAME tag = everything, in
#   two gpurun calls of < 20 minutes each; files merge in gpurun_out/TAG)
# default bench line, rocprofv3 kernel traces (default command + eager, one dispatch per kernel per step), PMC passes
# (SQ x2, FETCH, WRITE, TCC - each set in its own pass, kernel trace off) of the eager default step, and FETCH / WRITE
# passes + an eager kernel trace of the step VARIANTS (bench.py --step seg_only | fused_loss | unfused_loss | both_heads |
# silhouette_only).  "quick": bench + eager trace only.  "traffic": skip the SQ passes and the batch sweep.
TAG=${1:-r04}; MODE=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "$MODE" = variants ]; then SKIP_MAIN=1; fi
if [ -z "$SKIP_MAIN" ]; then
if [ "$MODE" != traffic ]; then
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
fi
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/rocprof_eager.log 2>&1; echo "rocprof(eager) rc=$?")
cp $(ls $O/prof_eager/*/*_kernel_stats.csv | head -1) $O/eager_kernel_stats.csv 2>/dev/null
[ "$MODE" = quick ] && exit 0
if [ "$MODE" != traffic ]; then
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1; echo "rocprof(default) rc=$?")
cp $(ls $O/prof_default/*/*_kernel_stats.csv | head -1) $O/default_kernel_stats.csv 2>/dev/null
fi
# PMC: each set in its own pass, kernel-trace off
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ "$MODE" = traffic ] && [ $i -le 2 ]; then continue; fi
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_step_$i -- python3 $R/bench.py --steps 3 --warmup 2 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/pmc_step_$i.log 2>&1; echo "pmc step pass $i rc=$?") || exit 1
done
fi   # SKIP_MAIN
VARIANTS="seg_only fused_loss unfused_loss both_heads silhouette_only"
[ "$MODE" = main ] && VARIANTS=""
for V in $VARIANTS; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -- python3 $R/bench.py --step $V --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/rocprof_$V.log 2>&1; echo "rocprof($V) rc=$?") || exit 1
  cp $(ls $O/prof_$V/*/*_kernel_stats.csv | head -1) $O/${V}_kernel_stats.csv 2>/dev/null
  j=0
  for SET in "FETCH_SIZE" "WRITE_SIZE"; do
    j=$((j+1))
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_${V}_$j -- python3 $R/bench.py --step $V --steps 3 --warmup 2 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/pmc_${V}_$j.log 2>&1; echo "pmc $V pass $j rc=$?") || exit 1
  done
done
[ "$MODE" = variants ] && { find $O -name "*.db" -delete 2>/dev/null; du -sh $O; exit 0; }
# the reference's other shipped sizes: W = 64 (train.py:320-321, predict.py:129-136) and
# vertex_sampling 5 / 2 at W = 48 (profiling_renderer.py:26) - eager kernel trace of the same step
for CFG in "w64 --wh 64" "vs5 --vertex-sampling 5" "vs2 --vertex-sampling 2"; do
  set -- $CFG; N=$1; shift
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -- python3 $R/bench.py "$@" --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $O/rocprof_$N.log 2>&1; echo "rocprof($N) rc=$?") || exit 1
  cp $(ls $O/prof_$N/*/*_kernel_stats.csv | head -1) $O/${N}_kernel_stats.csv 2>/dev/null
done
if [ "$MODE" != traffic ]; then
python tools/pmc_summary.py $O/pmc_step_1 $O/pmc_step_2 > $O/pmc_step_sq.txt
for b in 32 128 512 2048; do timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d' % $b, l['value'], l['ms_per_step'])"; done > $O/batch_sweep.txt; cat $O/batch_sweep.txt
# the step variants with the headline's timing method: variant, meshes/s, ms per step (first window), min / median / max of ten windows
for V in default seg_only fused_loss unfused_loss both_heads silhouette_only; do timeout -k 10 200 python bench.py --step $V --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print('$V', l['value'], l['ms_per_step'], w['min'], w['median'], w['max'])"; done > $O/step_variants.txt; cat $O/step_variants.txt
fi
# keep only the CSVs (the rocprofv3 output dirs also hold large databases)
find $O -name "*.db" -delete 2>/dev/null; du -sh $O
