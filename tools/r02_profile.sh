#!/bin/bash
# Runs on the GPU box (one gpurun call): the round's reference measurements -> gpurun_out/$TAG/
#   bash tools/r02_profile.sh TAG [quick]
# default bench line, rocprofv3 kernel traces (default + eager), VALU issue probe, PMC passes (SQ, FETCH, WRITE, TCC) of
# the eager step and of the silhouette rasteriser.  "quick": bench + eager trace only.
TAG=${1:-r02}; MODE=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > $O/rocprof_eager.log 2>&1; echo "rocprof(eager) rc=$?")
cp $(ls $O/prof_eager/*/*_kernel_stats.csv | head -1) $O/eager_kernel_stats.csv 2>/dev/null
[ "$MODE" = quick ] && exit 0
[ -x tools/probes/bin/valu_issue_probe ] && (timeout -k 10 120 tools/probes/bin/valu_issue_probe > $O/valu_issue_probe.txt 2>&1; echo "probe rc=$?"; cat $O/valu_issue_probe.txt)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/rocprof_default.log 2>&1; echo "rocprof(default) rc=$?")
cp $(ls $O/prof_default/*/*_kernel_stats.csv | head -1) $O/default_kernel_stats.csv 2>/dev/null
# PMC: each set in its own pass, kernel-trace off
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_step_$i -- python3 $R/bench.py --steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-breakdown > $O/pmc_step_$i.log 2>&1; echo "pmc step pass $i rc=$?") || exit 1
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_silh_$i -- python3 $R/tools/silh_time.py > $O/pmc_silh_$i.log 2>&1; echo "pmc silh pass $i rc=$?") || exit 1
done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_silh -- python3 $R/tools/silh_time.py > $O/rocprof_silh.log 2>&1; echo "rocprof(silh) rc=$?")
cp $(ls $O/prof_silh/*/*_kernel_stats.csv | head -1) $O/silh_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $O/pmc_step_1 $O/pmc_step_2 > $O/pmc_step_sq.txt
python tools/pmc_summary.py $O/pmc_silh_1 $O/pmc_silh_2 $O/pmc_silh_3 $O/pmc_silh_4 $O/pmc_silh_5 silh > $O/pmc_silh.txt
for b in 32 128 512 2048; do timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d' % $b, l['value'], l['ms_per_step'])"; done > $O/batch_sweep.txt; cat $O/batch_sweep.txt
# keep only the CSVs (the rocprofv3 output dirs also hold large databases)
find $O -name "*.db" -delete 2>/dev/null; du -sh $O
