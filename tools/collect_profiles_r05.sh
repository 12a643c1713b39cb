#!/bin/bash
# Copies the summaries of tools/r05_profile.sh runs (gpurun_out/TAG, modes main / variants / big under the same or
# different tags) into profiles/ under the round prefix and rebuilds the JSON files bench.py reads (stamped with the measured
# library's build id):  bash tools/collect_profiles_r05.sh MAIN_TAG [VARIANTS_TAG] [BIG_TAG]
TAG=$1; VT=${2:-$1}; BT=${3:-$1}; P=r05; S=gpurun_out/$TAG; D=profiles
for d in $S/pmc_*/ $S/prof_*/; do
  n=$(ls $d/*/*_agent_info.csv 2>/dev/null | wc -l)
  if [ "$n" -gt 1 ]; then echo "collect: $d holds $n runs (tag reused?) - rerun under a fresh tag"; exit 1; fi
done
BID=$(python -c "import json; print(json.load(open('$S/bench.json'))['build_id'])")
export PROFILE_BUILD_ID=$BID
cp $S/bench.json $D/${P}_bench.json
cp $S/eager_kernel_stats.csv $D/${P}_eager_kernel_stats.csv
cp $S/default_kernel_stats.csv $D/${P}_default_bench_kernel_stats.csv
cp $S/pmc_step_sq.txt $D/${P}_pmc_step_sq_counters.txt
cp $S/batch_sweep.txt $D/${P}_batch_sweep.txt
cp $S/same_box.txt $D/${P}_same_box.txt
[ -f $S/step_variants.txt ] && cp $S/step_variants.txt $D/${P}_step_variants.txt
python tools/pmc_traffic.py $S/pmc_step_3 $S/pmc_step_4 $S/pmc_step_5
python tools/raster_sq.py $S/eager_kernel_stats.csv 1.45,2.62 $S/pmc_step_1 $S/pmc_step_2
python tools/kernel_table.py $S/eager_kernel_stats.csv $S/pmc_step_3 $S/pmc_step_4 $S/pmc_step_5 --batch 128 > $D/${P}_kernel_table_b128.txt
V=gpurun_out/$VT
if [ -f $V/seg_only_kernel_stats.csv ]; then
  for X in seg_only fused_loss unfused_loss both_heads silhouette_only; do
    cp $V/${X}_kernel_stats.csv $D/${P}_${X}_kernel_stats.csv
    PMC_TRAFFIC_NAME=${P}_traffic_${X}.json python tools/pmc_traffic.py $V/pmc_${X}_1 $V/pmc_${X}_2
  done
  for N in w64 vs5 vs2; do cp $V/${N}_kernel_stats.csv $D/${P}_${N}_kernel_stats.csv; done
fi
G=gpurun_out/$BT
if [ -f $G/kernel_table.txt ]; then
  cp $G/eager_kernel_stats.csv $D/${P}_b2048_kernel_stats.csv
  cp $G/kernel_table.txt $D/${P}_b2048_kernel_table.txt
  cp $G/step.txt $D/${P}_b2048_step.txt
  python tools/kernel_table.py $G/eager_kernel_stats.csv $G/pmc_1 $G/pmc_2 $G/pmc_3 --batch 2048 --json $D/${P}_traffic_b2048.json > /dev/null
  [ -f $G/record_sweep.txt ] && grep -v amdgpu.ids $G/record_sweep.txt > $D/${P}_record_sweep_final.txt
fi
ls $D | grep $P
