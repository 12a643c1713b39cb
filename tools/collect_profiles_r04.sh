#!/bin/bash
# Copies the summaries of one tools/r04_profile.sh run (gpurun_out/TAG) into profiles/ under the round prefix and
# rebuilds the JSON files bench.py reads (stamped with the measured library's build id, taken from the run's bench
# line):  bash tools/collect_profiles_r04.sh TAG [PREFIX]      (e.g. r03a r03)
TAG=$1; P=${2:-r04}; S=gpurun_out/$TAG; D=profiles
# a tag used twice leaves two runs' files side by side in the merged directory: refuse to mix them
for d in $S/pmc_*/ $S/prof_*/; do
  n=$(ls $d/*/*_agent_info.csv 2>/dev/null | wc -l)
  if [ "$n" -gt 1 ]; then echo "collect: $d holds $n runs (tag reused?) - rerun tools/r04_profile.sh under a fresh tag"; exit 1; fi
done
BID=$(python -c "import json; print(json.load(open('$S/bench.json'))['build_id'])")
export PROFILE_BUILD_ID=$BID
cp $S/bench.json $D/${P}_bench.json
cp $S/eager_kernel_stats.csv $D/${P}_eager_kernel_stats.csv
cp $S/default_kernel_stats.csv $D/${P}_default_bench_kernel_stats.csv
cp $S/pmc_step_sq.txt $D/${P}_pmc_step_sq_counters.txt
cp $S/batch_sweep.txt $D/${P}_batch_sweep.txt
[ -f $S/step_variants.txt ] && cp $S/step_variants.txt $D/${P}_step_variants.txt
for V in seg_only fused_loss unfused_loss both_heads silhouette_only; do
  cp $S/${V}_kernel_stats.csv $D/${P}_${V}_kernel_stats.csv
  PMC_TRAFFIC_NAME=${P}_traffic_${V}.json python tools/pmc_traffic.py $S/pmc_${V}_1 $S/pmc_${V}_2
done
python tools/pmc_traffic.py $S/pmc_step_3 $S/pmc_step_4 $S/pmc_step_5
for N in w64 vs5 vs2; do cp $S/${N}_kernel_stats.csv $D/${P}_${N}_kernel_stats.csv; done
# issue costs at the kernel's occupancy (8 waves per SIMD): profiles/r04_valu_issue_probe.txt
python tools/raster_sq.py $S/eager_kernel_stats.csv 1.45,2.62 $S/pmc_step_1 $S/pmc_step_2
ls $D
