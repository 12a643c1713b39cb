#!/bin/bash
# bash tools/pmc_variants.sh v1 v2 ...: time + SQ counters of the silhouette forward under each library build
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd
cp $PKG/libsmplraster_hip.so $PKG/lib_keep.so
for v in "$@"; do
  cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so
  timeout -k 10 200 python3 tools/probes/silh_hash.py > /dev/null 2> gpurun_out/ab/silh_time_$v.txt; echo "$v: $(tail -1 gpurun_out/ab/silh_time_$v.txt)"
  bash tools/pmc_silh.sh $v | grep -A17 "silh_px\|silh_fused" | grep -E "kernel|INSTS_VALU|INSTS_SALU|INSTS_LDS|WAVE_CYCLES|WAIT_ANY|ACTIVE_INST_VALU|BANK"
done
cp $PKG/lib_keep.so $PKG/libsmplraster_hip.so
