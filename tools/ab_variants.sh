#!/bin/bash
# times tools/probes/silh_hash.py (its stderr line) under several library builds: bash tools/ab_variants.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; mkdir -p gpurun_out/ab
cp $PKG/libsmplraster_hip.so $PKG/lib_keep.so
for v in "$@"; do
  cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so
  timeout -k 10 200 python3 tools/probes/silh_hash.py > gpurun_out/ab/silh_hash_$v.txt 2> gpurun_out/ab/silh_time_$v.txt; echo "$v: $(tail -1 gpurun_out/ab/silh_time_$v.txt)"
done
cp $PKG/lib_keep.so $PKG/libsmplraster_hip.so
