#!/bin/bash
# A/B of two library builds on one box for the silhouette rasteriser: bit-exactness (hashes) + timing.
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; mkdir -p gpurun_out/ab
for v in old new; do
  cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so
  timeout -k 10 200 python3 tools/probes/silh_hash.py > gpurun_out/ab/silh_hash_$v.txt 2> gpurun_out/ab/silh_time_$v.txt || { tail -20 gpurun_out/ab/silh_time_$v.txt; exit 1; }
  tail -1 gpurun_out/ab/silh_time_$v.txt
done
if diff gpurun_out/ab/silh_hash_old.txt gpurun_out/ab/silh_hash_new.txt; then echo "silhouette hashes old == new ($(wc -l < gpurun_out/ab/silh_hash_new.txt) cases)"; else echo "SILHOUETTE HASHES DIFFER"; fi
cp $PKG/lib_new.so $PKG/libsmplraster_hip.so
