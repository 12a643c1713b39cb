#!/bin/bash
# bash tools/ab_lib_env.sh KERNEL "BATCHES" LIB "ENV1" "ENV2" ...: tools/ab_env.sh under the library build lib_LIB.so
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; K=$1; BS=$2; LIB=$3; shift; shift; shift
cp $PKG/libsmplraster_hip.so $PKG/lib_keep.so; cp $PKG/lib_$LIB.so $PKG/libsmplraster_hip.so
echo "== library $LIB"; bash tools/ab_env.sh "$K" "$BS" "$@"
cp $PKG/lib_keep.so $PKG/libsmplraster_hip.so
