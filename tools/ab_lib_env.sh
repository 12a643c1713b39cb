#!/bin/bash
# bash tools/ab_lib_env.sh KERNEL "BATCHES" LIB "ENV1" "ENV2" ...: tools/ab_env.sh under the library build lib_LIB.so
# (through SMPLR_LIB_PATH: the product library stays where it is)
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; K=$1; BS=$2; LIB=$3; shift; shift; shift
echo "== library $LIB"; SMPLR_LIB_PATH=$GRAFT_REPO_ROOT/$PKG/lib_$LIB.so bash tools/ab_env.sh "$K" "$BS" "$@"
