#!/bin/bash
# bash tools/ab_run.sh "command" lib1 lib2 ...: run a command under each library build
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; CMD=$1; shift
cp $PKG/libsmplraster_hip.so $PKG/lib_keep.so
for v in "$@"; do cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so; echo "== $v"; bash -c "$CMD"; done
cp $PKG/lib_keep.so $PKG/libsmplraster_hip.so
