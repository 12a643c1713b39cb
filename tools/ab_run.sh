#!/bin/bash
# bash tools/ab_run.sh "command" lib1 lib2 ...: run a command under each library build ("keep" = the product library),
# chosen through SMPLR_LIB_PATH
cd "$GRAFT_REPO_ROOT"; PKG=indirect_learning_pose-shape_amd; CMD=$1; shift
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/$PKG/lib_$v.so; [ "$v" = keep ] && L=
  echo "== $v"; SMPLR_LIB_PATH=$L bash -c "$CMD"
done
