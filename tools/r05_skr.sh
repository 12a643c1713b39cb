#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05d; mkdir -p $O
bash tools/ab_kernel_b.sh skin_bwd_rec "128 2048" keep skrA skrB skrAB skrC skrABC skrD 2>&1 | tee $O/skr_ab.txt
for v in keep skrA skrB skrAB skrC skrABC skrD; do
  L=$GRAFT_REPO_ROOT/indirect_learning_pose-shape_amd/lib_$v.so; [ "$v" = keep ] && L=
  SMPLR_LIB_PATH=$L python tools/probes/bwd_hash.py > $O/bwd_hash_$v.txt 2>/dev/null
  diff -q $O/bwd_hash_keep.txt $O/bwd_hash_$v.txt > /dev/null && echo "$v bwd_hash IDENTICAL" || { echo "$v bwd_hash DIFFERS"; diff $O/bwd_hash_keep.txt $O/bwd_hash_$v.txt | head -4; }
done
