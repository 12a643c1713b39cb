#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "skinning_inside or decoder_end_to_end or vis_seg_fused or seg_forward or visibility_exact" 2>&1 | tail -2
bash tools/ab_kernel_b.sh seg_bin "128 2048" prev keep prev keep 2>&1 | tee $O/bin_ab.txt
python tools/probes/seg_hash.py > $O/hash_keep.txt 2>/dev/null; SMPLR_LIB_PATH=$GRAFT_REPO_ROOT/indirect_learning_pose-shape_amd/lib_prev.so python tools/probes/seg_hash.py > $O/hash_prev.txt 2>/dev/null; diff -q $O/hash_keep.txt $O/hash_prev.txt && echo "seg_hash IDENTICAL"
