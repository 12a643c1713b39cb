#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hostile_inputs.py tests/test_gpu_baseline_sizes.py -x -q -m gpu -k "records_equals or hostile or bad_rows or lds_attribute or gathered_by_vertex or decoder_end_to_end or repeatable or row_independent or deterministic" > $O/pytest1.txt 2>&1; echo "pytest1 rc=$?"; tail -5 $O/pytest1.txt
for b in 128 2048; do
  for e in 0 1; do
    SMPLR_SKIN_BWD_REC=$e timeout -k 10 200 python bench.py --batch $b --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print('B=$b REC=$e', l['value'], l['ms_per_step'], w['min'], w['median'], w['max'])"
  done
done
SMPLR_SKIN_BWD_REC=0 bash tools/ab_kernel_b.sh skin_bwd,seg_bwd,pose_bwd "128 2048" keep 
mv gpurun_out/abk_keep_128 $O/abk_rec0_128; mv gpurun_out/abk_keep_2048 $O/abk_rec0_2048
bash tools/ab_kernel_b.sh skin_bwd,seg_bwd,pose_bwd "128 2048" keep
