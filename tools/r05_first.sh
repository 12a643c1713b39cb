#!/bin/bash
# round 5, first GPU session: correctness of the chunked table passes, then what they cost / buy
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "record_list or raster_plan or generic_masks or custom_part or seg_forward or vis_seg or seg_stages" > gpurun_out/r05a/pytest1.txt 2>&1; echo "pytest1 rc=$?"; tail -3 gpurun_out/r05a/pytest1.txt
python -m pytest tests/test_gpu_raster_variants.py tests/test_abi.py -x -q > gpurun_out/r05a/pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/r05a/pytest2.txt
python tools/record_sweep.py > gpurun_out/r05a/sweep_new.txt 2>&1; echo "sweep new rc=$?"
SMPLR_LIB_PATH=$GRAFT_REPO_ROOT/indirect_learning_pose-shape_amd/lib_r04.so python tools/record_sweep.py > gpurun_out/r05a/sweep_r04.txt 2>&1; echo "sweep r04 rc=$?"
cat gpurun_out/r05a/sweep_new.txt gpurun_out/r05a/sweep_r04.txt
for i in 1 2; do
  SMPLR_LIB_PATH=$GRAFT_REPO_ROOT/indirect_learning_pose-shape_amd/lib_r04.so python bench.py --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('r04 ', d['ms_per_step'], d['ms_per_step_windows'])"
  python bench.py --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('new ', d['ms_per_step'], d['ms_per_step_windows'])"
done
BENCH_ARGS="" bash tools/ab_kernel.sh raster2,seg_bin r04 keep
