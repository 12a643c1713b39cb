#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05e; mkdir -p $O
bash tools/ab_run.sh "python tools/probes/skinrec_timeline.py" tl 2>&1 | grep -v amdgpu.ids | tee $O/skinrec_tl.txt
bash tools/ab_kernel_b.sh skin_bwd_rec "128 2048" keep 2>&1 | tee $O/skr_ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hostile_inputs.py -x -q -m gpu -k "records_equals or bad_rows or decoder_end_to_end" 2>&1 | tail -2
for b in 128 2048; do timeout -k 10 200 python bench.py --batch $b --steps 50 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=l['ms_per_step_windows']; print('B=$b', l['value'], l['ms_per_step'], w['min'], w['median'], w['max'])"; done
