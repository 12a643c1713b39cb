#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
for s in 1 2 4; do timeout -k 10 200 python bench.py --batch 2048 --streams $s --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown --no-train-leg 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=2048 streams=$s', l['value'], l['ms_per_step'])"; done
