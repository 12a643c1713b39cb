#!/bin/bash
# Runs on the GPU box: parity tests, rocprof kernel stats (eager), graph bench. Output in gpurun_out/.
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > $R/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?")
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log | cut -c1-200
