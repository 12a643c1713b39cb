#!/bin/bash
# Runs on the GPU box: BASELINE configs[3] / [4] (one-GPU leg) train step + rocprofv3 kernel stats. Output: gpurun_out/train_TAG/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_$TAG
mkdir -p $O
timeout -k 10 300 python tools/train_bench.py --batch 256 --steps 10 --warmup 3 > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"; tail -1 $O/c3.json
timeout -k 10 300 python tools/train_bench.py --batch 128 --steps 10 --warmup 3 --silhouette > $O/c4_leg.json 2> $O/c4.err; echo "c4 rc=$?"; tail -1 $O/c4_leg.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/train_bench.py --batch 256 --steps 4 --warmup 2 > $O/rocprof.log 2>&1; echo "rocprof rc=$?")
