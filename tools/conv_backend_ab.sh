#!/bin/bash
# bash tools/conv_backend_ab.sh: the train step (tools/train_bench.py, 128 images + silhouette) under the two conv back-end
# settings of training.configure_conv_backend: wall time of the whole run (incl. MIOpen's first-call work), ms per step,
# MIOpen workspace warnings on stderr
cd $GRAFT_REPO_ROOT
for m in default find; do
  export SMPLR_CONV_BACKEND=$m
  t0=$(date +%s.%N)
  timeout -k 10 500 python tools/train_bench.py --batch 128 --silhouette --steps 5 --warmup 3 > gpurun_out/conv_$m.json 2> gpurun_out/conv_$m.err
  rc=$?
  t1=$(date +%s.%N)
  echo "[$m] rc=$rc wall $(python3 -c "print(round($t1-$t0,1))") s  warnings $(grep -c IsEnoughWorkspace gpurun_out/conv_$m.err)  $(cat gpurun_out/conv_$m.json | cut -c1-200)"
done
