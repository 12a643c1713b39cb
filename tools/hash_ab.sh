#!/bin/bash
# bash tools/hash_ab.sh "ENV_A" "ENV_B" ...: tools/probes/seg_hash.py (or $HASH_PROBE) under each environment; every output must equal the first
cd $GRAFT_REPO_ROOT
i=0
for e in "$@"; do
  i=$((i+1))
  (export $e; python ${HASH_PROBE:-tools/probes/seg_hash.py} > gpurun_out/hash_$i.txt 2>&1)
  if [ $i -gt 1 ]; then diff -q gpurun_out/hash_1.txt gpurun_out/hash_$i.txt > /dev/null && echo "[$e] IDENTICAL to [$1]" || { echo "[$e] DIFFERS from [$1]"; diff gpurun_out/hash_1.txt gpurun_out/hash_$i.txt | head -6; }; fi
done
