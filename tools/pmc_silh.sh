#!/bin/bash
# SQ counters of the silhouette kernels for the current library build (GPU box): bash tools/pmc_silh.sh TAG
TAG=${1:-x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_silh_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- python3 $R/tools/silh_time.py > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
done
cd $R; python tools/pmc_summary.py $O/p1 $O/p2 silh | tee $O/summary.txt
