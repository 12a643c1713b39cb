#!/bin/bash
# bash tools/pmc_quick.sh TAG KERNEL_SUBSTRING "ENV": two SQ counter passes of the eager bench step under ENV, then the
# per-dispatch averages of the named kernel -> stdout
TAG=$1; K=$2; E=$3
R=$GRAFT_REPO_ROOT
cd $R
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcq_${TAG}_$i
  (cd /tmp && export TMPDIR=/tmp $E && timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmcq_${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 2 --min-warmup 20 --mode eager --no-cpu-baseline --no-breakdown --no-train-leg > $R/gpurun_out/pmcq_${TAG}_$i.log 2>&1; echo "pmc pass $i rc=$?") || exit 1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for d in ("gpurun_out/pmcq_${TAG}_1", "gpurun_out/pmcq_${TAG}_2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "$K" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("[$E] $K")
for k, v in sorted(acc.items()):
    print("  %-24s %14.0f" % (k, sum(v) / len(v)))
PY
