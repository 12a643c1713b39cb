#!/bin/bash
# Copies the summaries of one tools/r02_profile.sh run (gpurun_out/TAG) into profiles/ under a round prefix and
# rebuilds the two JSON files bench.py reads:  bash tools/collect_profiles.sh TAG PREFIX   (e.g. r02b r02)
TAG=$1; P=${2:-r02}; S=gpurun_out/$TAG; D=profiles
cp $S/bench.json $D/${P}_bench.json
cp $S/eager_kernel_stats.csv $D/${P}_eager_kernel_stats.csv
cp $S/default_kernel_stats.csv $D/${P}_default_bench_kernel_stats.csv
cp $S/silh_kernel_stats.csv $D/${P}_silh_kernel_stats.csv
cp $S/pmc_step_sq.txt $D/${P}_pmc_step_sq_counters.txt
cp $S/pmc_silh.txt $D/${P}_pmc_silh_counters.txt
cp $S/valu_issue_probe.txt $D/${P}_valu_issue_probe.txt
cp $S/batch_sweep.txt $D/${P}_batch_sweep.txt
python tools/pmc_traffic.py $S/pmc_step_3 $S/pmc_step_4 $S/pmc_step_5
python tools/raster_sq.py $S/eager_kernel_stats.csv 1.95,3.25 $S/pmc_step_1 $S/pmc_step_2
python - <<PY
import json
# silhouette kernels' HBM-side bytes next to the step's
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$S/pmc_silh_3", "$S/pmc_silh_4", "$S/pmc_silh_5"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smplr::", "").split("<")[0]
            if "silh" in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
mean = lambda v: sum(v) / len(v)
for k, cs in acc.items():
    e = {c: round(mean(v), 1) for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
    out[k] = e
json.dump({"note": "silhouette rasteriser (tools/silh_time.py, B = 128, W = 48): FETCH_SIZE / WRITE_SIZE in KiB per dispatch, "
                   "read side doubled as in pmc_traffic.json", "kernels": out}, open("$D/${P}_silh_traffic.json", "w"), indent=1)
print(json.dumps(out)[:400])
PY
ls $D
