# per-kernel time per mesh in the throughput regime (B = 2048), eager launches under rocprofv3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/big
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/big -- python3 bench.py --batch 2048 --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-breakdown > gpurun_out/big.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/big/*/*_kernel_stats.csv')[0]
tot = 0
for r in csv.DictReader(open(f)):
    if int(r['Calls']) >= 10 and 'smplr' in r['Name']:
        us = float(r['AverageNs']) / 1e3
        tot += us
        print("%-40s %8.1f us  %.3f us/mesh" % (r['Name'].split('(')[0][-40:], us, us / 2048))
print("sum %.1f us = %.3f us/mesh" % (tot, tot / 2048))
PY
