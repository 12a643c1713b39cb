# raster_fwd time against the number of workgroups (9 tiles per mesh at W = 48; 512 resident workgroups)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for b in 56 57 64 112 113 114 128 170 171; do
  rm -rf gpurun_out/tail_$b
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail_$b -- python3 bench.py --batch $b --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > gpurun_out/tail_$b.log 2>&1 || exit 1
  python3 - $b <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/tail_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if 'raster_fwd' in r['Name'] or 'seg_bin' in r['Name']:
        print("B=%s blocks=%d %-28s avg=%7.2f us" % (sys.argv[1], int(sys.argv[1]) * 9, r['Name'][7:30], float(r['AverageNs']) / 1e3))
PY
done
