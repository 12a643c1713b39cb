cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do
for m in 1 0; do
  echo "FUSE_SKIN=$m: $(SMPLR_FUSE_SKIN=$m python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
for m in 1 0; do
  rm -rf gpurun_out/fsk_$m
  (cd /tmp && SMPLR_FUSE_SKIN=$m TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fsk_$m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline --no-breakdown > /dev/null 2>&1)
  python3 -c "
import csv,glob
f=glob.glob('gpurun_out/fsk_$m/*/*_kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=20:
        tot+=float(r['AverageNs'])/1e3
        if 'seg_bin' in r['Name'] or 'skin_fwd' in r['Name']: print('$m', r['Name'][:44], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
print('$m sum %.1f' % tot)
"
done
