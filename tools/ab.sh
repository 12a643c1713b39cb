#!/bin/bash
# A/B of builds of the library on ONE box: lib_<name>.so for the names in $LIBS (default: old new; in the package directory) are copied over
# libsmplraster_hip.so in turn and the bench is profiled; kernel averages are printed per run.
#   tools/ab.sh "<pytest -k expression run on the new build first>" [kernel-name-filter]
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
PKG=indirect_learning_pose-shape_amd
KEXPR="$1"; FILT="${2:-.}"
mkdir -p gpurun_out/ab
cp $PKG/lib_new.so $PKG/libsmplraster_hip.so
if [ -n "$KEXPR" ]; then
  timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/ab/pytest.log 2>&1 || { tail -30 gpurun_out/ab/pytest.log; exit 1; }
  tail -2 gpurun_out/ab/pytest.log
fi
if [ -n "$HASH" ]; then
  # bit-exactness of the segmentation raster across the builds (each against the first)
  first=""
  for v in ${LIBS:-old new}; do
    cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so
    timeout -k 10 200 python3 tools/probes/seg_hash.py > gpurun_out/ab/hash_$v.txt 2>&1 || { tail -20 gpurun_out/ab/hash_$v.txt; exit 1; }
    if [ -z "$first" ]; then first=$v; continue; fi
    if diff gpurun_out/ab/hash_$first.txt gpurun_out/ab/hash_$v.txt > gpurun_out/ab/hash_diff_$v.txt; then echo "seg hashes $first == $v ($(wc -l < gpurun_out/ab/hash_$v.txt) cases)"; else echo "SEG HASHES DIFFER: $first vs $v"; fi
  done
fi
for rep in 1 2; do
  for v in ${LIBS:-old new}; do
    cp $PKG/lib_$v.so $PKG/libsmplraster_hip.so
    rm -rf gpurun_out/ab/prof_$v$rep
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/prof_$v$rep -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-breakdown --mode eager $BENCH_ARGS > gpurun_out/ab/bench_$v$rep.log 2>&1 || { tail -20 gpurun_out/ab/bench_$v$rep.log; exit 1; }
    echo "== $v $rep"
    python3 - "$v$rep" "$FILT" <<'PY'
import csv, glob, sys, re
f = glob.glob('gpurun_out/ab/prof_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if int(r['Calls']) >= 40:
        tot += float(r['TotalDurationNs']) / int(r['Calls'])
        if re.search(sys.argv[2], r['Name']):
            print("  %-50s calls=%4s avg=%7.2f us" % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3))
print("  sum of per-call averages: %.1f us" % (tot / 1e3))
PY
  done
done
cp $PKG/lib_new.so $PKG/libsmplraster_hip.so
