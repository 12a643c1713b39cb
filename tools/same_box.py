"""Same-lease evidence pair: the eager kernel trace's per-kernel averages against the graph-replayed
headline step measured on the SAME box right after it.
    python tools/same_box.py <eager_kernel_stats.csv> <headline_runs.txt> [bench.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
calls = max(int(r["Calls"]) for r in rows)
ks = [(r["Name"].split("(")[0].replace("void ", "").replace("smplr::", ""), float(r["AverageNs"]) / 1e3, int(r["Calls"]))
      for r in rows]
step = [(n, t) for n, t, c in ks if c >= calls // 2 and not n.startswith("at::") and "pack" not in n and "copy" not in n.lower()]
tot = sum(t for _, t in step)
print("# one lease, one box: rocprofv3 --kernel-trace of `bench.py --mode eager` (one dispatch per kernel per step), then the")
print("# headline (`bench.py --steps 50`: ten steps per HIP-graph replay) three times")
for n, t in sorted(step, key=lambda x: -x[1]):
    print("  %-46s %7.2f us" % (n[:46], t))
print("  %-46s %7.2f us  (%d kernels)" % ("sum of the kernels' averages", tot, len(step)))
runs = [l.split() for l in open(sys.argv[2]) if l.strip()]
for r in runs:
    print("  headline: %.4f ms per step (windows %s / %s / %s), one step per replay %s ms, build %s"
          % (float(r[0]), r[1], r[2], r[3], r[4], r[5]))
best = min(float(r[0]) for r in runs) * 1e3
print("  step - sum = %+.2f us (%.1f %% of the step): launch gaps inside the graph (~1 us per boundary) minus the overlap of a"
      % (best - tot, 100.0 * (best - tot) / best))
print("  kernel's ramp-down with the next one's ramp-up; a negative number means the trace's eager launches run each kernel")
print("  slower than the graph does (cold instruction cache, clock ramp between host launches)")
if len(sys.argv) > 3:
    try:
        b = json.load(open(sys.argv[3]))
        print("  default bench line of the same lease: %.4f ms per step, value_f32 %s (%.4f ms)"
              % (b["ms_per_step"], b.get("value_f32"), b.get("ms_per_step_f32") or 0.0))
    except Exception as e:
        print("  (bench.json unreadable: %s)" % e)
