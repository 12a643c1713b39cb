"""python tools/graph_gaps.py <rocprofv3 output dir>: from a --kernel-trace of a graph-replayed bench run, the last 2000
dispatches: per kernel the mean duration and the mean gap to the next dispatch's start (queue + drain + launch)."""
import csv
import glob
import sys
from collections import OrderedDict

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -2000:]
acc = OrderedDict()
for i, (s, e, n) in enumerate(rows[:-1]):
    n = n.split("(")[0][-44:]
    a = acc.setdefault(n, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e - s
    a[2] += rows[i + 1][0] - e
tot_d = tot_g = 0.0
for n, (c, d, g) in acc.items():
    print("%-46s n=%5d  dur %7.2f us  gap after %6.2f us" % (n, c, d / c / 1e3, g / c / 1e3))
    tot_d += d / c
    tot_g += g / c
print("sum of durations %.2f us, of gaps %.2f us (one of each kernel)" % (tot_d / 1e3, tot_g / 1e3))
