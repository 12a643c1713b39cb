import csv, glob, sys
tag = sys.argv[1]
f = glob.glob('/root/repo/gpurun_out/prof_%s/*/*_kernel_stats.csv' % tag)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    print("%-58s calls=%4s avg=%8.1f us pct=%5s" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
print('total per step us', tot/25/1e3)
