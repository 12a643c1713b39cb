"""Per-kernel mean of each PMC counter from rocprofv3 --pmc CSVs: python tools/pmc_summary.py gpurun_out/pmc_TAG_* [kernel-substring]"""
import csv, glob, sys, collections
dirs = [a for a in sys.argv[1:] if '/' in a or a.startswith('gpurun_out')]
filt = [a for a in sys.argv[1:] if a not in dirs]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smplr::', '')
            if filt and not any(x in name for x in filt):
                continue
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean=%14.1f  n=%d" % (c, sum(v) / len(v), len(v)))
