"""Builds profiles/raster_sq.json: the SQ counters of raster_fwd_kernel (rocprofv3 --pmc passes of `bench.py --mode
eager`, tools/pmc.sh) + its kernel-trace average duration, from which bench.py derives `roofline.valu_issue_frac` and
`valu_active_frac`.  Usage:
    python tools/raster_sq.py <kernel_stats.csv of the eager trace> <issue cycles per plain VALU instr>,<per packed /
           three-operand instr> (both from tools/probes/valu_issue_probe at 4 waves per SIMD) gpurun_out/pmc_TAG_1 [...]
"""
import collections
import csv
import glob
import json
import os
import sys

KERNEL = "raster2_fwd_kernel"     # (round 4: the two-pixel kernel; rounds 1-3: raster_fwd_kernel)
stats = sys.argv[1]
issue_plain, issue_packed = (float(v) for v in sys.argv[2].split(","))
us = None
for r in csv.DictReader(open(stats)):
    if KERNEL in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
acc = collections.defaultdict(list)
for d in sys.argv[3:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"kernel": KERNEL, "kernel_us": us, "kernel_us_source": os.path.basename(stats), "simds": 1024, "clock_hz": 2.4e9,
       "clock_note": "nominal peak clock; the clock held under load is lower, so the fractions are lower bounds",
       "issue_cycles_plain": issue_plain, "issue_cycles_packed": issue_packed,
       "issue_cycles_source": "tools/probes/valu_issue_probe (profiles/r04_valu_issue_probe.txt) at 8 waves per SIMD, the kernel's "
                              "occupancy: v_fma_f32 (plain) and v_pk_fma_f32 / v_pk_add_f32 / v_min3_f32 (the pair loop's instructions)",
       "units": "SQ_INSTS_* = wave-instructions; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* = quad-cycles (x4 = cycles), summed over the chip"}
for k, v in sorted(acc.items()):
    out[k] = sum(v) / len(v)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import ilps_amd  # noqa: E402,F401
from ilps_amd import _lib  # noqa: E402
out["build_id"] = os.environ.get("PROFILE_BUILD_ID") or _lib.source_build_id()   # the library the passes ran on
path = os.path.join(root, "profiles", "raster_sq.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, {k: out[k] for k in ("kernel_us", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU") if k in out})
