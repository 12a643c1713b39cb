"""Builds profiles/pmc_traffic.json from rocprofv3 --pmc passes (tools/pmc.sh TAG "FETCH_SIZE" "WRITE_SIZE"
"TCC_HIT_sum TCC_MISS_sum").  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; per MI355X_MICROARCH.md (HBM section) the
read side is doubled on gfx950 (128-B requests tallied as 64 B) - exact for wide streaming reads, uncalibrated for
narrow / gather reads.  Usage: python tools/pmc_traffic.py gpurun_out/pmc_TAG_1 gpurun_out/pmc_TAG_2 gpurun_out/pmc_TAG_3
"""
import collections
import csv
import glob
import json
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in [a for a in sys.argv[1:]]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            # (template arguments kept: focal_kernel<.., false> / <.., true> are the loss head's forward and backward)
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smplr::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, each in its own pass (bench.py "
               "--mode eager, B=128, W=48); FETCH_SIZE/WRITE_SIZE are KiB per dispatch; per MI355X_MICROARCH.md the "
               "read side is doubled (gfx950 tallies 128-B requests as 64 B) - exact for wide streaming reads, "
               "uncalibrated for scalar/gather reads",
       "kernels": {}}
mean = lambda v: sum(v) / len(v) if v else None
for k, cs in acc.items():
    f, w = mean(cs.get("FETCH_SIZE", [])), mean(cs.get("WRITE_SIZE", []))
    h, m = mean(cs.get("TCC_HIT_sum", [])), mean(cs.get("TCC_MISS_sum", []))
    e = {}
    if f is not None:
        e["FETCH_SIZE_KiB"] = round(f, 1)
    if w is not None:
        e["WRITE_SIZE_KiB"] = round(w, 1)
    if f is not None and w is not None:
        e["hbm_bytes_per_launch"] = int(2 * f * 1024 + w * 1024)
    if h is not None and m is not None and h + m > 0:
        e["l2_hit_rate"] = round(h / (h + m), 3)
    out["kernels"][k] = e
seg = sum(v.get("hbm_bytes_per_launch", 0) for k, v in out["kernels"].items()
          if k.startswith("seg_bin_kernel") or k.startswith(("raster_fwd_kernel", "raster2_fwd_kernel")))
out["seg_fwd_hbm_bytes_per_launch"] = seg
out["seg_fwd_note"] = "seg_bin_kernel + raster2_fwd_kernel (the two kernels of smplr_seg_fwd / smplr_vis_seg_fwd)"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import ilps_amd  # noqa: E402,F401
from ilps_amd import _lib  # noqa: E402
out["build_id"] = os.environ.get("PROFILE_BUILD_ID") or _lib.source_build_id()   # the library the passes ran on
step = [k for k in out["kernels"] if "pack" not in k and "copy" not in k and not k.startswith("at::")]
out["step_hbm_bytes"] = sum(out["kernels"][k].get("hbm_bytes_per_launch", 0) for k in step)
name = os.environ.get("PMC_TRAFFIC_NAME", "pmc_traffic.json")       # variants: PMC_TRAFFIC_NAME=r03_traffic_fused_loss.json
path = os.path.join(root, "profiles", name)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "seg_fwd bytes/launch", seg)
