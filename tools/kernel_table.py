"""Per-kernel table of one step from a rocprofv3 kernel trace and FETCH / WRITE / TCC counter passes of the same command:
us, HBM-side MB (read side x2 on gfx950, MI355X_MICROARCH.md), TB/s, share of the step, L2 hit rate, bytes per mesh.
    python tools/kernel_table.py <kernel_stats.csv> <pmc dir> ... --batch B [--json out.json]"""
import argparse
import collections
import csv
import glob
import json


def short(n):
    return n.split("(")[0].replace("void ", "").replace("smplr::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("pmc", nargs="*")
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--json")
    a = ap.parse_args()
    us = {}
    for r in csv.DictReader(open(a.stats)):
        us[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.pmc:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = lambda v: sum(v) / len(v) if v else None
    rows, tot_us, tot_b = [], 0.0, 0.0
    calls = max(c for _, c in us.values())
    for k, (t, c) in us.items():
        if c < calls // 2 or k.startswith("at::") or "pack" in k or "copy" in k.lower():
            continue                                      # (set-up kernels, not one per step)
        cs = acc.get(k, {})
        f, w = mean(cs.get("FETCH_SIZE", [])), mean(cs.get("WRITE_SIZE", []))
        h, m = mean(cs.get("TCC_HIT_sum", [])), mean(cs.get("TCC_MISS_sum", []))
        b = None if f is None or w is None else 2 * f * 1024 + w * 1024
        rows.append((k, t, b, None if f is None else 2 * f * 1024, None if w is None else w * 1024,
                     None if not h and not m else h / (h + m)))
        tot_us += t
        tot_b += b or 0.0
    rows.sort(key=lambda r: -r[1])
    print("# batch %d: one dispatch per kernel per step (eager trace); bytes = HBM side, per launch" % a.batch)
    print("%-52s %8s %6s %9s %9s %9s %7s %6s %9s" % ("kernel", "us", "share", "read MB", "write MB", "MB", "TB/s", "L2hit", "KB/mesh"))
    out = {}
    for k, t, b, fr, wr, hr in rows:
        print("%-52s %8.1f %5.1f%% %9s %9s %9s %7s %6s %9s"
              % (k[:52], t, 100 * t / tot_us, "-" if fr is None else "%.1f" % (fr / 1e6), "-" if wr is None else "%.1f" % (wr / 1e6),
                 "-" if b is None else "%.1f" % (b / 1e6), "-" if b is None else "%.2f" % (b / t / 1e6),
                 "-" if hr is None else "%.2f" % hr, "-" if b is None else "%.1f" % (b / a.batch / 1e3)))
        out[k] = {"us": round(t, 2), "hbm_bytes_per_launch": None if b is None else int(b), "l2_hit_rate": hr}
    print("%-52s %8.1f %6s %9s %9s %9.1f %7.2f %6s %9.1f" % ("sum", tot_us, "", "", "", tot_b / 1e6, tot_b / tot_us / 1e6, "",
                                                          tot_b / a.batch / 1e3))
    if a.json:
        json.dump({"batch": a.batch, "kernels": out, "step_us_sum": round(tot_us, 1), "step_hbm_bytes": int(tot_b)},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
