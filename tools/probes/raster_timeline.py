"""raster_fwd_kernel by phase and by wave, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh;
`bash tools/ab_run.sh "python tools/probes/raster_timeline.py" tl` on the GPU box).  B = 128, W = 48.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

NAMES = ["entry", "requests issued", "barrier (records in LDS)", "row tables built", "parts scanned", "barrier",
         "merged + written"]


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev)
    if os.environ.get("TL_LOSS"):                      # the kernel with the loss head as its epilogue
        from ilps_amd.focal_loss import softmax_focal_loss
        dec = SMPLDecoder(None, img_wh=W, outputs=(), loss=softmax_focal_loss(2.0, True))
        labels = torch.randint(0, 32, (B, W, W), device=dev, dtype=torch.int32)
        for it in range(3):
            dec(x, labels)
    else:
        dec = SMPLDecoder(None, img_wh=W)
        for it in range(3):
            dec(x)
    torch.cuda.synchronize()
    t = read_stamps("raster", 1152, 16)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    print("median / p90 / max over waves, clocks since the wave's entry:")
    for i in range(1, 7):
        v = d[..., i].reshape(-1)
        print("  %-26s %8.0f %8.0f %8.0f" % (NAMES[i], np.median(v), np.percentile(v, 90), v.max()))
    scan = d[..., 4] - d[..., 3]                       # (wg, wave)
    print("part scan per wave: median %d, p10 %d, p90 %d; per workgroup max / mean of its 16 waves: median %.2f, p90 %.2f"
          % (np.median(scan), np.percentile(scan, 10), np.percentile(scan, 90),
             np.median(scan.max(1) / scan.mean(1)), np.percentile(scan.max(1) / scan.mean(1), 90)))
    g = t[..., 8]
    for k in range(8):
        sel = scan[g == k]
        print("  range %d: waves %d, scan median %d p90 %d, parts median %d" % (k, sel.size, np.median(sel), np.percentile(sel, 90),
                                                                              np.median(t[..., 7][g == k])))
    wait = d[..., 5] - d[..., 4]
    print("wait at the barrier after the scan: median %d, p90 %d" % (np.median(wait), np.percentile(wait, 90)))
    placement(t)
    # when workgroups enter and leave (10 ns ticks after the launch's first wave; exit = entry + life at 2.4 GHz)
    wall = t[:, 0, 28]
    rel = (wall - wall.min()) & 0xFFFFFFFF
    life = d[..., 6].max(1)
    end = rel + life / 24.0
    order = np.argsort(rel)
    for name, sel in (("first 512 to enter", order[:512]), ("next 512", order[512:1024]), ("last 128", order[1024:])):
        if len(sel):
            print("  %-18s enter p10/50/90 = %5d %5d %5d ticks, life p10/50/90 = %5d %5d %5d clocks, leave p50/max = %5d %5d ticks"
                  % (name, *[int(np.percentile(rel[sel], q)) for q in (10, 50, 90)],
                     *[int(np.percentile(life[sel], q)) for q in (10, 50, 90)], int(np.median(end[sel])), int(end[sel].max())))


if __name__ == "__main__":
    main()


def part_detail():
    """Stamps 10..14 of the two-pixel kernel's stream scan (first non-empty part of every wave)."""
    t = read_stamps("raster", 1152, 16)
    ok = t[..., 14] > 0
    n = t[..., 14][ok]
    grp = ((t[..., 11] - t[..., 10]) & 0xFFFFFFFF)[ok]
    rsc = ((t[..., 12] - t[..., 11]) & 0xFFFFFFFF)[ok]
    fin = ((t[..., 13] - t[..., 12]) & 0xFFFFFFFF)[ok]
    print("first part of a wave: groups median %d; group loop median %d clocks (%.0f per group), re-scan %d, finalise %d"
          % (np.median(n), np.median(grp), np.median(grp / np.maximum(n, 1)), np.median(rsc), np.median(fin)))
    for k in (1, 2, 3, 4, 6, 8, 12):
        sel = n == k
        if sel.sum() > 50:
            print("  parts of %2d groups: %5d waves, loop %5.0f clocks (%.0f per group), re-scan %4.0f, finalise %4.0f"
                  % (k, sel.sum(), np.median(grp[sel]), np.median(grp[sel]) / k, np.median(rsc[sel]), np.median(fin[sel])))


if os.environ.get("TL_PARTS"):
    part_detail()
