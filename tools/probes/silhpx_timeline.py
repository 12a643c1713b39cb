"""silh_px_kernel by phase and by tile, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh;
`bash tools/ab_run.sh "python tools/probes/silhpx_timeline.py" tl` on the GPU box).  B = 128, W = 48, two workgroups per
mesh; TL_HINT=1: with the 31-part head's per-pixel hint (both heads on).  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

NAMES = ["entry", "requests + zeroing (barrier)", "cell counts (barrier)", "scan + offset table (barrier)",
         "placement (barrier)", "pixels done"]


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev)
    heads = ("seg", "silhouette") if os.environ.get("TL_HINT") else ("silhouette",)
    dec = SMPLDecoder(None, img_wh=W, heads=heads)
    for it in range(3):
        dec(x)
    torch.cuda.synchronize()
    t = read_stamps("silhpx", 256, 16)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    print("heads %s; median / p90 / max over waves, clocks since the wave's entry:" % (heads,))
    for i in range(1, 6):
        v = d[..., i].reshape(-1)
        print("  %-34s %8.0f %8.0f %8.0f" % (NAMES[i], np.median(v), np.percentile(v, 90), v.max()))
    px = d[..., 5] - d[..., 4]                         # (wg, wave): the wave's pixel phase
    nt = t[..., 6]
    print("pixel phase per wave: median %d p90 %d max %d; per workgroup max / mean of its 16 waves: median %.2f p90 %.2f"
          % (np.median(px), np.percentile(px, 90), px.max(), np.median(px.max(1) / px.mean(1)),
             np.percentile(px.max(1) / px.mean(1), 90)))
    print("tiles per wave: %s" % dict(zip(*np.unique(nt, return_counts=True))))
    # per tile durations: end clocks at 9 + 2k, the first tile starts at stamp 4
    dur = []
    for k in range(6):
        end = (t[..., 9 + 2 * k] - t[..., 0]) & 0xFFFFFFFF
        beg = d[..., 4] if k == 0 else ((t[..., 7 + 2 * k] - t[..., 0]) & 0xFFFFFFFF)
        sel = nt > k
        dur.append((end - beg)[sel])
    dur = np.concatenate(dur)
    print("tile time: n %d median %d p90 %d max %d; sum over a workgroup / 16 waves: median %d"
          % (dur.size, np.median(dur), np.percentile(dur, 90), dur.max(),
             np.median([sum(((t[w, :, 9 + 2 * k] - t[w, :, 0]) & 0xFFFFFFFF)[nt[w] > k].sum()
                            - (d[w, :, 4] if k == 0 else ((t[w, :, 7 + 2 * k] - t[w, :, 0]) & 0xFFFFFFFF))[nt[w] > k].sum()
                            for k in range(6)) / 16 for w in range(256)])))
    a, b, c = t[..., 20], t[..., 21], t[..., 22]
    print("per wave, clocks: own cell / nearest occupied cell (steps 1-2) median %d p90 %d | candidate map (3) median %d p90 %d | "
          "ranges + write-out median %d p90 %d" % (np.median(a), np.percentile(a, 90), np.median(b), np.percentile(b, 90),
                                                   np.median(c), np.percentile(c, 90)))
    placement(t)


if __name__ == "__main__":
    main()
