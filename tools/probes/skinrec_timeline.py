"""skin_bwd_rec_kernel by phase, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh; run as
`bash tools/ab_run.sh "python tools/probes/skinrec_timeline.py" tl` on the GPU box).  B = 128, W = 48.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

NAMES = ["entry", "vslot + A landed", "ballots done", "barrier 1", "list written, zero rows stored", "barrier 2",
         "record operands landed, T, dv_posed stored", "barrier 3", "weights gathered, mfma done", "after the rounds",
         "barrier 5 (partial sums)", "exit"]


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    dec = SMPLDecoder(None, img_wh=W)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev, requires_grad=True)
    for it in range(3):
        out = dec(x)
        (out["seg"] ** 2).sum().backward()
    torch.cuda.synchronize()
    t = read_stamps("skinrec", 896, 4)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    print("median / p90 / max over waves, clocks since the wave's entry:")
    for i in range(1, 12):
        v = d[..., i].reshape(-1)
        print("  %-46s %8.0f %8.0f %8.0f" % (NAMES[i], np.median(v), np.percentile(v, 90), v.max()))
    R = t[..., 12].reshape(-1)
    print("records per workgroup: median %d, p90 %d, max %d" % (np.median(R), np.percentile(R, 90), R.max()))
    placement(t)
    wall = t[:, 0, 28]
    print("kernel span by the 100 MHz clock: first entry to last entry %.2f us" % ((wall.max() - wall.min()) / 100.0))


if __name__ == "__main__":
    main()
