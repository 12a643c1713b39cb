"""skin_bwd_kernel by phase, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh; run as
`bash tools/ab_run.sh "python tools/probes/skinbwd_timeline.py" tl` on the GPU box).  B = 128, W = 48.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

NAMES = ["entry", "operands landed (2 hops)", "barrier 1", "dv_posed stored, weights requested", "barrier 2",
         "mfma done", "barrier 3", "exit"]


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    dec = SMPLDecoder(None, img_wh=W)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev, requires_grad=True)
    for it in range(3):
        out = dec(x)
        ((out["seg"] ** 2).sum() + (out["verts"] ** 2).sum()).backward()
    torch.cuda.synchronize()
    t = read_stamps("skin", 3456, 4)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    print("median / p90 over waves, clocks since the wave's entry:")
    for i in range(1, 8):
        v = d[..., i].reshape(-1)
        print("  %-36s %8.0f %8.0f" % (NAMES[i], np.median(v), np.percentile(v, 90)))
    placement(t)


if __name__ == "__main__":
    main()
