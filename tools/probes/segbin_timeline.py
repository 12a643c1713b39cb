"""seg_bin_kernel by phase, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh; run as
`bash tools/ab_run.sh "python tools/probes/segbin_timeline.py" tl` on the GPU box).  B = 128, W = 48.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

NAMES = ["entry", "requests + zeroing done", "barrier", "staged + z-buffer atomics", "barrier", "cell scan", "barrier",
         "mask written", "classified + counted", "barrier", "block scan", "pixel offsets", "barrier", "part starts (wave 0)",
         "barrier", "placed", "header + sentinels", "barrier", "vslot copied"]


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    dec = SMPLDecoder(None, img_wh=W)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev)
    for it in range(3):
        dec(x)
    torch.cuda.synchronize()
    t = read_stamps("bin", 128, 16)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    print("median / max over waves, clocks since the wave's entry, and the step:")
    prev = 0
    for i in range(1, 19):
        v = d[..., i].reshape(-1)
        m = np.median(v)
        print("  %2d %-28s %8.0f %8.0f   +%d" % (i, NAMES[i], m, v.max(), m - prev))
        prev = m
    placement(t)


if __name__ == "__main__":
    main()
