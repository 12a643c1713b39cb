"""seg_bwd_kernel by phase, from the in-kernel clock stamps of a timeline build (bash tools/build_tl.sh; run as
`bash tools/ab_run.sh "python tools/probes/segbwd_timeline.py" tl` on the GPU box): per wave the shader clock at entry,
after the header load, after the zeroing barrier, around each batch of 8 pixels of the row walk, at the closing barrier
and at exit.  B = 128, W = 48.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import placement, read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd import ops  # noqa: E402
from ilps_amd.smpl_model import synthetic_smpl_model  # noqa: E402


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
    pt = ops.get_part_table(1, dev, consts.V)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev)
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    v_posed = ops._blend_fwd(coef, consts, B)
    verts, proj = ops._skin_fwd(v_posed, A, consts, cam=x, vertex_sampling=1)
    mask, seg, arg, rec = ops._vis_seg_fwd(proj, W, pt)
    dseg = torch.randn_like(seg)
    for it in range(3):
        ws, nsplit = ops._seg_bwd(dseg, arg, rec, consts.V, W, pt, merge=False)
    torch.cuda.synchronize()
    nw = 12
    t = read_stamps("segbwd", B * nsplit, nw)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    names = {1: "header", 2: "zeroed", 24: "row done", 25: "barrier", 26: "exit"}
    for b in range(W // 8):
        names[3 + 2 * b] = "batch %d gathered" % b
        names[4 + 2 * b] = "batch %d summed" % b
    print("median / p90 wave, clocks since the wave's entry:")
    for i in sorted(names):
        v = d[..., i].reshape(-1)
        print("  %-18s %8.0f %8.0f" % (names[i], np.median(v), np.percentile(v, 90)))
    rd = d[..., 24]
    print("row done, by wave of the workgroup (median): " + " ".join("%d" % np.median(rd[:, w]) for w in range(nw)))
    print("row done: workgroup max, median over workgroups %d; slowest workgroup %d" % (np.median(rd.max(axis=1)), rd.max()))
    placement(t)


if __name__ == "__main__":
    main()
