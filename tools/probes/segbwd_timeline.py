"""Reads the in-kernel clock stamps a timeline build of seg_bwd_kernel (raster.hip built with the SB_TL edits of
DESIGN.md section 4, `lib_tl.so` copied over the library) leaves in its workspace: per wave, the shader clock at
kernel entry, after the header load, after the zeroing barrier, around each batch of 8 pixels, at the closing barrier
and at exit, plus HW_ID / XCC_ID.  Prints the median wave's phase lengths and how many workgroups shared a CU.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    t = ws.view(torch.int32).cpu().numpy().view(np.uint32)[:B * nsplit * 5 * 8192].reshape(B, nsplit, 5, 8192)[:, :, 4, :]
    nw = 12
    st = t[:, :, :nw * 32].reshape(B, nsplit, nw, 32).astype(np.int64)
    t0 = st[..., 0]
    d = (st - t0[..., None]) & 0xFFFFFFFF
    names = {1: "header", 2: "zeroed", 16: "row done", 17: "barrier", 18: "exit"}
    for b in range(6):
        names[3 + 2 * b] = "batch %d gathered" % b
        names[4 + 2 * b] = "batch %d summed" % b
    print("median / p90 wave, clocks since the wave's entry:")
    for i in sorted(names):
        v = d[..., i].reshape(-1)
        print("  %-18s %8.0f %8.0f" % (names[i], np.median(v), np.percentile(v, 90)))
    # spread of entry times over the launch, and co-residency
    allt0 = t0.reshape(-1)
    rd = d[..., 16]
    print("row done, by wave of the workgroup (median): " + " ".join("%d" % np.median(rd[:, :, w]) for w in range(nw)))
    print("row done: workgroup max, median over workgroups %d; slowest workgroup %d" % (np.median(rd.max(axis=2)), rd.max()))
    hw = st[..., 20][:, :, 0]
    xcc = st[..., 21][:, :, 0] & 0xF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 1
    key = (xcc * 8 + se) * 32 + sh * 16 + cu
    uniq, cnt = np.unique(key.reshape(-1), return_counts=True)
    print("workgroups %d on %d distinct (xcc, se, sh, cu); per CU: max %d" % (key.size, uniq.size, cnt.max()))


if __name__ == "__main__":
    main()
