"""Hashes of the segmentation raster's outputs for fixed seeded inputs (several W, B, mesh scales, vertex
samplings): run under two builds of the library to show a kernel change is bit-exact.  GPU only."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ilps_amd import ops  # noqa: E402
from ilps_amd.smpl_model import synthetic_smpl_model  # noqa: E402


def h(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def main():
    dev = torch.device("cuda", 0)
    model = synthetic_smpl_model(1234)
    consts = ops.SMPLConstants.from_model(model, dev)
    for W, B, vs, scale in [(48, 16, 1, 1.0), (48, 5, 1, 2.5), (64, 8, 1, 1.0), (32, 4, 1, 1.0), (96, 3, 1, 1.0),
                            (48, 6, 2, 1.0), (40, 3, 1, 0.5), (128, 2, 1, 1.0), (48, 4, 1, 6.0),
                            # (batches that take the large block shape by default)
                            (48, 128, 1, 1.0), (48, 70, 1, 1.3), (64, 37, 1, 1.0)]:
        pt = ops.get_part_table(vs, dev, consts.V)
        xn = bench.make_x(B, W, 7 + W + B)
        xn[:, 0] *= scale                      # camera scale: bigger mesh -> more visible records
        x = torch.tensor(xn, device=dev)
        coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
        v_posed = ops._blend_fwd(coef, consts, B)
        verts, proj = ops._skin_fwd(v_posed, A, consts, cam=x, vertex_sampling=vs)
        mask, seg, arg, rec = ops._vis_seg_fwd(proj, W, pt)
        # seg from an explicit mask with everything visible (long record lists) and with nothing visible
        # slot ids of the visibility path depend on the (atomic) order of the local lists: hash the vertices
        outs = [seg, ops.argmin_vertices(arg, rec), arg[..., 0]]
        for m in (torch.ones_like(mask), torch.zeros_like(mask)):
            s2, a2, r2 = ops._seg_fwd(proj, m, W, pt)[:3]
            outs += [s2, a2]
        torch.cuda.synchronize()
        print("W=%d B=%d vs=%d scale=%.1f vis=%.0f :" % (W, B, vs, scale, float(mask.float().sum() / B)),
              " ".join(h(o) for o in outs), "bg-sum %.6f" % float(seg[..., 0].double().sum()))


if __name__ == "__main__":
    main()
