"""Distribution of visible vertices per mesh for the bench input (sizes the raster's LDS record copy)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
dev = torch.device("cuda", 0)
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
pt = ops.get_part_table(1, dev, consts.V)
x = torch.tensor(bench.make_x(128, 48, 1000), device=dev)
coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
verts, proj = ops._skin_fwd(ops._blend_fwd(coef, consts, 128), A, consts, cam=x)
mask, seg, arg, rec = ops._vis_seg_fwd(proj, 48, pt)
print("mask values:", torch.unique(mask)[:8].tolist())
m = (mask == 1).sum(1).cpu().numpy()
print("visible/mesh: mean %.1f min %d max %d" % (m.mean(), m.min(), m.max()))
print("percentiles 50/90/99:", np.percentile(m, [50, 90, 99]))
print("meshes with > 640/672/736/760/800 visible:", [(int((m > t).sum())) for t in (640, 672, 736, 760, 800)])
