"""Times the silhouette rasteriser (smplr_silh_fwd / smplr_silh_bwd) alone: python tools/silh_time.py [B] [W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ilps_amd  # noqa: F401
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
x = torch.tensor(bench.make_x(B, W, 1000), device=dev)
coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
vp = ops._blend_fwd(coef, consts, B)
verts, proj = ops._skin_fwd(vp, A, consts, cam=x)
st = torch.cuda.current_stream()
silh, arg = ops._silh_fwd(proj, W)
d = torch.randn_like(silh)
tf = bench.event_time_ms(lambda: ops._silh_fwd(proj, W), 20, st)
tb = bench.event_time_ms(lambda: ops._silh_bwd(d, silh, arg, proj, W), 20, st)
print("silhouette B=%d W=%d: fwd %.1f us  bwd %.1f us" % (B, W, tf * 1e3, tb * 1e3))
