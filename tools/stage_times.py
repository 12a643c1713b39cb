#!/usr/bin/env python3
"""Per-stage HIP-event timings of the C-ABI calls (B=128, W=48), incl. the silhouette rasteriser."""
import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ilps_amd
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
sys.argv = ["x"]
import bench
dev = torch.device("cuda:0")
model = synthetic_smpl_model(1234); consts = ops.SMPLConstants.from_model(model, dev); pt = ops.get_part_table(1, dev, consts.V)
B, W = int(os.environ.get("B", 128)), 48
x = torch.tensor(bench.make_x(B, W, 1000), device=dev)
st = torch.cuda.current_stream()
res = bench.stage_breakdown(x, consts, pt, W)
coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts); vp = ops._blend_fwd(coef, consts, x.shape[0]); verts, proj = ops._skin_fwd(vp, A, consts, cam=x)
silh, sarg = ops._silh_fwd(proj, W)
res["silh_fwd"] = round(bench.event_time_ms(lambda: ops._silh_fwd(proj, W), 10, st) * 1e3, 1)
ds = torch.randn_like(silh)
res["silh_bwd"] = round(bench.event_time_ms(lambda: ops._silh_bwd(ds, silh, sarg, proj, W), 10, st) * 1e3, 1)
print(json.dumps(res))
