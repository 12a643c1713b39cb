"""Dump silhouette outputs for one seeded decoder batch (debug A/B across library builds): python silh_dump.py OUT.npz"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
dev = torch.device("cuda", 0)
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
x = torch.tensor(bench.make_x(4, 48, 75), device=dev)
coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
proj = ops._skin_fwd(ops._blend_fwd(coef, consts, 4), A, consts, cam=x)[1]
silh, arg = ops._silh_fwd(proj, 48)
torch.cuda.synchronize()
np.savez(sys.argv[1], silh=silh.cpu().numpy(), arg=arg.cpu().numpy(), proj=proj.cpu().numpy())
