import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
dev = torch.device("cuda", 0)
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
st = torch.cuda.current_stream()
for B in (128, 512, 2048):
    x = torch.tensor(bench.make_x(B, 48, 5), device=dev)
    out = ops._pose_blend_fwd(x, 4, consts)
    vp = out[4]
    t1 = bench.graph_time_ms(lambda: ops._pose_blend_fwd(x, 4, consts, out=out[:4], v_posed=vp), 20, st)
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    t2 = bench.graph_time_ms(lambda: ops._pose_fwd(x, 4, consts, out=(coef, Rs, J, A, Jt)), 20, st)
    t3 = bench.graph_time_ms(lambda: ops._blend_fwd(coef, consts, B, out=vp), 20, st)
    print("B=%d pose_blend3_fwd %.1f us | pose_fwd %.1f + blend3_fwd %.1f us" % (B, t1 * 1e3, t2 * 1e3, t3 * 1e3), flush=True)
