"""Run-to-run bit-equality of the forward stages (debug probe): python tools/probes/det_fwd.py [B] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
c = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
pt = ops.get_part_table(1, dev, c.V)
x = torch.tensor(bench.make_x(B, 48, 1024), device=dev)
ref = [t.clone() for t in ops._pose_blend_fwd(x, 4, c)]
names = ["Rs", "J", "A", "Jt", "v_posed"]
bad = {n: 0 for n in names}
junk = []
for k in range(reps):
    if k % 7 == 0:      # churn the allocator so that outputs land in memory with other contents
        junk = [torch.randn(1 << 20, device=dev) for _ in range(3)]
    out = ops._pose_blend_fwd(x, 4, c)
    for n, a, b in zip(names, out, ref):
        if not torch.equal(a, b):
            bad[n] += 1
            if bad[n] <= 2:
                d = torch.nonzero((a != b).reshape(B, -1))
                print("run", k, n, "differs at", d.shape[0], "places; first", d[:3].tolist(), float((a - b).abs().max()))
print("fused fwd mismatching runs:", bad)
# the separate path for comparison
coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, c)
vp = ops._blend_fwd(coef, c, B)
print("fused == separate:", all(torch.equal(a, b) for a, b in zip(ref, (Rs, J, A, Jt, vp))))
verts, proj = ops._skin_fwd(ref[4], ref[2], c, cam=x)
rv = verts.clone(); nb = 0
for k in range(reps):
    v2, _ = ops._skin_fwd(ref[4], ref[2], c, cam=x)
    nb += int(not torch.equal(v2, rv))
print("skin_fwd mismatching runs:", nb)
