"""Run-to-run bit-equality of every output of the decoder step, with allocator churn (debug probe)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
det = len(sys.argv) > 3 and sys.argv[3] == "det"
dev = torch.device("cuda", 0)
c = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
pt = ops.get_part_table(1, dev, c.V)
x = torch.tensor(bench.make_x(B, 48, 77), device=dev)
g = torch.randn(B, 48, 48, 32, device=dev)
gl = torch.randn(B, 48, 48, 2, device=dev)
names = ["verts", "proj", "mask", "seg", "silh", "Jt", "loss", "dx"]
def step():
    xg = x.detach().requires_grad_(True)
    outs = ops.DecoderFn.apply(xg, c, 4, 48, 1, pt, 64, True, True, 1, det)
    torch.autograd.backward([outs[3], outs[4]], [g, gl])
    return [o.detach().clone() for o in outs] + [xg.grad.clone()]
ref = step()
bad = {n: 0 for n in names}
for k in range(reps):
    if k % 5 == 0:
        junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(4)]
        del junk
    if k % 9 == 0:      # another batch size in between, as the tests do
        xs = x[:32].detach().requires_grad_(True)
        o = ops.DecoderFn.apply(xs, c, 4, 48, 1, pt, 64, True, True, 1, det)
        torch.autograd.backward([o[3], o[4]], [g[:32], gl[:32]])
    cur = step()
    for n, a, b in zip(names, cur, ref):
        if n == "dx" and not det:
            continue
        if not torch.equal(a, b):
            bad[n] += 1
            if bad[n] <= 2:
                d = torch.nonzero((a != b).reshape(B, -1))
                print("run", k, n, "differs at", d.shape[0], "places; first", d[:3].tolist(), "max", float((a.float() - b.float()).abs().max()))
print("B", B, "det", det, "mismatching runs:", bad)
