"""Hashes of the silhouette rasteriser's outputs (scores + arg-max vertices) for fixed seeded inputs - decoder
meshes at several W / camera scales, the reference's rand * 80 recipe, sparse meshes, meshes entirely outside the
cell window - run under two builds of the library to show a kernel change is bit-exact; plus HIP-event timings.
GPU only: python tools/probes/silh_hash.py"""
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
    consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
    cases = []
    for W, B, scale in [(48, 16, 1.0), (48, 5, 2.5), (48, 4, 0.3), (40, 3, 1.0), (33, 3, 1.0), (24, 2, 1.0), (16, 2, 4.0),
                        (48, 3, 8.0), (64, 3, 1.0)]:
        xn = bench.make_x(B, W, 11 + W + B)
        xn[:, 0:2] *= scale
        x = torch.tensor(xn, device=dev)
        coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
        proj = ops._skin_fwd(ops._blend_fwd(coef, consts, B), A, consts, cam=x)[1]
        cases.append(("decoder W=%d B=%d scale=%g" % (W, B, scale), proj, W))
    g = torch.Generator().manual_seed(3)
    cases.append(("rand*80 W=48", (torch.rand(2, 6890, 3, generator=g) * 80.0).to(dev), 48))
    cases.append(("rand*80-16 W=48", (torch.rand(2, 6890, 3, generator=g) * 80.0 - 16.0).to(dev), 48))
    cases.append(("all outside W=48", (torch.rand(2, 6890, 3, generator=g) * 10.0 + 300.0).to(dev), 48))
    sp = torch.full((2, 6890, 3), 1e4)
    sp[:, :5] = torch.rand(2, 5, 3, generator=g) * 48
    cases.append(("5 vertices W=48", sp.to(dev), 48))
    lat = torch.zeros(1, 6890, 3)
    lat[0, :, 0] = (torch.arange(6890) % 83) * 0.5 + 3.25          # many vertices exactly on half-cell borders / equal keys
    lat[0, :, 1] = (torch.arange(6890) // 83) * 0.5 + 2.5
    cases.append(("lattice ties W=48", lat.to(dev), 48))
    for name, proj, W in cases:
        proj = proj.contiguous()
        silh, arg = ops._silh_fwd(proj, W)
        torch.cuda.synchronize()
        print("%-28s silh %s arg %s" % (name, h(silh), h(arg)))
    # timing at the bench's size
    x = torch.tensor(bench.make_x(128, 48, 1000), device=dev)
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    proj = ops._skin_fwd(ops._blend_fwd(coef, consts, 128), A, consts, cam=x)[1]
    st = torch.cuda.current_stream()
    out = ops._silh_fwd(proj, 48)
    t = bench.graph_time_ms(lambda: ops._silh_fwd(proj, 48, out=out), 20, st)
    sys.stderr.write("silh_fwd B=128 W=48: %.2f us\n" % (t * 1e3))


if __name__ == "__main__":
    main()
