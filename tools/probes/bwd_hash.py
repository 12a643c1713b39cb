"""Hashes of the decoder's input gradient for fixed seeded inputs and cotangents (deterministic rasteriser backward):
run under two settings of the library to show a change of the backward's launches is bit-exact.  GPU only."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402
from ilps_amd.smpl_model import synthetic_smpl_model  # noqa: E402


def h(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def main():
    dev = torch.device("cuda", 0)
    model = synthetic_smpl_model(1234)
    for W, B, vs in [(48, 1, 1), (48, 5, 1), (48, 33, 1), (48, 128, 1), (32, 7, 2), (64, 40, 1), (48, 520, 1)]:
        dec = SMPLDecoder(model, img_wh=W, vertex_sampling=vs, deterministic=True).to(dev)
        g = torch.Generator(device="cpu").manual_seed(100 + B)
        x = torch.tensor(bench.make_x(B, W, 11 + B), device=dev, requires_grad=True)
        out = dec(x)
        loss = 0.0
        for k in ("seg", "verts", "projects", "J_transformed"):
            w = torch.randn(out[k].shape, generator=g).to(dev)
            loss = loss + (out[k] * w).sum()
        loss.backward()
        torch.cuda.synchronize()
        print("W=%d B=%d vs=%d :" % (W, B, vs), h(x.grad), "sum %.9e" % float(x.grad.double().sum()))


if __name__ == "__main__":
    main()
