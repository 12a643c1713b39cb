#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the float64 oracle (oracle/np_oracle.py, torch_oracle.py).

The reference cannot be imported (Python 2 + TensorFlow 1 + missing SMPL pkl),
so these vectors pin the ORACLE (regression) and give the GPU tests a fixed target; they are not
outputs of the reference itself.  Inputs: seeded synthetic SMPL model (seed 1234) and
`tests/_inputs.make_x`.  Run from the repo root: `python tools/make_golden.py`.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilps_amd  # noqa: E402,F401
from ilps_amd.smpl_model import synthetic_smpl_model, load_part_tables  # noqa: E402
from oracle import np_oracle as o  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402
from _inputs import make_x  # noqa: E402


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    model = synthetic_smpl_model(1234)
    B = 2
    # every (W, vertex_sampling) in {48, 64} x {None, 2, 5}; an existing file is left alone
    # unless --force (the first three were committed in round 1 and the tests were tuned on them)
    for W, vs in ((48, None), (48, 5), (64, 2), (48, 2), (64, None), (64, 5)):
        name = "decoder_w%d_vs%s.npz" % (W, vs or 1)
        if os.path.exists(os.path.join(out, name)) and "--force" not in sys.argv:
            print(name, "exists")
            continue
        x = make_x(B, W, seed=900 + W + (vs or 0))
        ids, off = load_part_tables(vs)
        x64 = x.astype(np.float64)
        r = o.smpl_layer_call(x64, model, return_all=True)
        proj = o.orthographic_project(r["verts"], x64, vs)
        mask = o.compute_mask(proj)
        seg = o.projects_to_seg(proj, mask, W, ids, off, vs)
        d = dict(x=x, verts=r["verts"].astype(np.float32), J_transformed=r["J_transformed"].astype(np.float32),
                 mask_visible=np.packbits(mask == 1.0, axis=1), seg=seg.astype(np.float32))
        rng = np.random.default_rng(W)
        gs = rng.normal(0, 1, seg.shape).astype(np.float32)
        xo = torch.tensor(x64, requires_grad=True)
        mo = torch.tensor(mask)
        _, po, _, so = to.decoder_forward(to.TorchSMPL(model), xo, lambda p: mo, W, ids, off, vs)
        loss = (so * torch.tensor(gs.astype(np.float64))).sum()
        if vs is None:
            silh = o.projects_to_silhouette(proj, W)
            gl = rng.normal(0, 1, silh.shape).astype(np.float32)
            loss = loss + (to.projects_to_silhouette(po, W) * torch.tensor(gl.astype(np.float64))).sum()
            d.update(silh=silh.astype(np.float32), cot_silh=gl)
        loss.backward()
        d.update(cot_seg_seed=np.array([W]), dx=xo.grad.numpy())
        np.savez_compressed(os.path.join(out, name), **d)
        print(name, {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
