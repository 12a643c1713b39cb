"""Differentiable torch-CPU restatement of the hot path ("reference-shaped").

TEST INFRASTRUCTURE ONLY (see `oracle/np_oracle.py` for the rules and the PARITY UNPINNED
statement).  Two uses:

* float64 + autograd = the gradient oracle for the hand-written HIP backward kernels
  (TensorFlow's autodiff rules for `reduce_max`, `norm`, `clip_by_value` coincide with torch's
  away from ties / d = 0);
* float32 = the CPU baseline `bench.py` times beside the GPU number: the same dense
  formulation, op order and materialised intermediates as the reference's TF graph
  (`keras_smpl/batch_smpl.py:96-153`, `projection.py:54-81`, `projects_to_seg.py:34-69`).

It is validated against the NumPy oracle in `tests/test_oracle_kat.py`.
"""
from __future__ import annotations

import torch


def batch_skew(vec):
    """`keras_smpl/batch_smpl.py:230-253`."""
    n = vec.shape[0]
    z = torch.zeros(n, dtype=vec.dtype)
    return torch.stack([z, -vec[:, 2], vec[:, 1],
                        vec[:, 2], z, -vec[:, 0],
                        -vec[:, 1], vec[:, 0], z], dim=1).reshape(n, 3, 3)


def batch_rodrigues(theta):
    """`keras_smpl/batch_smpl.py:255-276`."""
    angle = torch.sqrt(torch.sum((theta + 1e-8) ** 2, dim=1)).unsqueeze(-1)
    r = (theta / angle).unsqueeze(-1)
    angle = angle.unsqueeze(-1)
    cos, sin = torch.cos(angle), torch.sin(angle)
    outer = r @ r.transpose(1, 2)
    eyes = torch.eye(3, dtype=theta.dtype).unsqueeze(0).expand(theta.shape[0], 3, 3)
    return cos * eyes + (1 - cos) * outer + sin * batch_skew(r[:, :, 0])


def batch_global_rigid_transformation(Rs, Js, parent):
    """`keras_smpl/batch_smpl.py:168-228`."""
    N = Rs.shape[0]
    Js = Js.unsqueeze(-1)

    def make_A(R, t):
        R_homo = torch.cat([R, torch.zeros(N, 1, 3, dtype=R.dtype)], dim=1)
        t_homo = torch.cat([t, torch.ones(N, 1, 1, dtype=R.dtype)], dim=1)
        return torch.cat([R_homo, t_homo], dim=2)

    results = [make_A(Rs[:, 0], Js[:, 0])]
    for i in range(1, len(parent)):
        A_here = make_A(Rs[:, i], Js[:, i] - Js[:, int(parent[i])])
        results.append(results[int(parent[i])] @ A_here)
    results = torch.stack(results, dim=1)
    new_J = results[:, :, :3, 3]
    Js_w0 = torch.cat([Js, torch.zeros(N, 24, 1, 1, dtype=Rs.dtype)], dim=2)
    init_bone = torch.nn.functional.pad(results @ Js_w0, (3, 0))
    return new_J, results - init_bone


class TorchSMPL:
    """Constants of `SMPLLayer.build` (`batch_smpl.py:31-93`) as torch tensors."""

    def __init__(self, model, dtype=torch.float64):
        t = lambda a: torch.as_tensor(a, dtype=dtype)
        self.dtype = dtype
        self.V = model.v_template.shape[0]
        self.v_template = t(model.v_template)
        self.shapedirs = t(model.shapedirs.reshape(-1, 10).T.copy())
        self.J_regressor = t(model.J_regressor.T.copy())
        self.posedirs = t(model.posedirs.reshape(-1, 207).T.copy())
        self.parents = [int(p) for p in model.parents]
        self.lbs_weights = t(model.weights)

    def __call__(self, x, num_cam=4, return_all=False):
        """`SMPLLayer.call`, `batch_smpl.py:96-153`."""
        N, V = x.shape[0], self.V
        thetas = x[:, num_cam:num_cam + 72]
        betas = x[:, num_cam + 72:]
        v_shaped = (betas @ self.shapedirs).reshape(-1, V, 3) + self.v_template
        J = torch.stack([v_shaped[:, :, c] @ self.J_regressor for c in range(3)], dim=2)
        Rs = batch_rodrigues(thetas.reshape(-1, 3)).reshape(-1, 24, 3, 3)
        pose_feature = (Rs[:, 1:] - torch.eye(3, dtype=x.dtype)).reshape(-1, 207)
        v_posed = (pose_feature @ self.posedirs).reshape(-1, V, 3) + v_shaped
        J_transformed, A = batch_global_rigid_transformation(Rs, J, self.parents)
        W = self.lbs_weights.repeat(N, 1).reshape(N, -1, 24)
        T = (W @ A.reshape(N, 24, 16)).reshape(N, -1, 4, 4)
        v_homo = T @ torch.cat([v_posed, torch.ones(N, V, 1, dtype=x.dtype)], dim=2).unsqueeze(-1)
        verts = v_homo[:, :, :3, 0]
        if return_all:
            return verts, J_transformed, A
        return verts


def orthographic_project(verts, smpl, vertex_sampling=None):
    """`keras_smpl/projection.py:54-81`."""
    if vertex_sampling is not None:
        verts = verts[:, ::vertex_sampling, :]
    u = smpl[:, 2:3] + verts[:, :, 0] * smpl[:, 0:1]
    v = smpl[:, 3:4] + verts[:, :, 1] * smpl[:, 1:2]
    return torch.stack([u, v, verts[:, :, 2]], dim=2)


def _grid(img_wh, dtype):
    r, c = torch.meshgrid(torch.arange(img_wh), torch.arange(img_wh), indexing="ij")
    return torch.stack([c, r], dim=2).to(dtype).reshape(-1, 2)       # (x=c, y=r)


def projects_to_seg(projects_with_depth, mask_vals, img_wh, part_ids, part_off,
                    vertex_sampling=None):
    """`keras_smpl/projects_to_seg.py:9-69`, materialising (N, W*W, n_p, 2) like the TF graph."""
    proj = projects_with_depth[:, :, :2]
    grid = _grid(img_wh, proj.dtype)
    N = proj.shape[0]
    segs = []
    for part in range(len(part_off) - 1):
        idx = torch.as_tensor(part_ids[part_off[part]:part_off[part + 1]], dtype=torch.long)
        if vertex_sampling is not None:
            idx = idx // vertex_sampling
        pp = proj[:, idx, :].unsqueeze(1).expand(N, img_wh * img_wh, idx.numel(), 2)
        pm = mask_vals[:, idx].unsqueeze(1)
        diff = pp - grid.unsqueeze(1).unsqueeze(0)
        norm = torch.sqrt(torch.sum(diff * diff, dim=3)) * pm
        segs.append(torch.exp(-norm).max(dim=2).values.reshape(N, img_wh, img_wh))
    stacked = torch.stack(segs, dim=3)
    sil = 1.0 - torch.clamp(stacked.sum(dim=3), 0.0, 1.0)
    out = torch.cat([sil.unsqueeze(3), stacked], dim=3)
    return torch.flip(out, dims=[1])


def projects_to_silhouette(projects_with_depth, img_wh):
    """`keras_smpl/projects_to_silhouette.py:14-44`."""
    proj = projects_with_depth[:, :, :2]
    grid = _grid(img_wh, proj.dtype)
    N = proj.shape[0]
    diff = proj.unsqueeze(1) - grid.unsqueeze(1).unsqueeze(0)          # (N,W2,V,2)
    norm = torch.sqrt(torch.sum(diff * diff, dim=3))
    sil = torch.exp(-norm / 1.2).max(dim=2).values.reshape(N, img_wh, img_wh)
    return torch.flip(torch.stack([1.0 - sil, sil], dim=3), dims=[1])


def decoder_forward(smpl: TorchSMPL, x, mask_fn, img_wh, part_ids, part_off,
                    vertex_sampling=None):
    """model.py:108-118 wiring: SMPLLayer -> project -> compute_mask -> projects_to_seg.

    `mask_fn(proj_detached) -> mask` supplies the (non-differentiable) visibility mask.
    """
    verts = smpl(x)
    proj = orthographic_project(verts, x, vertex_sampling)
    mask = mask_fn(proj.detach())
    seg = projects_to_seg(proj, mask, img_wh, part_ids, part_off, vertex_sampling)
    return verts, proj, mask, seg


def seg_streaming_fwd_bwd(proj, mask_vals, dseg, img_wh, part_ids, part_off, chunk=8):
    """projects_to_seg forward + its gradient w.r.t. `proj` WITHOUT the reference's materialised
    (N, W*W, n_p, 2) tiles or an autograd tape: per part a (W*W, n_p) distance block, min over the part's
    vertices (`exp` is monotone: max_v exp(-m d) = exp(-min_v m d), projects_to_seg.py:53-56), the score from the
    winner only, and the TF gradient rules written out (gradient to the arg-min vertex,
    background passes -upstream while 0 <= sum <= 1, 0 at d = 0).  The CPU baseline's "streaming" leg: the same
    arithmetic as the dense restatement, the algorithmic reformulation the GPU path also uses, none of its
    pruning.  -> (seg (N,W,W,32), dproj (N,V',3)); checked against the dense autograd form in tests."""
    N, W = proj.shape[0], img_wh
    npix = W * W
    grid = _grid(W, proj.dtype)                                        # (npix, 2), pixel q = r * W + c
    P = len(part_off) - 1
    best = torch.empty(N, npix, P, dtype=proj.dtype)
    arg = torch.empty(N, npix, P, dtype=torch.long)
    idxs = [torch.as_tensor(part_ids[part_off[p]:part_off[p + 1]], dtype=torch.long) for p in range(P)]
    for n0 in range(0, N, chunk):
        pr = proj[n0:n0 + chunk, :, :2]
        for p, idx in enumerate(idxs):
            pp = pr[:, idx, :]                                           # (c, n_p, 2)
            du = pp[:, None, :, 0] - grid[None, :, None, 0]             # (c, npix, n_p)
            dv = pp[:, None, :, 1] - grid[None, :, None, 1]
            x = torch.sqrt(du * du + dv * dv) * mask_vals[n0:n0 + chunk, idx][:, None, :]
            m, a = x.min(dim=2)
            best[n0:n0 + chunk, :, p] = m
            arg[n0:n0 + chunk, :, p] = idx[a]
    score = torch.exp(-best)                                           # (N, npix, P)
    ssum = score.sum(dim=2)
    seg = torch.cat([(1.0 - ssum.clamp(0.0, 1.0)).unsqueeze(2), score], dim=2).reshape(N, W, W, P + 1)
    seg = torch.flip(seg, dims=[1])
    # backward
    g = torch.flip(dseg, dims=[1]).reshape(N, npix, P + 1)
    gate = ((ssum >= 0.0) & (ssum <= 1.0)).to(proj.dtype).unsqueeze(2)
    gs = g[:, :, 1:] - gate * g[:, :, :1]                              # d loss / d score
    pw = torch.gather(proj[:, :, :2].unsqueeze(1).expand(N, npix, proj.shape[1], 2), 2,
                      arg.unsqueeze(3).expand(N, npix, P, 2))          # winners' positions (N, npix, P, 2)
    diff = pw - grid[None, :, None, :]
    d = torch.sqrt((diff * diff).sum(dim=3))
    mw = torch.gather(mask_vals.unsqueeze(1).expand(N, npix, mask_vals.shape[1]), 2, arg)
    k = torch.where(d > 0, -gs * score * mw / d.clamp_min(1e-30), torch.zeros_like(d))
    dproj = torch.zeros_like(proj)
    flat = arg.reshape(N, -1)
    for c in range(2):
        dproj[:, :, c].scatter_add_(1, flat, (k * diff[..., c]).reshape(N, -1))
    return seg, dproj


# --------------------------------------------------------------------------- loss head
def softmax_focal_loss(scores, y_true, gamma=2.0, class_w=None):
    """`model.py:119-120` + `focal_loss.py:10-46`, differentiable: raw scores (N, W, W, C) or
    (N, W*W, C), y_true one-hot/soft (N, W*W, C) -> per-pixel loss (N, W*W)."""
    s = scores.reshape(scores.shape[0], -1, scores.shape[-1])
    p = torch.softmax(s, dim=-1).clamp(1e-7, 1.0 - 1e-7)           # focal_loss.py:17
    ce = -y_true * torch.log(p)                                    # :18
    if class_w is not None:
        ce = ce * class_w                                          # :41
    return ((1.0 - p) ** gamma * ce).sum(dim=2)                    # :43-44
