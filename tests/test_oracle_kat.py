"""Analytic known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors for this path ("parity unpinned"), so the oracle
is pinned by closed-form identities of the SMPL function and of the rasterisers.
"""
import numpy as np
import pytest
import torch

from _inputs import make_x
from oracle import np_oracle as o
from oracle import torch_oracle as to


def test_rodrigues_known_matrix():
    R = o.batch_rodrigues(np.array([[np.pi / 2, 0, 0], [0, 0, np.pi / 2], [0, 0, 0]]))
    assert np.allclose(R[0], [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-7)
    assert np.allclose(R[1], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-7)
    assert np.allclose(R[2], np.eye(3), atol=1e-12)          # theta = 0 -> r = 0, R = I


def test_rodrigues_orthonormal_and_skew():
    rng = np.random.default_rng(0)
    t = rng.normal(0, 1, (50, 3))
    R = o.batch_rodrigues(t)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-6)   # 1e-8 offset in angle only
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-6)
    K = o.batch_skew(t)
    assert np.allclose(K, -K.transpose(0, 2, 1))
    v = rng.normal(0, 1, (50, 3))
    assert np.allclose((K @ v[..., None])[..., 0], np.cross(t, v))


def test_theta_zero(smpl_model):
    rng = np.random.default_rng(1)
    x = np.zeros((3, 86))
    x[:, 76:] = rng.normal(0, 1, (3, 10))
    r = o.smpl_layer_call(x, smpl_model, return_all=True)
    S = smpl_model.shapedirs.reshape(-1, 10).T
    want = (x[:, 76:] @ S).reshape(3, -1, 3) + smpl_model.v_template
    assert np.abs(r["verts"] - want).max() < 1e-12
    assert np.abs(r["J_transformed"] - r["J"]).max() < 1e-12
    A = r["A"]
    assert np.allclose(A[:, :, :3, :3], np.eye(3), atol=1e-12) and np.allclose(A[:, :, :3, 3], 0, atol=1e-12)


def test_global_rotation_only(smpl_model):
    """Only theta[0:3] set: pose blend term vanishes and verts = R0 (v_shaped - J0) + J0."""
    rng = np.random.default_rng(2)
    x = np.zeros((2, 86))
    x[:, 4:7] = rng.normal(0, 0.7, (2, 3))
    x[:, 76:] = rng.normal(0, 1, (2, 10))
    r = o.smpl_layer_call(x, smpl_model, return_all=True)
    R0 = o.batch_rodrigues(x[:, 4:7])
    J0 = r["J"][:, 0]
    want = np.einsum("nij,nvj->nvi", R0, r["v_shaped"] - J0[:, None]) + J0[:, None]
    assert np.abs(r["pose_feature"]).max() < 1e-12
    assert np.abs(r["verts"] - want).max() < 1e-10


def test_A_is_relative_transform(smpl_model):
    """A = [R_g | t_g - R_g J] (batch_smpl.py:222-226) and rows 3 of A are [0,0,0,1]-[0,0,0,0]."""
    x = make_x(2, 48, 3).astype(np.float64)
    r = o.smpl_layer_call(x, smpl_model, return_all=True)
    A, J, Jt = r["A"], r["J"], r["J_transformed"]
    assert np.allclose(A[:, :, :3, 3], Jt - np.einsum("njab,njb->nja", A[:, :, :3, :3], J), atol=1e-12)
    assert np.allclose(A[:, :, 3], [0, 0, 0, 1])
    # partition of unity: a posed joint is the LBS image of its rest position under its own A
    assert np.allclose(np.einsum("njab,njb->nja", A[:, :, :3, :3], J) + A[:, :, :3, 3], Jt, atol=1e-12)


def test_batch_permutation_equivariance(smpl_model):
    x = make_x(4, 48, 4).astype(np.float64)
    perm = np.array([2, 0, 3, 1])
    assert np.array_equal(o.smpl_layer_call(x, smpl_model)[perm], o.smpl_layer_call(x[perm], smpl_model))


def test_torch_oracle_matches_numpy_and_fd(smpl_model, part_tables):
    x = make_x(2, 48, 5).astype(np.float64)
    smpl = to.TorchSMPL(smpl_model)
    xt = torch.tensor(x, requires_grad=True)
    v, jt, A = smpl(xt, return_all=True)
    ref = o.smpl_layer_call(x, smpl_model, return_all=True)
    assert np.abs(v.detach().numpy() - ref["verts"]).max() < 1e-12
    assert np.abs(A.detach().numpy() - ref["A"]).max() < 1e-12
    rng = np.random.default_rng(6)
    w = rng.normal(0, 1, ref["verts"].shape)
    (v * torch.tensor(w)).sum().backward()
    g = xt.grad.numpy()
    assert np.all(g[:, :4] == 0)                       # the camera does not enter SMPLLayer
    f = lambda xx: float((o.smpl_layer_call(xx, smpl_model) * w).sum())
    for col in (4, 5, 17, 40, 75, 76, 80, 85):
        e = np.zeros_like(x)
        e[0, col] = 1e-6
        fd = (f(x + e) - f(x - e)) / 2e-6
        assert abs(fd - g[0, col]) <= 1e-5 * max(1.0, abs(fd)), (col, fd, g[0, col])


def test_projection(smpl_model):
    rng = np.random.default_rng(7)
    verts = rng.normal(0, 1, (2, 6890, 3))
    x = make_x(2, 48, 8).astype(np.float64)
    p = o.orthographic_project(verts, x)
    assert np.allclose(p[..., 0], x[:, 2:3] + verts[..., 0] * x[:, 0:1])
    assert np.allclose(p[..., 1], x[:, 3:4] + verts[..., 1] * x[:, 1:2])
    assert np.array_equal(p[..., 2], verts[..., 2])
    for vs, n in ((2, 3445), (5, 1378)):
        q = o.orthographic_project(verts, x, vs)
        assert q.shape == (2, n, 3) and np.array_equal(q, p[:, ::vs])
    pt = to.orthographic_project(torch.tensor(verts), torch.tensor(x), 5).numpy()
    assert np.allclose(pt, p[:, ::5])


def test_mask_kats():
    p = np.full((1, 8, 3), 1000.0)
    p[0, 2] = [3.2, 4.4, 0.5]
    p[0, 3] = [2.6, 3.7, 0.9]        # same cell (3,4), larger z wins (arg-MAX, compute_mask.py:101)
    p[0, 4] = [10.5, 0.0, 0.1]       # half-to-even: 10
    p[0, 5] = [10.0, 0.0, 0.1]       # tie on z -> lowest index
    p[0, 6] = [11.5, 0.0, 0.1]       # -> 12
    p[0, 7] = [63.6, 5.0, 0.1]       # -> 64: outside the 64-grid
    assert o.compute_mask(p)[0].tolist() == [500, 1, 500, 1, 1, 500, 1, 500]
    assert o.compute_mask(p, ref_compat=False)[0].tolist() == [500, 500, 500, 1, 1, 500, 1, 500]
    # a full grid has no empty pixel -> no vertex-1 artefact
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4)), -1).reshape(-1, 2).astype(float)
    full = np.concatenate([g, np.zeros((16, 1))], 1)[None]
    assert o.compute_mask(full, grid_wh=4)[0].tolist() == [1.0] * 16


def test_seg_kats(part_tables):
    ids, off = part_tables[1]
    W = 16
    p = np.full((1, 6890, 3), 1e4)
    mask = np.ones((1, 6890))
    v7 = int(ids[off[7]])
    p[0, v7] = [5.0, 9.0, 0.0]
    v9 = int(ids[off[9] + 3])
    p[0, v9] = [12.1, 3.0, 0.0]
    mask[0, v9] = 500.0
    un = sorted(set(range(6890)) - set(ids.tolist()))
    assert len(un) == 11
    p[0, un[0]] = [8.0, 8.0, 0.0]
    seg, arg = o.projects_to_seg(p, mask, W, ids, off, return_argmin=True)
    assert seg.shape == (1, W, W, 32)
    img = seg[0, ::-1]
    assert img[9, 5, 8] == 1.0 and arg[0, ::-1][9, 5, 7] == v7
    assert np.isclose(img[9, 8, 8], np.exp(-3.0)) and np.isclose(img[13, 5, 8], np.exp(-4.0))
    assert np.isclose(img[3, 12, 10], np.exp(-500 * 0.1), rtol=1e-9)
    assert img[8, 8, 1:].max() <= np.exp(-3.1)            # the unassigned vertex paints nothing
    s = img[..., 1:].sum(-1)
    assert np.allclose(img[..., 0], 1 - np.clip(s, 0, 1))
    t = to.projects_to_seg(torch.tensor(p), torch.tensor(mask), W, ids, off).numpy()
    assert np.allclose(t, seg, atol=1e-14)


def test_seg_gradient_rules(part_tables):
    """Gradient reaches only the arg-min vertex; background path is gated by sum <= 1."""
    ids, off = part_tables[1]
    W = 8
    p = np.full((1, 6890, 3), 1e4)
    a, b = int(ids[off[0]]), int(ids[off[0] + 1])
    p[0, a] = [2.0, 2.3, 0.0]
    p[0, b] = [6.0, 5.6, 0.0]
    pt = torch.tensor(p, requires_grad=True)
    seg = to.projects_to_seg(pt, torch.ones(1, 6890, dtype=torch.float64), W, ids, off)
    seg[0, W - 1 - 2, 2, 1].backward()               # pixel (c=2, r=2), part 0: nearest is vertex a
    g = pt.grad[0]
    assert g[b].abs().sum() == 0 and g[a, 2] == 0
    d = 0.3
    assert np.isclose(g[a, 1].item(), -np.exp(-d) * (2.3 - 2.0) / d) and abs(g[a, 0].item()) < 1e-12
    pt.grad = None
    seg = to.projects_to_seg(pt, torch.ones(1, 6890, dtype=torch.float64), W, ids, off)
    seg[0, W - 1 - 2, 2, 0].backward()               # background: -d(sum)/dp where 0 <= sum <= 1
    assert np.isclose(pt.grad[0, a, 1].item(), +np.exp(-d) * (2.3 - 2.0) / d)


def test_silhouette_kats():
    W = 8
    p = np.full((1, 6890, 3), 1e4)
    p[0, 4000] = [3.0, 4.0, 7.0]
    s = o.projects_to_silhouette(p, W)
    img = s[0, ::-1]
    assert np.allclose(img.sum(-1), 1.0)
    assert img[4, 3, 1] == 1.0 and np.isclose(img[4, 6, 1], np.exp(-3.0 / 1.2))
    t = to.projects_to_silhouette(torch.tensor(p), W).numpy()
    assert np.allclose(t, s, atol=1e-14)


def test_conditioning():
    from ilps_amd.smpl_model import load_mean_params
    pose, shape = load_mean_params()
    z = np.zeros((2, 86))
    a = o.set_cam_params(z, 48)
    assert a[0, :4].tolist() == [24.0, 24.0, 24.0, 30.0] and np.all(a[:, 4:] == 0)
    b = o.load_mean_set_cam_params(z, 64, pose, shape)
    assert b[0, :4].tolist() == [32.0, 32.0, 32.0, 40.0]
    assert np.all(b[0, 4:7] == 0) and np.isclose(b[0, 76], 0.20560974, atol=1e-7)
    c = o.concat_mean_param(np.ones((2, 2048)), 48, pose, shape)
    assert c.shape == (2, 2134) and np.array_equal(c[:, 2048:], o.load_mean_set_cam_params(z, 48, pose, shape))


# ----------------------------------------------------------------------------------- loss head
def test_focal_loss_uniform_scores_kat():
    """Equal scores => softmax = 1/C; loss = w_t (1 - 1/C)^gamma log C at the labelled class."""
    N, W, C = 2, 3, 32
    y = np.zeros((N, W * W, C))
    lab = np.arange(N * W * W).reshape(N, W * W) % C
    np.put_along_axis(y, lab[..., None], 1.0, axis=2)
    p = o.softmax_last(np.full((N, W, W, C), 0.37))
    assert np.allclose(p, 1.0 / C)
    got = o.categorical_focal_loss(y, p, 2.0, True)
    want = o.FOCAL_CLASS_WEIGHTS[lab] * (1 - 1 / C) ** 2 * np.log(C)
    assert np.allclose(got, want, rtol=1e-12)
    assert np.allclose(o.categorical_focal_loss(y, p, 0.0, False), np.log(C))     # plain CE
    assert np.allclose(o.categorical_crossentropy(y, p), np.log(C))


def test_focal_loss_clip_kat():
    """A confident wrong pixel is clipped at eps: loss = (1-eps)^2 * -log(eps); a confident right
    one at 1-eps: loss = eps^2 * -log(1-eps) ~ 1e-21."""
    s = np.zeros((1, 1, 2, 32))
    s[0, 0, :, 5] = 100.0
    p = o.softmax_last(s)
    y = np.zeros((1, 2, 32))
    y[0, 0, 5] = 1.0      # right
    y[0, 1, 7] = 1.0      # wrong
    got = o.categorical_focal_loss(y, p, 2.0, False)[0]
    assert np.isclose(got[0], 1e-14 * -np.log(1 - 1e-7), rtol=1e-6)
    assert np.isclose(got[1], (1 - 1e-7) ** 2 * -np.log(1e-7), rtol=1e-12)


def test_focal_torch_oracle_matches_numpy():
    rng = np.random.default_rng(5)
    s = rng.uniform(0, 1, (2, 4, 4, 32))
    y = np.eye(32)[rng.integers(0, 32, (2, 16))]
    a = o.categorical_focal_loss(y, o.softmax_last(s), 2.0, True)
    b = to.softmax_focal_loss(torch.tensor(s), torch.tensor(y), 2.0, torch.tensor(o.FOCAL_CLASS_WEIGHTS)).numpy()
    assert np.allclose(a, b, rtol=1e-12)


def test_sorted_mask_equals_loop_mask(smpl_model):
    """The vectorised visibility used by the timed CPU baseline == the cell-by-cell restatement."""
    rng = np.random.default_rng(3)
    p = (rng.random((3, 6890, 3)) * 80 - 8).astype(np.float32).astype(np.float64)   # collisions, out-of-grid, ties
    p[0, 100:140, 2] = p[0, 100, 2]                         # equal depths
    p[0, 100:140, :2] = [10.2, 20.4]
    p[1, 5, 2], p[1, 6, 2], p[1, 5, :2], p[1, 6, :2] = 0.0, -0.0, [3.0, 3.0], [3.0, 3.0]
    for rc in (True, False):
        assert np.array_equal(o.compute_mask_sorted(p, ref_compat=rc), o.compute_mask(p, ref_compat=rc))
    dense = np.zeros((1, 4096, 3))                          # every cell occupied -> no vertex-1 artefact
    dense[0, :, 0], dense[0, :, 1] = np.tile(np.arange(64), 64), np.repeat(np.arange(64), 64)
    dense[0, :, 2] = np.arange(4096) % 7
    assert np.array_equal(o.compute_mask_sorted(dense), o.compute_mask(dense))
    assert (o.compute_mask_sorted(dense) == 1.0).all()


def test_streaming_seg_equals_dense_autograd(smpl_model, part_tables):
    """The non-materialising seg forward + hand-written gradient (the CPU baseline's streaming leg) against the
    dense restatement under autograd, float64."""
    import torch
    from oracle import torch_oracle as to
    ids, off = part_tables[1]
    W = 24
    rng = np.random.default_rng(12)
    proj = torch.tensor(rng.random((2, 6890, 3)) * (W + 6) - 3, dtype=torch.float64)
    mask = torch.tensor(np.where(rng.random((2, 6890)) < 0.1, 1.0, 500.0))
    g = torch.tensor(rng.normal(0, 1, (2, W, W, 32)))
    pd = proj.clone().requires_grad_(True)
    dense = to.projects_to_seg(pd, mask, W, ids, off)
    (dense * g).sum().backward()
    seg, dproj = to.seg_streaming_fwd_bwd(proj, mask, g, W, ids, off)
    assert torch.allclose(seg, dense.detach(), rtol=1e-12, atol=1e-300)
    assert torch.allclose(dproj, pd.grad, rtol=1e-9, atol=1e-14)


def test_rotate_base_kat(smpl_model):
    """rotate_base (batch_smpl.py:185-190): the root rotation times diag(1, -1, -1); every world transform of the tree
    is that of the unrotated chain with its rotation part multiplied from the... root side: G'_i = G_0' G_0^-1 G_i, so
    with theta = 0 (all R = I) the posed joints are the rest joints mirrored in y and z about the root joint."""
    J = np.random.default_rng(0).normal(0, 0.3, (2, 24, 3))
    Rs = np.tile(np.eye(3), (2, 24, 1, 1))
    newJ, A = o.batch_global_rigid_transformation(Rs, J, np.asarray(smpl_model.parents), rotate_base=True)
    want = J[:, :1] + (J - J[:, :1]) * np.array([1.0, -1.0, -1.0])
    assert np.abs(newJ - want).max() < 1e-12
    assert np.abs(A[:, :, :3, :3] - np.diag([1.0, -1.0, -1.0])).max() < 1e-12
    newJ0, _ = o.batch_global_rigid_transformation(Rs, J, np.asarray(smpl_model.parents))
    assert np.abs(newJ0 - J).max() < 1e-12
