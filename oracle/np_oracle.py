"""CPU oracle (float64 NumPy) for the SMPL decoder + soft-rasteriser hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `indirect_learning_pose-shape_amd/` may import this
file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and
only as the checker.

PARITY UNPINNED: the reference (akashsengupta1997/indirect_learning_pose-shape) has no tests,
fixtures or golden vectors for this path, its arithmetic lives in un-vendored TensorFlow 1.x /
Keras 2.1 (versions named only in README.md:19-27, no lock file) and it cannot be imported here
(`import cPickle`, `import tensorflow` raise ModuleNotFoundError; the SMPL pkl it opens is
absent).  This file is therefore a restatement written from the source text, op for op in the
reference's order, pinned by the analytic known-answer tests in `tests/test_oracle_kat.py`
and the golden vectors it generated itself (`tests/golden/`, `tools/make_golden.py`).  The DATA it reads
is the reference's own (part tables from keras_smpl/*part_vertices.pkl, cross-checked against the colour
classes of template-bodyparts.ply; mean pose / shape from neutral_smpl_mean_params.h5: tools/make_fixtures.py).

Every function cites the reference lines it follows (paths relative to the reference root).
"""
from __future__ import annotations

import numpy as np

F = np.float64


# --------------------------------------------------------------------------- SMPL decoder
def batch_skew(vec):
    """`keras_smpl/batch_smpl.py:230-253`: flat columns [1,2,3,5,6,7] <- [-z,y,z,-x,-y,x]."""
    vec = np.asarray(vec, F)
    n = vec.shape[0]
    res = np.zeros((n, 9), F)
    res[:, 1] = -vec[:, 2]
    res[:, 2] = vec[:, 1]
    res[:, 3] = vec[:, 2]
    res[:, 5] = -vec[:, 0]
    res[:, 6] = -vec[:, 1]
    res[:, 7] = vec[:, 0]
    return res.reshape(n, 3, 3)


def batch_rodrigues(theta):
    """`keras_smpl/batch_smpl.py:255-276`.

    angle = ||theta + 1e-8||  (:265),  r = theta / angle  (:266, NOT (theta+1e-8)/angle),
    R = cos*I + (1-cos)*r r^T + sin*skew(r)  (:269-275).
    """
    theta = np.asarray(theta, F)
    angle = np.sqrt(np.sum((theta + 1e-8) ** 2, axis=1))[:, None]       # (n,1)
    r = (theta / angle)[:, :, None]                                      # (n,3,1)
    angle = angle[:, :, None]                                            # (n,1,1)
    cos = np.cos(angle)
    sin = np.sin(angle)
    outer = r @ np.transpose(r, (0, 2, 1))
    eyes = np.tile(np.eye(3, dtype=F)[None], (theta.shape[0], 1, 1))
    return cos * eyes + (1 - cos) * outer + sin * batch_skew(r[:, :, 0])


def batch_global_rigid_transformation(Rs, Js, parent, rotate_base=False):
    """`keras_smpl/batch_smpl.py:168-228`; rotate_base (:185-190, never used by the reference's callers) multiplies the
    root rotation by diag(1, -1, -1) from the right.

    Returns (new_J (N,24,3), A (N,24,4,4)).
    """
    Rs = np.asarray(Rs, F)
    Js = np.asarray(Js, F)
    N = Rs.shape[0]
    if rotate_base:                                                      # :185-190
        Rs = Rs.copy()
        Rs[:, 0] = Rs[:, 0] @ np.diag([1.0, -1.0, -1.0])
    Js = Js[..., None]                                                   # (N,24,3,1)  :194

    def make_A(R, t):                                                    # :197-202
        R_homo = np.concatenate([R, np.zeros((N, 1, 3), F)], axis=1)     # (N,4,3)
        t_homo = np.concatenate([t, np.ones((N, 1, 1), F)], axis=1)      # (N,4,1)
        return np.concatenate([R_homo, t_homo], axis=2)                  # (N,4,4)

    results = [make_A(Rs[:, 0], Js[:, 0])]                               # :204-205
    for i in range(1, parent.shape[0]):                                  # :206-211
        j_here = Js[:, i] - Js[:, parent[i]]
        A_here = make_A(Rs[:, i], j_here)
        results.append(results[parent[i]] @ A_here)
    results = np.stack(results, axis=1)                                  # (N,24,4,4) :214
    new_J = results[:, :, :3, 3]                                         # :216
    Js_w0 = np.concatenate([Js, np.zeros((N, 24, 1, 1), F)], axis=2)     # :222
    init_bone = results @ Js_w0                                          # (N,24,4,1) :223
    init_bone = np.pad(init_bone, [[0, 0], [0, 0], [0, 0], [3, 0]])      # :225
    A = results - init_bone                                              # :226
    return new_J, A


def smpl_constants(model):
    """`SMPLLayer.build`, `keras_smpl/batch_smpl.py:31-93`: the matrices `call` multiplies by."""
    V = model.v_template.shape[0]
    return dict(
        v_template=np.asarray(model.v_template, F),                              # :38-41
        shapedirs=np.reshape(np.asarray(model.shapedirs, F), [-1, 10]).T.copy(),  # (10,3V) :50-55
        J_regressor=np.asarray(model.J_regressor, F).T.copy(),                   # (V,24) :58-61
        posedirs=np.reshape(np.asarray(model.posedirs, F), [-1, 207]).T.copy(),   # (207,3V) :64-68
        parents=np.asarray(model.parents, np.int32),                             # :71
        lbs_weights=np.asarray(model.weights, F),                                # :76-79
        V=V,
    )


def smpl_layer_call(x, model, num_cam=4, return_all=False):
    """`SMPLLayer.call`, `keras_smpl/batch_smpl.py:96-153`.  x: (N, num_cam+72+10)."""
    c = smpl_constants(model)
    x = np.asarray(x, F)
    N, V = x.shape[0], c["V"]
    thetas = x[:, num_cam:num_cam + 72]                                  # :98
    betas = x[:, num_cam + 72:]                                          # :99
    v_shaped = (betas @ c["shapedirs"]).reshape(-1, V, 3) + c["v_template"]      # :106-108
    Jx = v_shaped[:, :, 0] @ c["J_regressor"]                            # :112-114
    Jy = v_shaped[:, :, 1] @ c["J_regressor"]
    Jz = v_shaped[:, :, 2] @ c["J_regressor"]
    J = np.stack([Jx, Jy, Jz], axis=2)                                   # :115
    Rs = batch_rodrigues(thetas.reshape(-1, 3)).reshape(-1, 24, 3, 3)    # :119-120
    pose_feature = (Rs[:, 1:] - np.eye(3, dtype=F)).reshape(-1, 207)     # :122
    v_posed = (pose_feature @ c["posedirs"]).reshape(-1, V, 3) + v_shaped        # :126-128
    J_transformed, A = batch_global_rigid_transformation(Rs, J, c["parents"])    # :131
    W = np.tile(c["lbs_weights"], (N, 1)).reshape(N, -1, 24)             # :135-136
    T = (W @ A.reshape(N, 24, 16)).reshape(N, -1, 4, 4)                  # :138-140
    v_posed_homo = np.concatenate([v_posed, np.ones((N, V, 1), F)], axis=2)      # :141-142
    v_homo = T @ v_posed_homo[..., None]                                 # :143
    verts = v_homo[:, :, :3, 0]                                          # :145
    if return_all:
        return dict(verts=verts, J_transformed=J_transformed, A=A, Rs=Rs, J=J,
                    v_shaped=v_shaped, v_posed=v_posed, pose_feature=pose_feature)
    return verts


# --------------------------------------------------------------------------- projection
def orthographic_project(verts, smpl, vertex_sampling=None):
    """`keras_smpl/projection.py:54-81`: (u,v,z) = (u0 + x*k_u, v0 + y*k_v, z)."""
    verts = np.asarray(verts, F)
    smpl = np.asarray(smpl, F)
    k_u, k_v, u0, v0 = smpl[:, 0], smpl[:, 1], smpl[:, 2], smpl[:, 3]    # :62-65
    if vertex_sampling is not None:
        verts = verts[:, ::vertex_sampling, :]                           # :67-68
    u = u0[:, None] + verts[:, :, 0] * k_u[:, None]                      # :77
    v = v0[:, None] + verts[:, :, 1] * k_v[:, None]                      # :78
    return np.stack([u, v, verts[:, :, 2]], axis=2)                      # :79


# --------------------------------------------------------------------------- visibility
def compute_mask(projects_with_depth, grid_wh=64, ref_compat=True):
    """Stateless semantics of `keras_smpl/compute_mask.py:12-108`.

    Per sample: round (u,v) half-to-even (:22); for every pixel (c,r) of a fixed 64x64 grid
    (:44, meshgrid 'xy' :49-54) the candidates are the vertices whose rounded pixel equals it
    exactly (:90-92); the winner is the candidate of LARGEST z, lowest index on ties
    (`tf.argmax`, :98-103); an empty pixel yields the constant ones([1,1,4]) whose index field
    is 1 (:99), i.e. vertex 1 is marked visible whenever any pixel is empty.  mask = 500
    everywhere, 1 at winners (:65-70).  The reference's persistent, never-reset mask variable
    (:68-70, raced by the parallel map_fn :27-30) is NOT reproduced.
    """
    p = np.asarray(projects_with_depth, F)
    B, V = p.shape[0], p.shape[1]
    masks = np.full((B, V), 500.0, F)
    idx = np.arange(V)
    for n in range(B):
        pu = np.rint(p[n, :, 0])
        pv = np.rint(p[n, :, 1])
        z = p[n, :, 2]
        winners = set()
        any_empty = False
        for r in range(grid_wh):                     # pixel_coords order: (x=c, y=r)
            row = pv == float(r)
            if not row.any():
                any_empty = True
                continue
            cand_r = idx[row]
            for c in range(grid_wh):
                cand = cand_r[pu[cand_r] == float(c)]
                if cand.size == 0:
                    any_empty = True
                else:
                    winners.add(int(cand[np.argmax(z[cand])]))   # first max = lowest index
        if any_empty and ref_compat and V > 1:
            winners.add(1)
        masks[n, sorted(winners)] = 1.0
    return masks


def compute_mask_sorted(projects_with_depth, grid_wh=64, ref_compat=True):
    """The same stateless definition as `compute_mask` (compute_mask.py:12-108) without the Python loop over the
    4096 grid cells: one lexicographic sort per sample by (cell, -z, index); the first vertex of every cell group
    is its arg-max-z, lowest index on ties (`tf.argmax`, :98-103).  Used by the timed CPU baseline so that it does
    not time the interpreter; pinned to `compute_mask` in tests/test_oracle_kat.py."""
    p = np.asarray(projects_with_depth, F)
    B, V = p.shape[0], p.shape[1]
    masks = np.full((B, V), 500.0, F)
    idx = np.arange(V)
    for n in range(B):
        pu, pv = np.rint(p[n, :, 0]), np.rint(p[n, :, 1])               # :22
        inside = (pu >= 0) & (pu < grid_wh) & (pv >= 0) & (pv < grid_wh)
        ii = idx[inside]
        cell = (pv[inside] * grid_wh + pu[inside]).astype(np.int64)
        z = p[n, inside, 2] + 0.0                                        # -0.0 ties with +0.0
        order = np.lexsort((ii, -z, cell))                               # last key is the primary one
        cs = cell[order]
        first = np.ones(cs.shape[0], bool)
        first[1:] = cs[1:] != cs[:-1]
        masks[n, ii[order][first]] = 1.0
        if int(first.sum()) < grid_wh * grid_wh and ref_compat and V > 1:   # an empty cell: vertex 1 (:99)
            masks[n, 1] = 1.0
    return masks


# --------------------------------------------------------------------------- rasterisers
def _grid(img_wh):
    """`projects_to_seg.py:26-31`: pixel q = r*W + c has coordinate (x=c, y=r)."""
    t1, t2 = np.meshgrid(np.arange(img_wh), np.arange(img_wh))
    return np.stack([t1, t2], axis=2).astype(F).reshape(-1, 2)


def projects_to_seg(projects_with_depth, mask_vals, img_wh, part_ids, part_off,
                    vertex_sampling=None, return_argmin=False):
    """`keras_smpl/projects_to_seg.py:9-69`.

    part_ids/part_off: ORIGINAL vertex ids, part-major CSR (the pkl lists, :18-24); with
    vertex sampling they are mapped by `index // vertex_sampling` (:36-37).
    Output (N, W, W, 32): channel 0 = 1 - clip(sum_p, 0, 1) (:61-64), channels 1..31 the part
    scores max_v exp(-mask_v * ||proj_v - q||) (:41-56), rows flipped (:68).
    """
    proj = np.asarray(projects_with_depth, F)[:, :, :2]                  # :13
    mask = np.asarray(mask_vals, F)
    N = proj.shape[0]
    grid = _grid(img_wh)                                                 # (W*W,2)
    segs, args = [], []
    nparts = len(part_off) - 1
    for part in range(nparts):                                           # :34
        indices = np.asarray(part_ids[part_off[part]:part_off[part + 1]], np.int64)
        if vertex_sampling is not None:
            indices = indices // vertex_sampling                         # :36-37
        part_projects = proj[:, indices, :]                              # (N,n,2) :41
        part_mask = mask[:, indices]                                     # (N,n)   :45
        diff = part_projects[:, None, :, :] - grid[None, :, None, :]     # (N,W2,n,2) :52
        norm = np.sqrt(np.sum(diff * diff, axis=3))                      # :53
        norm = norm * part_mask[:, None, :]                              # :54
        e = np.exp(-norm)                                                # :55
        segs.append(e.max(axis=2).reshape(N, img_wh, img_wh))            # :56-58
        if return_argmin:
            args.append(indices[np.argmin(norm, axis=2)].reshape(N, img_wh, img_wh))
    stacked = np.stack(segs, axis=3)                                     # :60
    sil = 1.0 - np.clip(stacked.sum(axis=3), 0.0, 1.0)                   # :61-64
    out = np.concatenate([sil[..., None], stacked], axis=3)              # :66-67
    out = out[:, ::-1]                                                   # :68
    if return_argmin:
        return out, np.stack(args, axis=3)[:, ::-1]
    return out


def projects_to_silhouette(projects_with_depth, img_wh):
    """`keras_smpl/projects_to_silhouette.py:14-44`: all vertices, no mask, exp(-d/1.2)."""
    proj = np.asarray(projects_with_depth, F)[:, :, :2]                  # :20
    N = proj.shape[0]
    grid = _grid(img_wh)
    out = np.empty((N, img_wh * img_wh), F)
    for n in range(N):                                                   # batch loop: memory only
        diff = proj[n][None, :, :] - grid[:, None, :]                    # (W2,V,2) :35
        norm = np.sqrt(np.sum(diff * diff, axis=2))                      # :36
        out[n] = np.exp(-norm / 1.2).max(axis=1)                         # :37-38
    sil = out.reshape(N, img_wh, img_wh)                                 # :39
    res = np.stack([1.0 - sil, sil], axis=3)                             # :40-41
    return res[:, ::-1]                                                  # :42


# --------------------------------------------------------------------------- conditioning
def set_cam_params(smpl, img_wh):
    """`keras_smpl/set_cam_params.py:13-26`."""
    cam = np.zeros((1, 86), F)
    cam[0, 0] = img_wh / 2.0
    cam[0, 1] = img_wh / 2.0
    cam[0, 2] = img_wh / 2.0
    cam[0, 3] = img_wh / 1.6
    return np.asarray(smpl, F) + cam.astype(np.float32).astype(F)


def _mean_row(img_wh, mean_pose, mean_shape):
    mean = np.zeros((1, 86), F)
    mean[0, 0] = img_wh / 2.0
    mean[0, 1] = img_wh / 2.0
    mean[0, 2] = img_wh / 2.0
    mean[0, 3] = img_wh / 1.6
    pose = np.array(mean_pose, F).copy()
    pose[:3] = 0.0                                                       # :44-45 / :13-14
    mean[0, 4:] = np.hstack((pose, np.asarray(mean_shape, F)))
    return mean.astype(np.float32).astype(F)                             # tf.constant(..., float32)


def load_mean_set_cam_params(smpl, img_wh, mean_pose, mean_shape):
    """`keras_smpl/set_cam_params.py:29-51`."""
    return np.asarray(smpl, F) + _mean_row(img_wh, mean_pose, mean_shape)


def concat_mean_param(img_features, img_wh, mean_pose, mean_shape):
    """`keras_smpl/concat_mean_param.py:8-31`: state = [features | mean row]."""
    f = np.asarray(img_features, F)
    mean = np.tile(_mean_row(img_wh, mean_pose, mean_shape), (f.shape[0], 1))
    return np.concatenate([f, mean], axis=1)


# --------------------------------------------------------------------------- loss head
FOCAL_CLASS_WEIGHTS = np.ones(32, F)                     # `focal_loss.py:22-40`
FOCAL_CLASS_WEIGHTS[0] = 0.3
FOCAL_CLASS_WEIGHTS[[1, 2, 3, 4, 10, 12, 14, 15, 16, 17, 23, 25]] = 2.0
K_EPSILON = 1e-7                                         # keras.backend.epsilon()


def softmax_last(scores):
    """`model.py:119-120`: Reshape((W*W, C)) + Activation('softmax') over the class axis."""
    s = np.asarray(scores, F)
    s = s.reshape(s.shape[0], -1, s.shape[-1])
    e = np.exp(s - s.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def categorical_focal_loss(y_true, y_pred, gamma=2.0, weight_classes=False):
    """`focal_loss.py:10-46`: y_true, y_pred (N, W*W, C) -> per-pixel loss (N, W*W)."""
    y_true, y_pred = np.asarray(y_true, F), np.asarray(y_pred, F)
    p = np.clip(y_pred, K_EPSILON, 1.0 - K_EPSILON)               # :17
    ce = -y_true * np.log(p)                                       # :18
    if weight_classes:
        ce = ce * FOCAL_CLASS_WEIGHTS[: p.shape[-1]]               # :20-41
    return (np.power(1.0 - p, gamma) * ce).sum(axis=2)             # :43-44


def categorical_crossentropy(y_true, y_pred):
    """Keras 2.1 `categorical_crossentropy` on probabilities (the silhouette head's loss,
    `train_stage2_silhouette.py:226-229`): rescale to row sum 1, clip, -sum(y log p)."""
    y_true, y_pred = np.asarray(y_true, F), np.asarray(y_pred, F)
    p = y_pred / y_pred.sum(axis=-1, keepdims=True)
    p = np.clip(p, K_EPSILON, 1.0 - K_EPSILON)
    return -(y_true * np.log(p)).sum(axis=-1)
