"""SMPL constants for the decoder: a seeded synthetic SMPL-shaped model and the part tables.

The reference loads its constants from a licence-restricted pickle
(`keras_smpl/batch_smpl.py:31-93`) that is not shipped.  This module provides

* `SMPLModelData`      - the arrays `SMPLLayer.build` extracts, in the pkl's own layout,
* `synthetic_smpl_model` - a deterministic generator with the shapes, sparsity and
  magnitudes of the real model, used by tests and `bench.py`,
* `load_part_tables`   - the 31 body-part vertex lists read at `projects_to_seg.py:18-24`
  (converted to CSR by `tools/make_fixtures.py`),
* `load_mean_params`   - the 82 mean pose/shape numbers of `neutral_smpl_mean_params.h5`
  (`set_cam_params.py:41-47`).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207
NUM_PARTS = 31

# Public SMPL kinematic tree (`kintree_table[0]`, root's uint32-max read as -1;
# `batch_smpl.py:71`; the loop at :206 never indexes the root's parent).
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int32)

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@dataclass
class SMPLModelData:
    """Arrays in the SMPL pkl's layout (float64 on the host)."""
    v_template: np.ndarray          # (V,3)
    shapedirs: np.ndarray           # (V,3,10)
    posedirs: np.ndarray            # (V,3,207)
    J_regressor: np.ndarray         # (24,V) dense (pkl: scipy sparse)
    weights: np.ndarray             # (V,24)
    parents: np.ndarray = field(default_factory=lambda: SMPL_PARENTS.copy())
    cocoplus_regressor: Optional[np.ndarray] = None   # (19,V), unused by the hot path

    @property
    def num_verts(self) -> int:
        return int(self.v_template.shape[0])

    def validate(self) -> None:
        V = self.num_verts
        if self.v_template.shape != (V, 3):
            raise ValueError("v_template must be (V,3)")
        if self.shapedirs.shape != (V, 3, NUM_BETAS):
            raise ValueError("shapedirs must be (V,3,10)")
        if self.posedirs.shape != (V, 3, NUM_POSE_BASIS):
            raise ValueError("posedirs must be (V,3,207)")
        if self.J_regressor.shape != (NUM_JOINTS, V):
            raise ValueError("J_regressor must be (24,V)")
        if self.weights.shape != (V, NUM_JOINTS):
            raise ValueError("weights must be (V,24)")
        if self.parents.shape != (NUM_JOINTS,):
            raise ValueError("parents must be (24,)")
        for i in range(1, NUM_JOINTS):
            if not (0 <= int(self.parents[i]) < i):
                raise ValueError("parents must be topologically ordered (parent[i] < i)")


# Rest-pose joint centres of a 1.7 m body, y up, origin near the chest (metres).
_REST_JOINTS = np.array([
    [0.00, -0.24, 0.03], [0.06, -0.33, 0.02], [-0.06, -0.33, 0.02], [0.00, -0.13, -0.01],
    [0.10, -0.71, 0.02], [-0.10, -0.71, 0.02], [0.00, 0.01, 0.02], [0.09, -1.10, -0.03],
    [-0.09, -1.10, -0.03], [0.00, 0.06, 0.04], [0.11, -1.16, 0.09], [-0.11, -1.16, 0.09],
    [0.00, 0.27, -0.01], [0.08, 0.18, 0.00], [-0.08, 0.18, 0.00], [0.00, 0.36, 0.04],
    [0.18, 0.23, -0.01], [-0.18, 0.23, -0.01], [0.44, 0.22, -0.03], [-0.44, 0.22, -0.03],
    [0.69, 0.22, -0.03], [-0.69, 0.22, -0.03], [0.78, 0.21, -0.04], [-0.78, 0.21, -0.04],
], dtype=np.float64)


def synthetic_smpl_model(seed: int = 1234, num_verts: int = NUM_VERTS) -> SMPLModelData:
    """Seeded SMPL-shaped model.

    Vertices of body part p are scattered around a point on a bone of the rest skeleton so
    that projected parts are spatially coherent; skinning rows are >=0, sum to 1 and have at
    most 4 non-zeros; J_regressor rows are >=0, sparse and sum to 1 (Appendix A.1).
    """
    rng = np.random.default_rng(seed)
    V = num_verts
    parents = SMPL_PARENTS.copy()

    # vertex -> part (-1 = the vertices that belong to no part)
    part_of = np.full(V, -1, dtype=np.int64)
    if V == NUM_VERTS:
        ids, off = load_part_tables(1)
        for p in range(NUM_PARTS):
            part_of[ids[off[p]:off[p + 1]]] = p
    else:
        part_of[:] = np.arange(V) % NUM_PARTS

    # part -> (bone child joint, position along the bone)
    child = 1 + (np.arange(NUM_PARTS) % (NUM_JOINTS - 1))
    frac = 0.25 + 0.5 * (np.arange(NUM_PARTS) // (NUM_JOINTS - 1))
    anchor_joint = np.where(part_of >= 0, child[np.clip(part_of, 0, None)], 9)
    t = np.where(part_of >= 0, frac[np.clip(part_of, 0, None)], 0.5)
    a = _REST_JOINTS[parents[anchor_joint]]
    b = _REST_JOINTS[anchor_joint]
    centre = a + (b - a) * t[:, None]
    v_template = centre + rng.normal(0.0, 0.035, size=(V, 3))

    # skinning weights: anchor joint, its parent, and up to two further joints
    weights = np.zeros((V, NUM_JOINTS))
    for v in range(V):
        j0 = int(anchor_joint[v])
        cand = [j0, int(parents[j0])]
        kids = np.nonzero(parents == j0)[0]
        if kids.size:
            cand.append(int(kids[rng.integers(kids.size)]))
        if parents[cand[1]] >= 0:
            cand.append(int(parents[cand[1]]))
        cand = list(dict.fromkeys(cand))[:4]
        w = rng.dirichlet(np.r_[4.0, np.ones(len(cand) - 1)])
        weights[v, cand] = w
    weights /= weights.sum(axis=1, keepdims=True)

    # joint regressor: each joint from ~20 nearby vertices, convex weights
    J_regressor = np.zeros((NUM_JOINTS, V))
    for j in range(NUM_JOINTS):
        d = np.linalg.norm(v_template - _REST_JOINTS[j], axis=1)
        near = np.argsort(d)[:20]
        J_regressor[j, near] = rng.dirichlet(np.ones(near.size))

    shapedirs = rng.normal(0.0, 1.0, size=(V, 3, NUM_BETAS)) * \
        np.geomspace(0.03, 0.004, NUM_BETAS)[None, None, :]
    posedirs = rng.normal(0.0, 0.004, size=(V, 3, NUM_POSE_BASIS))
    cocoplus = np.zeros((19, V))
    for k in range(19):
        near = rng.choice(V, size=8, replace=False)
        cocoplus[k, near] = rng.dirichlet(np.ones(8))

    m = SMPLModelData(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs,
                      J_regressor=J_regressor, weights=weights, parents=parents,
                      cocoplus_regressor=cocoplus)
    m.validate()
    return m


_part_cache = {}


def load_part_tables(vertex_sampling: Optional[int] = None):
    """Return (ids, offsets): ORIGINAL vertex ids in part-major order + 32 CSR offsets.

    `vertex_sampling` in {None,1,2,5} selects the table the reference opens at
    `projects_to_seg.py:18-21`.  Positions in the strided vertex list are `ids // vs`
    (`projects_to_seg.py:36-37`).
    """
    vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
    if vs not in (1, 2, 5):
        raise ValueError("vertex_sampling must be None, 2 or 5 (tables shipped by the reference)")
    if vs not in _part_cache:
        z = np.load(os.path.join(_DATA_DIR, "part_tables.npz"))
        _part_cache[vs] = (z["ids_vs%d" % vs].astype(np.int32), z["off_vs%d" % vs].astype(np.int32))
    return _part_cache[vs]


def load_mean_params():
    """(mean_pose[72] with the global rotation zeroed, mean_shape[10]) as float64.

    `set_cam_params.py:41-47` / `concat_mean_param.py:9-15`: `mean_pose[:3] = 0`.
    """
    z = np.load(os.path.join(_DATA_DIR, "mean_params.npz"))
    pose = z["pose"].astype(np.float64).copy()
    pose[:3] = 0.0
    return pose, z["shape"].astype(np.float64).copy()


def mean86(img_wh: float) -> np.ndarray:
    """[W/2, W/2, W/2, W/1.6 | mean_pose | mean_shape] (`set_cam_params.py:30-47`)."""
    pose, shape = load_mean_params()
    out = np.zeros(86)
    out[0] = img_wh / 2.0
    out[1] = img_wh / 2.0
    out[2] = img_wh / 2.0
    out[3] = img_wh / 1.6
    out[4:76] = pose
    out[76:] = shape
    return out
