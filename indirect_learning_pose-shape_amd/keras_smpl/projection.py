"""orthographic_project: drop-in for `keras_smpl/projection.py:54-81` (smplr_project_fwd/bwd)."""
from .. import ops


def orthographic_project(inputs, vertex_sampling=None):
    """inputs = [verts (B,V,3), smpl (B,86)] -> (B, ceil(V/vs), 3) = (u0 + k_u x, v0 + k_v y, z)."""
    verts, smpl = inputs
    vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
    if verts.dim() != 3 or verts.shape[2] != 3 or smpl.dim() != 2 or smpl.shape[1] < 4 \
            or smpl.shape[0] != verts.shape[0]:
        raise RuntimeError("orthographic_project expects verts (B,V,3) and smpl (B,>=4)")
    return ops.ProjectFn.apply(verts, smpl, vs)


def persepective_project(verts):
    """Declared incomplete and unused in the reference (projection.py:10-51); out of scope."""
    raise NotImplementedError("persepective_project is dead code in the reference (projection.py:11)")
