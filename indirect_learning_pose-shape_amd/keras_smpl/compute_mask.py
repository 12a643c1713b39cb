"""compute_mask: drop-in for `keras_smpl/compute_mask.py:12-32` (smplr_visibility).

Stateless semantics: the reference's never-reset, raced mask variable
(compute_mask.py:68-70 under the parallel map_fn of :27-30) is deliberately not reproduced.
"""
from .. import ops


def compute_mask(batch_projects_with_depth, grid_wh=64, ref_compat=True):
    """(B,V',3) -> (B,V') float mask: 1 visible / 500 invisible.  No gradient (compute_mask.py:30).

    grid_wh: the reference hard-codes 64 (compute_mask.py:44).  ref_compat keeps the vertex-1
    artefact of compute_mask.py:99 (an empty grid cell marks vertex 1 visible).
    """
    return ops.visibility(batch_projects_with_depth, grid_wh, ref_compat)
