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


class StickyVisibility:
    """Opt-in reproduction of the reference's AS-SHIPPED statefulness (SURVEY.md 8(b) Ownership, Appendix A.4):
    `compute_mask_map_over_batch` scatters the visible vertices into ONE `K.variable` of shape (V',) created at
    graph build and never reset to 500 (compute_mask.py:68-70), so across the samples of a batch and across steps
    the set of "visible" vertices only grows.  The reference's outer `map_fn` runs the samples concurrently
    (compute_mask.py:27-30, parallel_iterations = 10), which makes the order of those updates - and hence each
    sample's mask - nondeterministic; this module implements the sequential-order model (sample n sees the union of
    the winners of samples 0..n of this call and of every earlier call): the deterministic member of the family
    of results the reference can produce.  The state is an explicit buffer owned by the module (`reset()` restores
    500 everywhere, i.e. a fresh session); the stateless `compute_mask` above stays the default everywhere."""

    def __init__(self, grid_wh=64, ref_compat=True):
        self.grid_wh, self.ref_compat = int(grid_wh), bool(ref_compat)
        self.state = None                                   # (V',) running minimum of every mask seen so far

    def reset(self):
        self.state = None

    def __call__(self, batch_projects_with_depth):
        fresh = ops.visibility(batch_projects_with_depth, self.grid_wh, self.ref_compat)     # (B, V') in {1, 500}
        if fresh.shape[0] == 0:
            return fresh
        if self.state is None or self.state.shape[0] != fresh.shape[1] or self.state.device != fresh.device:
            self.state = fresh.new_full((fresh.shape[1],), 500.0)
        import torch
        out = torch.minimum(torch.cummin(fresh, dim=0).values, self.state.unsqueeze(0))
        self.state = out[-1].clone()
        return out
