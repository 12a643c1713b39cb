"""projects_to_silhouette: drop-in for `keras_smpl/projects_to_silhouette.py:14-44`."""
from .. import ops


def projects_to_silhouette(projects_with_depth, img_wh, return_argmin=False, deterministic=False):
    """(B,V,3) -> (B, img_wh, img_wh, 2) = [1-s, s], s = max_v exp(-d/1.2), rows flipped."""
    proj = projects_with_depth
    if proj.dim() != 3 or proj.shape[2] != 3:
        raise RuntimeError("projects_to_silhouette expects projects (B,V,3)")
    silh, arg = ops.SilhRasterFn.apply(proj, int(img_wh), bool(deterministic))
    return (silh, arg) if return_argmin else silh
