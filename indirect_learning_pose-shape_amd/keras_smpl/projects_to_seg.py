"""projects_to_seg: drop-in for `keras_smpl/projects_to_seg.py:9-69` (smplr_seg_fwd/bwd)."""
from .. import ops


def projects_to_seg(input, img_wh, vertex_sampling=None, return_argmin=False, deterministic=False):
    """input = [projects_with_depth (B,V',3), mask_vals (B,V')] -> (B, img_wh, img_wh, 32).

    Channel 0 = background, 1..31 = body parts; rows flipped; raw scores (no softmax), as in the
    reference.  The part tables are the reference's pkl lists (projects_to_seg.py:18-24).
    deterministic=True: bit-reproducible gradient (fixed-point accumulation in the backward).
    """
    proj, mask = input
    if proj.dim() != 3 or proj.shape[2] != 3 or mask.shape != proj.shape[:2]:
        raise RuntimeError("projects_to_seg expects projects (B,V',3) and mask (B,V')")
    if int(img_wh) <= 0:
        raise RuntimeError("img_wh must be positive")
    vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
    pt = ops.get_part_table(vs, proj.device)
    seg, arg, rec = ops.SegRasterFn.apply(proj, mask, int(img_wh), pt, bool(deterministic))
    return (seg, ops.argmin_vertices(arg, rec)) if return_argmin else seg
