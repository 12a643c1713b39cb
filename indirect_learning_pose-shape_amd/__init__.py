"""MI355X-native SMPL decoder + body-part soft-rasteriser.

Importing the package does not touch the GPU or the HIP library; the first operator call loads
`libsmplraster_hip.so` and raises if it is missing (there is no CPU fallback).
"""
from . import smpl_model  # noqa: F401
from .smpl_model import SMPLModelData, synthetic_smpl_model, load_part_tables, load_mean_params, mean86  # noqa: F401


def __getattr__(name):
    if name in ("SMPLLayer",):
        from .keras_smpl.batch_smpl import SMPLLayer
        return SMPLLayer
    if name == "SMPLDecoder":
        from .decoder import SMPLDecoder
        return SMPLDecoder
    if name in ("orthographic_project", "compute_mask", "projects_to_seg", "projects_to_silhouette",
                "set_cam_params", "load_mean_set_cam_params", "concat_mean_param"):
        import importlib
        mod = {"orthographic_project": "projection", "set_cam_params": "set_cam_params",
               "load_mean_set_cam_params": "set_cam_params"}.get(name, name)
        return getattr(importlib.import_module(__name__ + ".keras_smpl." + mod), name)
    raise AttributeError(name)
