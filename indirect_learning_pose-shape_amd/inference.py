"""Predict step of the reference (`predict.py:80-126`) without its file and plotting I/O: one forward of the
encoder + regressor + HIP decoder per batch of images, returning what `predict` hands to its visualiser -
vertices, projected vertices, raw 32-channel scores and their arg-max part map (`predict.py:106-118`)."""
from __future__ import annotations

import torch

from .model import FullModel


@torch.no_grad()
def predict_batch(smpl_model, decoder, images):
    """images (N,3,H,W) or (N,H,W,3) on the HIP device -> dict(smpl (N,86), verts (N,6890,3),
    projects (N,V',3), segs (N,W,W,32) raw scores, seg_maps (N,W,W) int64 = argmax over channels)."""
    was_training = smpl_model.training
    smpl_model.eval()
    try:
        out = FullModel(smpl_model, decoder, "all")(images)
    finally:
        smpl_model.train(was_training)
    return {"smpl": out["smpl"], "verts": out["verts"], "projects": out["projects"], "segs": out["seg"],
            "seg_maps": out["seg"].argmax(dim=-1)}
