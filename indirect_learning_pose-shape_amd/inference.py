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


class GraphedPredictor:
    """`predict_batch` for a fixed image shape replayed from ONE captured HIP graph: at batch 1 the eager forward
    is bound by ~500 host-side kernel launches (6.7 ms per image), the graph by the kernels themselves.
    `predictor(images)` copies the images into the static input and returns views of the static outputs
    (valid until the next call)."""

    def __init__(self, smpl_model, decoder, example_images, warmup=3):
        self.smpl_model, self.decoder = smpl_model, decoder
        self._in = example_images.detach().clone()
        was_training = smpl_model.training
        smpl_model.eval()
        try:
            side = torch.cuda.Stream(device=self._in.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(max(1, warmup)):                    # MIOpen picks its solvers outside the capture
                    predict_batch(smpl_model, decoder, self._in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph), torch.no_grad():
                self._out = predict_batch(smpl_model, decoder, self._in)
        finally:
            smpl_model.train(was_training)

    def __call__(self, images):
        if images.shape != self._in.shape:
            raise RuntimeError("GraphedPredictor was captured for images of shape %s, got %s"
                               % (tuple(self._in.shape), tuple(images.shape)))
        self._in.copy_(images)
        self._graph.replay()
        return self._out
