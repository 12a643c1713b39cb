"""SMPLDecoder: the reference's decoder/segmenter wiring (model.py:108-118) as one module.

    final_param (B,86) -> SMPLLayer -> orthographic_project -> compute_mask -> projects_to_seg
                          [-> projects_to_silhouette  (train_stage2_silhouette.py:82-86)]

Uses the fused autograd node `ops.DecoderFn` (projection as the skinning epilogue, one backward
chain).  The op-by-op surface in `keras_smpl/` computes the same values.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .keras_smpl.batch_smpl import _resolve_model


class SMPLDecoder(nn.Module):
    def __init__(self, smpl_path=None, img_wh=48, vertex_sampling=None, num_cam=4, grid_wh=64,
                 ref_compat=True, with_silhouette=False, streams=1, silh_wh=None, deterministic=False):
        super().__init__()
        self._model = _resolve_model(smpl_path)
        self.img_wh = int(img_wh)
        self.vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
        if with_silhouette and self.vs != 1:
            raise ValueError("projects_to_silhouette hard-codes 6890 vertices "
                             "(projects_to_silhouette.py:33): no vertex sampling")
        self.num_cam, self.grid_wh, self.ref_compat = int(num_cam), int(grid_wh), bool(ref_compat)
        self.with_silhouette = bool(with_silhouette)
        # the silhouette may have its own resolution (train_stage2_silhouette.py:72-86: `silhs_output_wh`)
        self.silh_wh = int(silh_wh) if silh_wh is not None else self.img_wh
        self.streams = int(streams)      # concurrent mesh chunks (HIP streams); results do not depend on it
        # deterministic=True: the rasterisers' backward accumulates in 64-bit fixed point instead of fp32 LDS atomics,
        # so the same inputs give the same gradient bit for bit on every launch (the reference's ops are pure functions;
        # the default's gradients repeat to rounding).  Forward outputs are bit-reproducible in either mode.
        self.deterministic = bool(deterministic)
        self._consts = None
        self._dev = None

    def constants(self, device):
        if self._consts is None or self._dev != device:
            self._consts = ops.SMPLConstants.from_model(self._model, device)
            self._dev = device
        return self._consts

    def forward(self, x):
        """Returns dict(verts, projects, mask, seg[, silhouette], J_transformed)."""
        if x.dim() != 2 or x.shape[1] != self.num_cam + 82:
            raise RuntimeError("SMPLDecoder expects x of shape (B, %d)" % (self.num_cam + 82))
        c = self.constants(x.device)
        pt = ops.get_part_table(self.vs, x.device, c.V)
        verts, proj, mask, seg, silh, jt = ops.DecoderFn.apply(
            x, c, self.num_cam, self.img_wh, self.vs, pt, self.grid_wh, self.ref_compat,
            (True if self.silh_wh == self.img_wh else self.silh_wh) if self.with_silhouette else False, self.streams,
            self.deterministic)
        out = dict(verts=verts, projects=proj, mask=mask, seg=seg, J_transformed=jt)
        if self.with_silhouette:
            out["silhouette"] = silh
        return out
