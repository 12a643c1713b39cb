"""SMPLDecoder: the reference's decoder/segmenter wiring (model.py:108-118) as one module.

    final_param (B,86) -> SMPLLayer -> orthographic_project -> compute_mask -> projects_to_seg
                          [-> projects_to_silhouette  (train_stage2_silhouette.py:82-86)]

Uses the fused autograd node `ops.DecoderFn` (projection as the skinning epilogue, one backward
chain).  The op-by-op surface in `keras_smpl/` computes the same values.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import ops
from .keras_smpl.batch_smpl import _resolve_model


class SMPLDecoder(nn.Module):
    """heads: ("seg",) (default), ("seg", "silhouette") (= with_silhouette=True) or ("silhouette",) - the silhouette-only
    pass of the reference's alternating stage-2 schedule (train_stage2_silhouette.py:262-270), which skips the
    visibility mask, the binning and the 31-part rasteriser altogether.
    outputs: which of "verts", "projects", "mask" to write out (default all three, as the reference's model handles
    expose them, model.py:124-153); the backward needs none of them, so a training step that only consumes the
    scores / the loss saves their 193 KB per mesh of stores with outputs=().
    loss: a `focal_loss.softmax_focal_loss(...)` - the loss head (model.py:119-120 + focal_loss.py:10-46) then runs
    inside the rasteriser: `forward(x, labels)` with an integer class map (B, W, W) returns `seg_loss` (B, W*W), the
    per-pixel loss, and the (B, W, W, 32) scores and their gradient never exist in memory (`seg` is returned only
    with keep_seg=True, detached)."""

    def __init__(self, smpl_path=None, img_wh=48, vertex_sampling=None, num_cam=4, grid_wh=64,
                 ref_compat=True, with_silhouette=False, streams=1, silh_wh=None, deterministic=False,
                 heads=None, outputs=("verts", "projects", "mask"), loss=None, keep_seg=False):
        super().__init__()
        self._model = _resolve_model(smpl_path)
        self.img_wh = int(img_wh)
        self.vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
        if heads is None:
            heads = ("seg", "silhouette") if with_silhouette else ("seg",)
        heads = tuple(heads)
        if not heads or any(h not in ("seg", "silhouette") for h in heads):
            raise ValueError("heads must be a non-empty subset of ('seg', 'silhouette'), got %r" % (heads,))
        self.heads = heads
        with_silhouette = "silhouette" in heads
        if with_silhouette and self.vs != 1:
            raise ValueError("projects_to_silhouette hard-codes 6890 vertices "
                             "(projects_to_silhouette.py:33): no vertex sampling")
        outputs = tuple(outputs)
        if any(o not in ("verts", "projects", "mask") for o in outputs):
            raise ValueError("outputs must be a subset of ('verts', 'projects', 'mask'), got %r" % (outputs,))
        self.outputs = outputs
        self.num_cam, self.grid_wh, self.ref_compat = int(num_cam), int(grid_wh), bool(ref_compat)
        if "seg" in heads and not 0 < self.grid_wh <= 128:   # compute_mask.py:44 hard-codes 64; the z-buffer holds 128 x 128
            raise ValueError("grid_wh must be in 1..128, got %r" % (grid_wh,))
        self.with_silhouette = bool(with_silhouette)
        # the silhouette may have its own resolution (train_stage2_silhouette.py:72-86: `silhs_output_wh`)
        self.silh_wh = int(silh_wh) if silh_wh is not None else self.img_wh
        self.streams = int(streams)      # concurrent mesh chunks (HIP streams); results do not depend on it
        # deterministic=True: the rasterisers' backward accumulates in 64-bit fixed point instead of fp32 LDS atomics,
        # so the same inputs give the same gradient bit for bit on every launch (the reference's ops are pure functions;
        # the default's gradients repeat to rounding).  Forward outputs are bit-reproducible in either mode.
        self.deterministic = bool(deterministic)
        if loss is not None:
            if "seg" not in heads:
                raise ValueError("a fused loss needs the 'seg' head")
            if not hasattr(loss, "gamma") or not hasattr(loss, "weight_classes"):
                raise ValueError("loss must come from focal_loss.softmax_focal_loss(gamma, weight_classes)")
        self.loss = loss
        self.keep_seg = bool(keep_seg)
        self._consts = None
        self._dev = None

    def constants(self, device):
        if self._consts is None or self._dev != device:
            src = getattr(self, "_share", None)
            self._consts = src.constants(device) if src is not None else ops.SMPLConstants.from_model(self._model, device)
            self._dev = device
        return self._consts

    def share_constants(self, other):
        """Use `other`'s device constants (same SMPL model) instead of uploading a second copy (91 MB)."""
        object.__setattr__(self, "_share", other)        # (not a submodule: no parameters, nothing to register)
        self._consts = None
        return self

    def forward(self, x, labels=None):
        """Returns dict(J_transformed [, verts, projects, mask] [, seg | seg_loss] [, silhouette])."""
        if x.dim() != 2 or x.shape[1] != self.num_cam + 82:
            raise RuntimeError("SMPLDecoder expects x of shape (B, %d)" % (self.num_cam + 82))
        c = self.constants(x.device)
        pt = ops.get_part_table(self.vs, x.device, c.V)
        # gradient-free forward of the plain decoder (predict.py:99-118): ONE host call into the at::Tensor layer
        # (torch.ops.smplraster.decoder_fwd, csrc/torch_ops.cpp) instead of the autograd node's ctypes calls - the same
        # two launches, outputs bit for bit; at batch 1 the eager forward is host-bound and this is what it costs
        if (labels is None and self.heads == ("seg",) and self.vs == 1 and self.streams == 1 and c.blend3_fwd is not None
                and not (torch.is_grad_enabled() and x.requires_grad) and x.is_cuda and x.dtype == torch.float32
                and x.shape[0] < ops.POSE_BLEND_SPLIT_B
                # (the one-call op always skins inside the binning launch: an A/B run that switches that off must get
                # the autograd node's launch sequence, which honours it)
                and os.environ.get("SMPLR_FUSE_SKIN", "1") != "0"):
            from . import torch_ops
            if torch_ops.available():
                o = torch_ops.load().decoder_fwd(x.contiguous(), c.as_list(), pt.part_pos, pt.part_off, self.img_wh,
                                                 self.grid_wh, self.ref_compat, self.num_cam)
                out = dict(J_transformed=o[4], seg=o[3])
                for k, t in (("verts", o[0]), ("projects", o[1]), ("mask", o[2])):
                    if k in self.outputs:
                        out[k] = t
                return out
        fused = self.loss is not None and labels is not None
        if labels is not None and self.loss is None:
            raise RuntimeError("labels were given but the decoder was built without loss=softmax_focal_loss(...)")
        spec = None
        if fused:
            from .focal_loss import class_weights
            w = class_weights(x.device)[:32].contiguous() if self.loss.weight_classes else None
            spec = (labels, w, float(self.loss.gamma))
        opts = ops.DecoderOpts(want_verts="verts" in self.outputs, want_proj="projects" in self.outputs,
                               want_mask="mask" in self.outputs, seg="seg" in self.heads,
                               want_seg=(not fused) or self.keep_seg, loss=spec)
        verts, proj, mask, seg, silh, jt, loss = ops.DecoderFn.apply(
            x, c, self.num_cam, self.img_wh, self.vs, pt, self.grid_wh, self.ref_compat,
            (True if self.silh_wh == self.img_wh else self.silh_wh) if self.with_silhouette else False, self.streams,
            self.deterministic, opts)
        out = dict(J_transformed=jt)
        if "verts" in self.outputs:
            out["verts"] = verts
        if "projects" in self.outputs:
            out["projects"] = proj
        if "mask" in self.outputs and "seg" in self.heads:
            out["mask"] = mask
        if "seg" in self.heads:
            if fused:
                out["seg_loss"] = loss
                if self.keep_seg:
                    out["seg"] = seg
            else:
                out["seg"] = seg
        if self.with_silhouette:
            out["silhouette"] = silh
        return out
