"""autograd.Functions over the C ABI (include/smplraster.h): the host side of the hot path.

Each Function allocates its outputs / saved tensors with torch (device memory + stream
plumbing only) and calls the HIP kernels through ctypes on torch's current stream.  There is
no eager/CPU fallback anywhere in this file.

Reference callables replaced (reference file:line):
  BatchSMPLFn  - SMPLLayer.call            keras_smpl/batch_smpl.py:96-153
  ProjectFn    - orthographic_project      keras_smpl/projection.py:54-81
  visibility   - compute_mask              keras_smpl/compute_mask.py:12-108
  SegRasterFn  - projects_to_seg           keras_smpl/projects_to_seg.py:9-69
  SilhRasterFn - projects_to_silhouette    keras_smpl/projects_to_silhouette.py:14-44
  DecoderFn    - the model.py:108-118 chain of the above as one autograd node
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import KPAD, CHUNK, check, on_device, ptr, require_cuda, stream
from .smpl_model import SMPLModelData, load_part_tables


# --------------------------------------------------------------------------- constants
@dataclass
class SMPLConstants:
    """Device-resident constants of SMPLLayer.build (keras_smpl/batch_smpl.py:31-93)."""
    V: int
    v_template: torch.Tensor    # (3V,)
    blend: torch.Tensor         # (220, 3V): rows 0..9 shapedirs, 10..216 posedirs, 217..219 zero
    blend_t: torch.Tensor       # (3V, 224): blend transposed, rows zero-padded (backward GEMM)
    J_template: torch.Tensor    # (24,3)   = J_regressor @ v_template
    J_dirs: torch.Tensor        # (24,3,10) = J_regressor @ shapedirs
    lbs_weights: torch.Tensor   # (V,24)
    parents: torch.Tensor       # (24,) int32
    joint_regressor: Optional[torch.Tensor] = None   # (V,19|14) cocoplus / lsp, optional
    lbs_top4: Optional[torch.Tensor] = None   # (V,8) sparse form [w0..w3 | j0..j3] when every row has <= 4 non-zeros
    # blend split into three bf16 terms per entry, in MFMA fragment order (smplr_blend3_pack): the operands of
    # the bf16x3 blend GEMMs, the default on a HIP device; None = the fp32 matrix-core GEMMs (blend / blend_t)
    blend3_fwd: Optional[torch.Tensor] = None
    blend3_bwd: Optional[torch.Tensor] = None

    @staticmethod
    def from_model(model: SMPLModelData, device, joint_type: str = "lsp") -> "SMPLConstants":
        model.validate()
        V = model.num_verts
        sd = np.asarray(model.shapedirs, np.float64).reshape(-1, 10).T      # (10,3V)  :50-55
        pd = np.asarray(model.posedirs, np.float64).reshape(-1, 207).T      # (207,3V) :64-68
        blend = np.zeros((KPAD, 3 * V), np.float64)
        blend[:10] = sd
        blend[10:217] = pd
        blend_t = np.zeros((3 * V, 224), np.float64)
        blend_t[:, :KPAD] = blend.T
        Jreg = np.asarray(model.J_regressor, np.float64)                    # (24,V)
        J_template = Jreg @ np.asarray(model.v_template, np.float64)        # (24,3)
        J_dirs = np.einsum("jv,vck->jck", Jreg, np.asarray(model.shapedirs, np.float64))
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(device)
        w32 = np.asarray(model.weights, np.float32)
        top4 = None
        if int((w32 != 0).sum(axis=1).max()) <= 4:          # real SMPL: <= 4 influences per vertex
            order = np.argsort(w32 == 0, axis=1, kind="stable")[:, :4]   # non-zero joints first, ascending
            wv = np.take_along_axis(w32, order, 1)                      # padding slots carry weight 0
            idx = np.where(wv != 0, order, 0)
            top4 = f32(np.concatenate([wv, idx.astype(np.float32)], axis=1))
        jr = None
        if model.cocoplus_regressor is not None:
            jr = np.asarray(model.cocoplus_regressor, np.float64).T         # (V,19) :82-85
            if joint_type == "lsp":
                jr = jr[:, :14]                                             # :86-87
            jr = f32(jr)
        c = SMPLConstants(
            V=V, v_template=f32(np.asarray(model.v_template).reshape(-1)), blend=f32(blend),
            blend_t=f32(blend_t), J_template=f32(J_template), J_dirs=f32(J_dirs), lbs_weights=f32(model.weights), lbs_top4=top4,
            parents=torch.as_tensor(np.asarray(model.parents, np.int32)).to(device),
            joint_regressor=jr)
        if c.blend.is_cuda and blend_gemm_mode() == "bf16x3":
            c.pack_blend3()
        return c

    def pack_blend3(self):
        """Split + lay out the blend constant for the bf16x3 GEMMs (one-off, on the device)."""
        lib = _lib.load()
        N3 = 3 * self.V
        dev = self.blend.device
        self.blend3_fwd = torch.empty(lib.smplr_blend3_fwd_bytes(N3), dtype=torch.uint8, device=dev)
        self.blend3_bwd = torch.empty(lib.smplr_blend3_bwd_bytes(N3), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.smplr_blend3_pack(ptr(self.blend), N3, ptr(self.blend3_fwd), ptr(self.blend3_bwd), stream()),
                  "smplr_blend3_pack")
        return self

    def as_list(self):
        """The constants in the order `torch.ops.smplraster.smpl_fwd / smpl_bwd / decoder_fwd` take them
        (csrc/torch_ops.cpp): J_template, J_dirs, parents, v_template, blend3_fwd, blend3_bwd, lbs_weights, lbs_top4,
        blend_t; absent optional ones as empty tensors."""
        e = lambda t, dt=torch.float32: t if t is not None else torch.empty(0, dtype=dt, device=self.v_template.device)
        return [self.J_template, self.J_dirs, self.parents, self.v_template, e(self.blend3_fwd, torch.uint8),
                e(self.blend3_bwd, torch.uint8), self.lbs_weights, e(self.lbs_top4), self.blend_t]

    def fp32_gemm(self):
        """A view of the same constants that runs the blend GEMMs on the fp32 matrix cores."""
        import dataclasses
        return dataclasses.replace(self, blend3_fwd=None, blend3_bwd=None)


def blend_gemm_mode() -> str:
    """'bf16x3' (default): blend GEMMs on the bf16 matrix cores with 3-way split, fp32-grade operands;
    'f32': v_mfma_f32_32x32x2_f32.  Chosen when the constants are uploaded (env SMPLR_BLEND_GEMM)."""
    import os
    m = os.environ.get("SMPLR_BLEND_GEMM", "bf16x3").lower()
    if m not in ("bf16x3", "f32"):
        raise RuntimeError("SMPLR_BLEND_GEMM must be bf16x3 or f32, got %r" % m)
    return m


@dataclass
class PartTable:
    """Part-major vertex positions for the rasteriser (projects_to_seg.py:18-24,36-37)."""
    P: int
    K: int
    part_pos: torch.Tensor   # (K,) int32 positions into the (strided) vertex list
    part_off: torch.Tensor   # (P+1,) int32 CSR offsets
    VP: int


_part_tables = {}


def build_part_table(ids, off, vertex_sampling, num_verts, device) -> PartTable:
    vs = 1 if vertex_sampling in (None, 1) else int(vertex_sampling)
    P = len(off) - 1
    part_pos = (np.asarray(ids, np.int64) // vs).astype(np.int32)          # :36-37
    VP = (num_verts + vs - 1) // vs
    assert part_pos.min() >= 0 and part_pos.max() < VP and int(off[-1]) == len(part_pos)
    if len(np.unique(part_pos)) != len(part_pos):
        # the backward keeps ONE record slot per vertex (plain stores / gather by vertex, no atomics); the
        # reference's three tables (part_vertices.pkl, 2_/5_sampled_part_vertices.pkl) are partitions
        raise ValueError("part table lists a vertex position more than once (after // vertex_sampling): "
                         "each vertex may belong to at most one part")
    return PartTable(P=P, K=int(len(part_pos)),
                     part_pos=torch.as_tensor(part_pos).to(device),
                     part_off=torch.as_tensor(np.asarray(off, np.int32)).to(device), VP=VP)


def get_part_table(vertex_sampling, device, num_verts=6890) -> PartTable:
    key = (1 if vertex_sampling in (None, 1) else int(vertex_sampling), str(device), num_verts)
    if key not in _part_tables:
        ids, off = load_part_tables(vertex_sampling)
        _part_tables[key] = build_part_table(ids, off, vertex_sampling, num_verts, device)
    return _part_tables[key]


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _workspace(nbytes, like):
    return torch.empty(max(int(nbytes), 4) // 4 + 1, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------- raw stage calls
class PoseCoef:
    """The blend GEMM's per-step operand as smplr_pose_fwd writes it: `kmajor` (220, ld) fp32 for the fp32
    matrix-core GEMM and/or `frag3`, the same columns as bf16x3 MFMA fragments for smplr_blend3_fwd."""
    __slots__ = ("kmajor", "frag3", "B")

    def __init__(self, kmajor, frag3, B):
        self.kmajor, self.frag3, self.B = kmajor, frag3, B

    @property
    def device(self):
        return (self.kmajor if self.kmajor is not None else self.frag3).device

    @property
    def is_cuda(self):
        return self.device.type == "cuda"


@on_device
def _pose_fwd(x, num_cam, c: SMPLConstants, out=None, want=None):
    """want: 'frag3' | 'kmajor' | 'both'; default = what the constants' blend GEMM reads."""
    lib = _lib.load()
    B = x.shape[0]
    if out is None:
        out = (None, _empty((B, 24, 9), x), _empty((B, 24, 3), x), _empty((B, 24, 12), x), _empty((B, 24, 3), x))
    _, Rs, J, A, Jt = out
    if want is None:
        want = "frag3" if c.blend3_fwd is not None else "kmajor"
    km = _empty((KPAD, lib.smplr_coef_ld(B)), x) if want in ("kmajor", "both") else None
    f3 = (torch.empty(max(int(lib.smplr_coef3_bytes(B)), 16), dtype=torch.uint8, device=x.device)
          if want in ("frag3", "both") else None)
    check(lib.smplr_pose_fwd(ptr(x), x.shape[1], num_cam, B, ptr(c.J_template), ptr(c.J_dirs),
                             ptr(c.parents), ptr(km), ptr(f3), ptr(Rs), ptr(J), ptr(A), ptr(Jt), stream()),
          "smplr_pose_fwd")
    return PoseCoef(km, f3, B), Rs, J, A, Jt


@on_device
def _pose_blend_fwd(x, num_cam, c: SMPLConstants, out=None, v_posed=None):
    """smplr_pose_blend3_fwd: pose kernel + blend GEMM in one launch (bf16x3 constants only) -> Rs, J, A, Jt, v_posed;
    bit-identical to _pose_fwd followed by _blend_fwd."""
    lib = _lib.load()
    if c.blend3_fwd is None:
        raise RuntimeError("the fused pose + blend launch needs the bf16x3 constants (SMPLConstants.pack_blend3)")
    B = x.shape[0]
    if out is None:
        out = (_empty((B, 24, 9), x), _empty((B, 24, 3), x), _empty((B, 24, 12), x), _empty((B, 24, 3), x))
    Rs, J, A, Jt = out
    if v_posed is None:
        v_posed = _empty((B, c.V, 3), x)
    check(lib.smplr_pose_blend3_fwd(ptr(x), x.shape[1], num_cam, B, ptr(c.J_template), ptr(c.J_dirs), ptr(c.parents),
                                    ptr(c.blend3_fwd), ptr(c.v_template), 3 * c.V, ptr(Rs), ptr(J), ptr(A), ptr(Jt),
                                    ptr(v_posed), stream()), "smplr_pose_blend3_fwd")
    return Rs, J, A, Jt, v_posed


@on_device
def _blend_fwd(coef: PoseCoef, c: SMPLConstants, B, out=None):
    """coef: what _pose_fwd returned for the same B meshes."""
    lib = _lib.load()
    if coef.B != B:
        raise RuntimeError("coef was computed for %d meshes, not %d" % (coef.B, B))
    v_posed = _empty((B, c.V, 3), c.v_template) if out is None else out
    if c.blend3_fwd is not None:
        if coef.frag3 is None:                    # k-major only: split it here (smplr_coef3_pack)
            coef.frag3 = torch.empty(max(int(lib.smplr_coef3_bytes(B)), 16), dtype=torch.uint8,
                                     device=v_posed.device)
            check(lib.smplr_coef3_pack(ptr(coef.kmajor), B, ptr(coef.frag3), stream()), "smplr_coef3_pack")
        check(lib.smplr_blend3_fwd(ptr(coef.frag3), ptr(c.blend3_fwd), ptr(c.v_template), B, 3 * c.V, ptr(v_posed),
                                   stream()), "smplr_blend3_fwd")
    else:
        if coef.kmajor is None:
            raise RuntimeError("the fp32 blend GEMM needs the k-major coef (_pose_fwd(..., want='kmajor'))")
        check(lib.smplr_blend_fwd(ptr(coef.kmajor), ptr(c.blend), ptr(c.v_template), B, 3 * c.V, ptr(v_posed),
                                  stream()), "smplr_blend_fwd")
    return v_posed


@on_device
def _skin_fwd(v_posed, A, c: SMPLConstants, cam=None, vertex_sampling=1, want_verts=True, out=None):
    lib = _lib.load()
    B = v_posed.shape[0]
    vs = int(vertex_sampling)
    if out is not None:
        verts, proj = out
    else:
        verts = _empty((B, c.V, 3), v_posed) if want_verts else None
        proj = _empty((B, (c.V + vs - 1) // vs, 3), v_posed) if cam is not None else None
    check(lib.smplr_skin_fwd(ptr(v_posed), ptr(c.lbs_weights), ptr(c.lbs_top4), ptr(A), ptr(cam),
                             cam.shape[1] if cam is not None else 0, B, c.V, vs, ptr(verts), ptr(proj),
                             stream()), "smplr_skin_fwd")
    return verts, proj


@on_device
def _smpl_bwd(x, num_cam, c: SMPLConstants, Rs, J, A, v_posed, dverts, dproj, dJt, vertex_sampling=1,
              out=None, seg_grad=None):
    """dverts and/or dproj (+ optional dJ_transformed) -> dx (B, x_stride).
    seg_grad = (part, vslot, nsplit): the segmentation gradient as _seg_bwd(..., merge=False) leaves it,
    gathered by vertex inside the skinning backward and added to dproj."""
    lib = _lib.load()
    B = x.shape[0]
    vs = int(vertex_sampling)
    dx = _empty(tuple(x.shape), x) if out is None else out
    if x.shape[1] > num_cam + 82:
        dx.zero_()
    ws = _workspace(lib.smplr_smpl_bwd_workspace(B, c.V), x)
    sp, sv, sn = seg_grad if seg_grad is not None else (None, None, 0)
    check(lib.smplr_smpl_bwd(ptr(dverts), ptr(dproj), ptr(sp), ptr(sv), int(sn), ptr(dJt), ptr(x), x.shape[1],
                             num_cam, B, c.V, vs,
                             ptr(c.blend_t), ptr(c.blend3_bwd), ptr(c.lbs_weights), ptr(c.lbs_top4), ptr(c.J_dirs), ptr(c.parents),
                             ptr(Rs), ptr(J),
                             ptr(A), ptr(v_posed), ptr(dx), ptr(ws), stream()), "smplr_smpl_bwd")
    return dx


@on_device
def visibility(proj, grid_wh=64, ref_compat=True, out=None):
    """compute_mask (keras_smpl/compute_mask.py:12-108), stateless; no gradient (:30)."""
    lib = _lib.load()
    proj = require_cuda(proj.detach(), "projects_with_depth")
    if proj.dim() != 3 or proj.shape[2] != 3:
        raise RuntimeError("projects_with_depth must be (B, V', 3)")
    B, VP = proj.shape[0], proj.shape[1]
    mask = _empty((B, VP), proj) if out is None else out
    check(lib.smplr_visibility(ptr(proj), B, VP, int(grid_wh), 1 if ref_compat else 0, ptr(mask),
                               stream()), "smplr_visibility")
    return mask


@on_device
def _seg_fwd(proj, mask, W, pt: PartTable, out=None, vslot=None):
    """vslot (B,VP) int16, optional output: each vertex' record slot, for the gather form of the backward."""
    lib = _lib.load()
    B, VP = proj.shape[0], proj.shape[1]
    if VP != pt.VP:
        raise RuntimeError("projects has %d vertices but the part table expects %d" % (VP, pt.VP))
    ws = _workspace(lib.smplr_seg_workspace(B, VP, W, pt.P, pt.K), proj)
    if out is not None:
        seg, arg, rec = out
    else:
        seg = _empty((B, W, W, pt.P + 1), proj)
        arg = _empty((B, W, W, 32), proj, torch.int16)
        rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), proj)
    check(lib.smplr_seg_fwd(ptr(proj), ptr(mask), B, VP, W, ptr(pt.part_pos), ptr(pt.part_off), pt.P,
                            pt.K, ptr(ws), ptr(seg), ptr(arg), ptr(rec), ptr(vslot), stream()), "smplr_seg_fwd")
    return seg, arg, rec


@on_device
def _vis_seg_fwd(proj, W, pt: PartTable, grid_wh=64, ref_compat=True, out=None, vslot=None):
    """compute_mask + projects_to_seg in one call (smplr_vis_seg_fwd): -> mask, seg, arg, rec."""
    lib = _lib.load()
    B, VP = proj.shape[0], proj.shape[1]
    if VP != pt.VP:
        raise RuntimeError("projects has %d vertices but the part table expects %d" % (VP, pt.VP))
    ws = _workspace(lib.smplr_seg_workspace(B, VP, W, pt.P, pt.K), proj)
    if out is not None:
        mask, seg, arg, rec = out
    else:
        mask = _empty((B, VP), proj)
        seg = _empty((B, W, W, pt.P + 1), proj)
        arg = _empty((B, W, W, 32), proj, torch.int16)
        rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), proj)
    check(lib.smplr_vis_seg_fwd(ptr(proj), B, VP, W, int(grid_wh), 1 if ref_compat else 0, ptr(pt.part_pos),
                                ptr(pt.part_off), pt.P, pt.K, ptr(ws), ptr(mask), ptr(seg), ptr(arg), ptr(rec),
                                ptr(vslot), stream()), "smplr_vis_seg_fwd")
    return mask, seg, arg, rec


@on_device
def _skin_vis_seg_fwd(v_posed, A, c: SMPLConstants, cam, W, pt: PartTable, grid_wh=64, ref_compat=True, out=None,
                      vslot=None):
    """_skin_fwd + _vis_seg_fwd as one call of two launches (smplr_skin_vis_seg_fwd): the binning workgroups skin
    their own vertices.  Needs the sparse skinning weights and no vertex sampling.
    -> verts, proj, mask, seg, arg, rec (bit for bit what the two calls give)."""
    lib = _lib.load()
    B, V = v_posed.shape[0], c.V
    if c.lbs_top4 is None or pt.VP != V:
        raise RuntimeError("_skin_vis_seg_fwd needs <= 4 skinning weights per vertex and vertex_sampling = 1")
    if not lib.smplr_skin_vis_seg_fits(V, int(W), int(grid_wh)):
        raise RuntimeError("_skin_vis_seg_fwd: V=%d, W=%d, grid_wh=%d do not fit the binning workgroup's LDS "
                           "(smplr_skin_vis_seg_fits): call _skin_fwd and _vis_seg_fwd" % (V, W, grid_wh))
    ws = _workspace(lib.smplr_seg_workspace(B, V, W, pt.P, pt.K), v_posed)
    if out is not None:
        verts, proj, mask, seg, arg, rec = out
    else:
        verts, proj = _empty((B, V, 3), v_posed), _empty((B, V, 3), v_posed)
        mask = _empty((B, V), v_posed)
        seg = _empty((B, W, W, pt.P + 1), v_posed)
        arg = _empty((B, W, W, 32), v_posed, torch.int16)
        rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), v_posed)
    check(lib.smplr_skin_vis_seg_fwd(ptr(v_posed), ptr(c.lbs_top4), ptr(A), ptr(cam), cam.shape[1], B, V, W,
                                     int(grid_wh), 1 if ref_compat else 0, ptr(pt.part_pos), ptr(pt.part_off), pt.P,
                                     pt.K, ptr(ws), ptr(verts), ptr(proj), ptr(mask), ptr(seg), ptr(arg), ptr(rec),
                                     ptr(vslot), stream()), "smplr_skin_vis_seg_fwd")
    return verts, proj, mask, seg, arg, rec


@on_device
def _skin_vis_seg_fwd_ex(v_posed, A, c: SMPLConstants, cam, W, pt: PartTable, labels=None, class_w=None, gamma=0.0,
                         grid_wh=64, ref_compat=True, verts=None, proj=None, mask=None, seg=None, arg=None, rec=None,
                         vslot=None, loss=None, stats=None, vmax=None):
    """_skin_vis_seg_fwd with optional extras (smplr_skin_vis_seg_fwd_ex): with `labels` (B,W,W) int32 the loss head
    runs as the rasteriser's epilogue -> loss (B, W*W), stats (B, W*W, 4); `vmax` (B,W,W) receives each pixel's largest
    part score (the silhouette rasteriser's hint); verts / proj / mask / seg are written only where a tensor is given
    (seg may be omitted only with labels)."""
    lib = _lib.load()
    B, V = v_posed.shape[0], c.V
    if c.lbs_top4 is None or pt.VP != V or not lib.smplr_skin_vis_seg_fits(V, int(W), int(grid_wh)):
        raise RuntimeError("_skin_vis_seg_fwd_ex: needs sparse skinning weights, vertex_sampling = 1 and sizes that fit "
                           "the binning workgroup's LDS (smplr_skin_vis_seg_fits)")
    ws = _workspace(lib.smplr_seg_workspace(B, V, W, pt.P, pt.K), v_posed)
    if arg is None:
        arg = _empty((B, W, W, 32), v_posed, torch.int16)
    if rec is None:
        rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), v_posed)
    if labels is not None:
        loss = _empty((B, W * W), v_posed) if loss is None else loss
        stats = _empty((B, W * W, 4), v_posed) if stats is None else stats
    else:
        loss = stats = None
    check(lib.smplr_skin_vis_seg_fwd_ex(ptr(v_posed), ptr(c.lbs_top4), ptr(A), ptr(cam), cam.shape[1], B, V, W,
                                        int(grid_wh), 1 if ref_compat else 0, ptr(pt.part_pos), ptr(pt.part_off),
                                        pt.P, pt.K, ptr(ws), ptr(labels), ptr(class_w), float(gamma), ptr(verts),
                                        ptr(proj), ptr(mask), ptr(seg), ptr(arg), ptr(rec), ptr(vslot), ptr(loss),
                                        ptr(stats), ptr(vmax), stream()), "smplr_skin_vis_seg_fwd_ex")
    return loss, stats, arg, rec


@on_device
def _seg_raster_ex(ws, rec, B, W, pt: PartTable, labels=None, class_w=None, gamma=0.0, seg=None, arg=None, loss=None,
                   stats=None, vmax=None):
    """Stage 2 with optional extras (smplr_seg_raster_ex) over a binned workspace -> loss, stats, arg (loss epilogue with
    `labels`; `vmax`: per-pixel largest part score)."""
    lib = _lib.load()
    if arg is None:
        arg = _empty((B, W, W, 32), rec, torch.int16)
    if labels is not None:
        loss = _empty((B, W * W), rec) if loss is None else loss
        stats = _empty((B, W * W, 4), rec) if stats is None else stats
    else:
        loss = stats = None
    check(lib.smplr_seg_raster_ex(B, W, pt.P, pt.K, ptr(ws), ptr(rec), ptr(labels), ptr(class_w), float(gamma),
                                  ptr(seg), ptr(arg), ptr(loss), ptr(stats), ptr(vmax), stream()), "smplr_seg_raster_ex")
    return loss, stats, arg


def _seg_raster_loss(ws, rec, B, W, pt, labels, class_w, gamma, seg=None, arg=None, loss=None, stats=None):
    """The loss-epilogue form of _seg_raster_ex (kept as a name of its own: tests, tools)."""
    return _seg_raster_ex(ws, rec, B, W, pt, labels, class_w, gamma, seg=seg, arg=arg, loss=loss, stats=stats)


@on_device
def _seg_loss_bwd(dloss, stats, arg, rec, VP, W, pt: PartTable, merge=True, deterministic=False):
    """_seg_bwd fed with dloss (B, W*W) + the forward's stats instead of dseg (smplr_seg_loss_bwd)."""
    lib = _lib.load()
    B = arg.shape[0]
    ws = _workspace(lib.smplr_seg_bwd_workspace(B, W), dloss)
    dproj = _empty((B, VP, 3), dloss) if merge else None
    check(lib.smplr_seg_loss_bwd(ptr(dloss), ptr(stats), ptr(arg), ptr(rec), B, VP, W, pt.P, pt.K, ptr(dproj), ptr(ws),
                                 1 if deterministic else 0, stream()), "smplr_seg_loss_bwd")
    return dproj if merge else (ws, int(lib.smplr_seg_bwd_nsplit(B, W)))


@on_device
def _seg_bin(proj, mask, W, pt: PartTable, grid_wh=0, ref_compat=True, rec=None, vslot=None, ws=None):
    """Stage 1 of the segmentation forward alone (smplr_seg_bin): grid_wh > 0 computes the mask inside (output),
    grid_wh = 0 reads it.  -> (ws, rec): what _seg_raster needs."""
    lib = _lib.load()
    B, VP = proj.shape[0], proj.shape[1]
    if VP != pt.VP:
        raise RuntimeError("projects has %d vertices but the part table expects %d" % (VP, pt.VP))
    if ws is None:
        ws = _workspace(lib.smplr_seg_workspace(B, VP, W, pt.P, pt.K), proj)
    if rec is None:
        rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), proj)
    check(lib.smplr_seg_bin(ptr(proj), ptr(mask), B, VP, W, int(grid_wh), 1 if ref_compat else 0, ptr(pt.part_pos),
                            ptr(pt.part_off), pt.P, pt.K, ptr(ws), ptr(rec), ptr(vslot), stream()), "smplr_seg_bin")
    return ws, rec


@on_device
def _seg_raster(ws, rec, B, W, pt: PartTable, out=None):
    """Stage 2 alone (smplr_seg_raster) over a binned workspace -> seg, arg."""
    lib = _lib.load()
    if out is not None:
        seg, arg = out
    else:
        seg = _empty((B, W, W, pt.P + 1), rec)
        arg = _empty((B, W, W, 32), rec, torch.int16)
    check(lib.smplr_seg_raster(B, W, pt.P, pt.K, ptr(ws), ptr(rec), ptr(seg), ptr(arg), stream()), "smplr_seg_raster")
    return seg, arg


@on_device
def _seg_bwd(dseg, arg, rec, VP, W, pt: PartTable, merge=True, deterministic=False):
    """merge=True -> dproj (B,VP,3).  merge=False -> (part, nsplit): the per-row-block slot sums, to be handed
    to _smpl_bwd(seg_grad=(part, vslot, nsplit)) which gathers them by vertex (no merge launch, no dproj)."""
    lib = _lib.load()
    B = arg.shape[0]
    ws = _workspace(lib.smplr_seg_bwd_workspace(B, W), dseg)
    dproj = _empty((B, VP, 3), dseg) if merge else None
    check(lib.smplr_seg_bwd(ptr(dseg), ptr(arg), ptr(rec), B, VP, W, pt.P, pt.K, ptr(dproj), ptr(ws),
                            1 if deterministic else 0, stream()), "smplr_seg_bwd")
    return dproj if merge else (ws, int(lib.smplr_seg_bwd_nsplit(B, W)))


def raster_plan(rec, W, pt: PartTable):
    """How the rasteriser ran (or will run) the batch whose record lists are `rec` - host arithmetic on the list headers
    and smplr_seg_raster_plan; synchronises (a diagnostic for bench.py and the tests, never on the product path).
    -> dict: far_records (B,) = each mesh's far-reaching list, padded per part; tile_records (tiles,) = what one pass
    over a tile's LDS table takes (0: the tile walks the list with scalar loads); passes (B, tiles); blocks,
    blocks_multi_pass, blocks_scalar_walk, passes_max; pair_lanes, part_ranges."""
    import ctypes
    lib = _lib.load()
    B = rec.shape[0]
    info = (ctypes.c_int32 * 8)()
    nt = lib.smplr_seg_raster_plan(B, W, pt.P, pt.K, info, None)
    if nt <= 0:
        raise RuntimeError("smplr_seg_raster_plan: bad sizes B=%d W=%d P=%d K=%d" % (B, W, pt.P, pt.K))
    tiles = (ctypes.c_int32 * nt)()
    lib.smplr_seg_raster_plan(B, W, pt.P, pt.K, info, tiles)
    head = rec[:, -1, :].contiguous().view(torch.int32).cpu().numpy()
    far, nonunit = head[:, 2].astype(np.int64), head[:, 1] != 0
    trec = np.asarray(list(tiles), np.int64)
    walk = (trec[None, :] == 0) | nonunit[:, None]                                 # (B, tiles)
    passes = np.where(walk, 0, np.maximum(1, -(-far[:, None] // np.maximum(trec[None, :], 1))))
    return {"far_records": far, "tile_records": trec, "passes": passes, "blocks": int(B * nt),
            "blocks_multi_pass": int((passes > 1).sum()), "blocks_scalar_walk": int(walk.sum()),
            "passes_max": int(passes.max()) if passes.size else 0, "pair_lanes": int(info[0]), "part_ranges": int(info[1])}


def argmin_vertices(arg, rec):
    """(B,W,W,31) int64 vertex positions of the maximising vertices (-1 = none) from arg/rec."""
    slots = arg[..., 1:].to(torch.int64)
    B = arg.shape[0]
    pos = rec[..., 3].contiguous().view(torch.int32).to(torch.int64)       # (B,S)
    flat = slots.reshape(B, -1)
    got = torch.gather(pos, 1, flat.clamp(min=0))
    return torch.where(flat >= 0, got, torch.full_like(got, -1)).reshape(slots.shape)


@on_device
def _silh_fwd(proj, W, out=None, hint=None):
    """hint (B,W,W), optional: the 31-part rasteriser's per-pixel largest score of the same meshes at the same W
    (_seg_raster_ex(vmax=)): bounds the exact search, same outputs bit for bit (smplr_silh_fwd_hint)."""
    lib = _lib.load()
    B, VP = proj.shape[0], proj.shape[1]
    if out is not None:
        silh, arg = out
    else:
        silh = _empty((B, W, W, 2), proj)
        arg = _empty((B, W, W), proj, torch.int32)
    ws = _workspace(lib.smplr_silh_workspace(B, VP, W), proj)
    check(lib.smplr_silh_fwd_hint(ptr(proj), ptr(hint), B, VP, W, ptr(silh), ptr(arg), ptr(ws), stream()),
          "smplr_silh_fwd_hint")
    return silh, arg


@on_device
def _silh_bwd(dsilh, silh, arg, proj, W, deterministic=False):
    lib = _lib.load()
    B, VP = proj.shape[0], proj.shape[1]
    dproj = _empty((B, VP, 3), proj)
    check(lib.smplr_silh_bwd(ptr(dsilh), ptr(silh), ptr(arg), ptr(proj), B, VP, W, ptr(dproj),
                             1 if deterministic else 0, stream()), "smplr_silh_bwd")
    return dproj


# --------------------------------------------------------------------------- autograd nodes
class BatchSMPLFn(torch.autograd.Function):
    """x (B, num_cam+82) -> verts (B,V,3), J_transformed (B,24,3)."""

    @staticmethod
    @on_device
    def forward(ctx, x, consts: SMPLConstants, num_cam: int):
        x = require_cuda(x, "x")
        ctx.set_materialize_grads(False)
        if consts.blend3_fwd is not None:
            Rs, J, A, Jt, v_posed = _pose_blend_fwd(x, num_cam, consts)        # one launch
        else:
            coef, Rs, J, A, Jt = _pose_fwd(x, num_cam, consts)
            v_posed = _blend_fwd(coef, consts, x.shape[0])
        verts, _ = _skin_fwd(v_posed, A, consts)
        ctx.consts, ctx.num_cam = consts, num_cam
        ctx.save_for_backward(x, Rs, J, A, v_posed)
        return verts, Jt

    @staticmethod
    @on_device
    def backward(ctx, dverts, dJt):
        x, Rs, J, A, v_posed = ctx.saved_tensors
        dverts = require_cuda(dverts, "dverts") if dverts is not None else None
        dJt = require_cuda(dJt, "dJ_transformed") if dJt is not None else None
        if dverts is None:
            dverts = torch.zeros_like(v_posed)
        dx = _smpl_bwd(x, ctx.num_cam, ctx.consts, Rs, J, A, v_posed, dverts, None, dJt)
        return dx, None, None


class ProjectFn(torch.autograd.Function):
    """(verts (B,V,3), smpl (B,>=4)) -> (B,V',3)   (keras_smpl/projection.py:54-81)."""

    @staticmethod
    @on_device
    def forward(ctx, verts, smpl, vertex_sampling: int):
        lib = _lib.load()
        verts = require_cuda(verts, "verts")
        smpl = require_cuda(smpl, "smpl")
        B, V = verts.shape[0], verts.shape[1]
        vs = int(vertex_sampling)
        proj = _empty((B, (V + vs - 1) // vs, 3), verts)
        check(lib.smplr_project_fwd(ptr(verts), ptr(smpl), smpl.shape[1], B, V, vs, ptr(proj), stream()),
              "smplr_project_fwd")
        ctx.vs = vs
        ctx.save_for_backward(verts, smpl)
        return proj

    @staticmethod
    @on_device
    def backward(ctx, dproj):
        lib = _lib.load()
        verts, smpl = ctx.saved_tensors
        dproj = require_cuda(dproj, "dproj")
        B, V = verts.shape[0], verts.shape[1]
        dverts = _empty(tuple(verts.shape), verts)
        dcam = _empty((B, 4), verts)
        check(lib.smplr_project_bwd(ptr(dproj), ptr(verts), ptr(smpl), smpl.shape[1], B, V, ctx.vs,
                                    ptr(dverts), ptr(dcam), stream()), "smplr_project_bwd")
        dsmpl = torch.zeros_like(smpl)
        dsmpl[:, :4] = dcam
        return dverts, dsmpl, None


class SegRasterFn(torch.autograd.Function):
    """(proj (B,V',3), mask (B,V')) -> seg (B,W,W,32)   (keras_smpl/projects_to_seg.py:9-69)."""

    @staticmethod
    @on_device
    def forward(ctx, proj, mask, img_wh: int, pt: PartTable, deterministic=False):
        proj = require_cuda(proj, "projects_with_depth")
        mask = require_cuda(mask, "mask_vals")
        ctx.set_materialize_grads(False)
        seg, arg, rec = _seg_fwd(proj, mask, int(img_wh), pt)
        ctx.W, ctx.pt, ctx.VP, ctx.det = int(img_wh), pt, proj.shape[1], bool(deterministic)
        ctx.save_for_backward(arg, rec)
        ctx.mark_non_differentiable(arg, rec)
        return seg, arg, rec

    @staticmethod
    @on_device
    def backward(ctx, dseg, _darg, _drec):
        arg, rec = ctx.saved_tensors
        if dseg is None:
            return None, None, None, None, None
        dseg = require_cuda(dseg, "dseg")
        return _seg_bwd(dseg, arg, rec, ctx.VP, ctx.W, ctx.pt, deterministic=ctx.det), None, None, None, None


class SilhRasterFn(torch.autograd.Function):
    """proj (B,V',3) -> silh (B,W,W,2)   (keras_smpl/projects_to_silhouette.py:14-44)."""

    @staticmethod
    @on_device
    def forward(ctx, proj, img_wh: int, deterministic=False):
        proj = require_cuda(proj, "projects_with_depth")
        silh, arg = _silh_fwd(proj, int(img_wh))
        ctx.W, ctx.det = int(img_wh), bool(deterministic)
        ctx.save_for_backward(proj, silh, arg)
        ctx.mark_non_differentiable(arg)
        return silh, arg

    @staticmethod
    @on_device
    def backward(ctx, dsilh, _darg):
        proj, silh, arg = ctx.saved_tensors
        dsilh = require_cuda(dsilh, "dsilh")
        return _silh_bwd(dsilh, silh, arg, proj, ctx.W, ctx.det), None, None


def _focal_targets(target, npix, C):
    """labels (int class ids, npix) -> (labels, None); dense y_true (npix, C) -> (None, y_true)."""
    if target.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8):
        if target.numel() != npix:
            raise RuntimeError("labels hold %d entries for %d pixels" % (target.numel(), npix))
        return require_cuda(target.to(torch.int32), "labels", torch.int32), None
    if target.numel() != npix * C:
        raise RuntimeError("y_true holds %d entries, expected %d x %d" % (target.numel(), npix, C))
    return None, require_cuda(target, "y_true")


class SoftmaxFocalFn(torch.autograd.Function):
    """Raw scores (..., C) + targets -> per-pixel focal loss (N, W*W)  (model.py:119-120 +
    focal_loss.py:10-46; gamma = 0 without weights = the silhouette head's cross-entropy).
    Targets are data: no gradient.  The softmax is recomputed in backward, not stored."""

    @staticmethod
    @on_device
    def forward(ctx, scores, target, class_w, gamma: float):
        scores = require_cuda(scores, "scores")
        C = scores.shape[-1]
        N = scores.shape[0]
        npix = scores.numel() // C if C else 0
        labels, y_true = _focal_targets(target, npix, C)
        if class_w is not None:
            class_w = require_cuda(class_w, "class_w")
            if class_w.numel() != C:
                raise RuntimeError("class_w needs %d entries" % C)
        loss = _empty((N, npix // N if N else 0), scores)
        check(_lib.load().smplr_focal_fwd(ptr(scores), ptr(labels), ptr(y_true), ptr(class_w), float(gamma),
                                          npix, C, ptr(loss), None, stream()), "smplr_focal_fwd")
        ctx.gamma, ctx.npix, ctx.C = float(gamma), npix, C
        ctx.save_for_backward(scores, labels if labels is not None else y_true, class_w)
        ctx.is_labels = labels is not None
        return loss

    @staticmethod
    @on_device
    def backward(ctx, dloss):
        scores, tgt, class_w = ctx.saved_tensors
        dloss = require_cuda(dloss, "dloss")
        labels, y_true = (tgt, None) if ctx.is_labels else (None, tgt)
        dscores = torch.empty_like(scores)
        check(_lib.load().smplr_focal_bwd(ptr(scores), ptr(labels), ptr(y_true), ptr(class_w), ctx.gamma,
                                          ptr(dloss), ctx.npix, ctx.C, ptr(dscores), stream()), "smplr_focal_bwd")
        return dscores, None, None, None


@on_device
def softmax_probs(scores):
    """Softmax over the last axis through the loss kernel's forward (the 'segs' model output,
    model.py:119-120); no autograd (use torch.softmax where a gradient through probs is needed)."""
    scores = require_cuda(scores, "scores")
    C = scores.shape[-1]
    npix = scores.numel() // C
    labels = torch.zeros(npix, dtype=torch.int32, device=scores.device)
    loss = _empty((npix,), scores)
    probs = torch.empty_like(scores)
    check(_lib.load().smplr_focal_fwd(ptr(scores), ptr(labels), None, None, 0.0, npix, C, ptr(loss), ptr(probs),
                                      stream()), "smplr_focal_fwd")
    return probs.reshape(scores.shape[0], -1, C)


class PReLUFn(torch.autograd.Function):
    """Per-channel PReLU on NCHW fp32 (the ENet encoder's activation, encoders/encoder_enet_simple.py:21):
    smplr_prelu_fwd / smplr_prelu_bwd.  x (N, C, ...) and weight (C,)."""

    @staticmethod
    @on_device
    def forward(ctx, x, weight):
        x = require_cuda(x, "x")
        weight = require_cuda(weight, "weight")
        N, C = x.shape[0], x.shape[1]
        if weight.numel() != C:
            raise RuntimeError("PReLU weight has %d entries for %d channels" % (weight.numel(), C))
        HW = x.numel() // (N * C) if N * C else 1
        y = torch.empty_like(x)
        check(_lib.load().smplr_prelu_fwd(ptr(x), ptr(weight), N, C, HW, ptr(y), stream()), "smplr_prelu_fwd")
        ctx.save_for_backward(x, weight)
        ctx.dims = (N, C, HW)
        return y

    @staticmethod
    @on_device
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, C, HW = ctx.dims
        gy = require_cuda(gy, "gy")
        lib = _lib.load()
        gx, gw = torch.empty_like(x), torch.empty_like(weight)
        ws = _workspace(lib.smplr_prelu_bwd_workspace(N, C, HW), x)
        check(lib.smplr_prelu_bwd(ptr(x), ptr(weight), ptr(gy), N, C, HW, ptr(gx), ptr(gw), ptr(ws), stream()),
              "smplr_prelu_bwd")
        return gx, gw


class BatchNormActFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ per-channel PReLU) on NCHW fp32: smplr_bn_fwd / smplr_bn_bwd (the ENet
    encoder's `BatchNormalization` + `PReLU(shared_axes=[1, 2])` pairs, encoders/encoder_enet_simple.py:19-21).
    running_mean / running_var are updated in place like torch.nn.BatchNorm2d; slope = None: no activation."""

    @staticmethod
    @on_device
    def forward(ctx, x, gamma, beta, slope, running_mean, running_var, eps, momentum):
        lib = _lib.load()
        x = require_cuda(x, "x")
        gamma, beta = require_cuda(gamma, "gamma"), require_cuda(beta, "beta")
        slope = require_cuda(slope, "slope") if slope is not None else None
        N, C = x.shape[0], x.shape[1]
        if gamma.numel() != C or beta.numel() != C or (slope is not None and slope.numel() != C):
            raise RuntimeError("batch norm parameters need %d entries" % C)
        HW = x.numel() // (N * C) if N * C else 1
        z = torch.empty_like(x)
        mean, rstd = _empty((C,), x), _empty((C,), x)
        ws = _workspace(lib.smplr_bn_workspace(N, C, HW), x)
        check(lib.smplr_bn_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(slope), N, C, HW, float(eps), float(momentum),
                               ptr(running_mean), ptr(running_var), ptr(z), ptr(mean), ptr(rstd), ptr(ws), stream()),
              "smplr_bn_fwd")
        ctx.save_for_backward(x, gamma, beta, slope, mean, rstd)
        ctx.dims = (N, C, HW)
        return z

    @staticmethod
    @on_device
    def backward(ctx, dz):
        lib = _lib.load()
        x, gamma, beta, slope, mean, rstd = ctx.saved_tensors
        N, C, HW = ctx.dims
        dz = require_cuda(dz, "dz")
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        ds = torch.empty_like(slope) if slope is not None else None
        ws = _workspace(lib.smplr_bn_workspace(N, C, HW), x)
        check(lib.smplr_bn_bwd(ptr(x), ptr(gamma), ptr(beta), ptr(slope), ptr(mean), ptr(rstd), ptr(dz), N, C, HW,
                               ptr(dx), ptr(dg), ptr(db), ptr(ds), ptr(ws), stream()), "smplr_bn_bwd")
        return dx, dg, db, ds, None, None, None, None


class BatchNormResActFn(torch.autograd.Function):
    """out = prelu(plane_scale * bn(x) + other, slope): the tail of an ENet bottleneck (BatchNormalization ->
    SpatialDropout2D -> Add -> PReLU, encoders/encoder_enet_simple.py:56-79) as one op (smplr_bn_res_fwd/bwd)."""

    @staticmethod
    @on_device
    def forward(ctx, x, other, gamma, beta, slope, plane_scale, running_mean, running_var, eps, momentum):
        lib = _lib.load()
        x, other = require_cuda(x, "x"), require_cuda(other, "other")
        if other.shape != x.shape:
            raise RuntimeError("other must have the shape of x")
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C) if N * C else 1
        out = torch.empty_like(x)
        mean, rstd = _empty((C,), x), _empty((C,), x)
        ws = _workspace(lib.smplr_bn_workspace(N, C, HW), x)
        check(lib.smplr_bn_res_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(plane_scale), ptr(other), ptr(slope), N, C, HW,
                                   float(eps), float(momentum), ptr(running_mean), ptr(running_var), ptr(out),
                                   ptr(mean), ptr(rstd), ptr(ws), stream()), "smplr_bn_res_fwd")
        ctx.save_for_backward(x, other, gamma, beta, slope, plane_scale, mean, rstd)
        ctx.dims = (N, C, HW)
        return out

    @staticmethod
    @on_device
    def backward(ctx, dout):
        lib = _lib.load()
        x, other, gamma, beta, slope, plane_scale, mean, rstd = ctx.saved_tensors
        N, C, HW = ctx.dims
        dout = require_cuda(dout, "dout")
        dx, dother = torch.empty_like(x), torch.empty_like(x)
        dg, db, ds = torch.empty_like(gamma), torch.empty_like(beta), torch.empty_like(slope)
        ws = _workspace(lib.smplr_bn_workspace(N, C, HW), x)
        check(lib.smplr_bn_res_bwd(ptr(x), ptr(gamma), ptr(beta), ptr(plane_scale), ptr(other), ptr(slope), ptr(mean),
                                   ptr(rstd), ptr(dout), N, C, HW, ptr(dx), ptr(dother), ptr(dg), ptr(db), ptr(ds),
                                   ptr(ws), stream()), "smplr_bn_res_bwd")
        return dx, dother, dg, db, ds, None, None, None, None, None


def _bn_fusable(x, bn):
    # (x.is_contiguous(): the kernels read NCHW planes - a channels_last tensor, the opt-in SMPLR_ENCODER_LAYOUT of
    # training.py, takes the stock modules instead of being transposed back for them)
    return (bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.affine and x.is_contiguous()
            and bn.track_running_stats and bn.momentum is not None and x.shape[2] * x.shape[3] >= 256
            and x.shape[0] > 0)


def batch_norm_residual_act(x, bn, dropout, other, act, plane_scale=None):
    """`act(dropout(bn(x)) + other)` for nn.BatchNorm2d, nn.Dropout2d (or None), a tensor and a per-channel
    nn.PReLU: one HIP op when training on a HIP device, the stock modules otherwise.  plane_scale (N, C): the
    dropout factors to use instead of drawing them (tests)."""
    if not (_bn_fusable(x, bn) and act.weight.numel() == x.shape[1] and other.shape == x.shape):
        y = bn(x)
        if dropout is not None:
            y = dropout(y)
        return act(y + other)
    if plane_scale is None and dropout is not None and dropout.training and dropout.p > 0:
        keep = 1.0 - float(dropout.p)
        plane_scale = torch.empty(x.shape[0], x.shape[1], device=x.device, dtype=torch.float32).bernoulli_(keep).div_(keep)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return BatchNormResActFn.apply(x.contiguous(), other.contiguous(), bn.weight, bn.bias, act.weight,
                                   plane_scale.contiguous() if plane_scale is not None else None,
                                   bn.running_mean, bn.running_var, bn.eps, bn.momentum)


def batch_norm_act(x, bn, act=None):
    """`act(bn(x))` for a torch.nn.BatchNorm2d and an optional per-channel nn.PReLU.  Training mode on a HIP
    device with planes of >= 256 elements runs the fused HIP kernels; everything else (eval mode, CPU, tiny
    planes, other dtypes) takes the stock modules.  Parameters, buffers and state dict are the modules' own."""
    fused = (bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.affine and x.is_contiguous()
             and bn.track_running_stats and bn.momentum is not None and x.shape[2] * x.shape[3] >= 256
             and (act is None or act.weight.numel() == x.shape[1]) and x.shape[0] > 0)
    if not fused:
        y = bn(x)
        return act(y) if act is not None else y
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return BatchNormActFn.apply(x.contiguous(), bn.weight, bn.bias, act.weight if act is not None else None,
                                bn.running_mean, bn.running_var, bn.eps, bn.momentum)


_side_streams = {}


def _chunk_streams(device, k):
    """k side streams per device, created once (HIP streams are cheap to keep)."""
    key = str(device)
    pool = _side_streams.setdefault(key, [])
    while len(pool) < k:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:k]


def _chunk_bounds(B, nchunk):
    nchunk = max(1, min(int(nchunk), B)) if B > 0 else 1
    base, rem = divmod(B, nchunk)
    out, lo = [], 0
    for i in range(nchunk):
        hi = lo + base + (1 if i < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _run_chunks(bounds, device, fn):
    """fn(lo, hi) for every chunk: one chunk inline, several on side streams that fork from and
    join back into the current stream (parallel branches when captured into a HIP graph)."""
    if len(bounds) == 1:
        fn(*bounds[0])
        return
    cur = torch.cuda.current_stream(device)
    streams = _chunk_streams(device, len(bounds))
    for st, (lo, hi) in zip(streams, bounds):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            fn(lo, hi)
    for st in streams:
        cur.wait_stream(st)


@dataclass
class DecoderOpts:
    """What a decoder pass computes and writes out (DecoderFn).  Defaults = everything the reference's four model
    handles expose (model.py:124-153): verts, projects, mask and the 31-part scores."""
    want_verts: bool = True      # write verts (B,V,3) out; the backward needs none of verts / projects / mask
    want_proj: bool = True       # (forced on with a silhouette head: the silhouette rasteriser reads it)
    want_mask: bool = True
    seg: bool = True             # the 31-part head (compute_mask + projects_to_seg); False: silhouette-only pass
    want_seg: bool = True        # with `loss`: also write the (B,W,W,32) scores (not differentiable then)
    # the loss head fused into the rasteriser (model.py:119-120 + focal_loss.py:10-46 at an integer class map):
    # (labels (B,W,W) integer, class_w (32,) or None, gamma) -> the pass returns the per-pixel loss (B, W*W)
    loss: Optional[tuple] = None


# Batch from which the decoder's forward runs the pose kernel and the blend GEMM as two launches instead of one: the
# single launch (pose chain hidden under the GEMM, coefficient rows staged in 116 KB of LDS) wins while the grid is one
# round of workgroups - B = 128: 17.3 us against 7.1 + 13.0 - and loses once it is many, because that LDS allows one
# workgroup per CU where the plain GEMM fits two: B = 512: 52.4 against 7.2 + 42.1, B = 2 048: 191 against 9.2 + 136.4
# (tools/probes/blend_big.py); whole steps on one box: B = 256 0.2237 (one launch) against 0.230 ms, B = 384 0.349
# against 0.3427.  Bit-identical either way.  (Round 4, re-measured, one launch / two: B = 256 0.2149 / 0.2185 ms, 320
# 0.3097 / 0.3162, 384 0.3300 / 0.3323, 448 0.3990 / 0.3844, 512 0.3985 / 0.3853: the crossover lies at about 400.)
POSE_BLEND_SPLIT_B = int(os.environ.get("SMPLR_POSE_BLEND_SPLIT_B", "400"))
# Batches at which the binning workgroups skin their own vertices (one launch less, no re-read of proj): with one
# workgroup per mesh the skinning lengthens a latency chain that a half-empty chip does not feel - B = 128: -5 us - and
# that every CU pays once the chip is full.  Round 4 (vslot straight to memory, requests split around the first barrier)
# moved the crossover: whole step, fused / separate skinning launch, same box: B = 256 0.2047 / 0.2149 ms, 384 0.3256 /
# 0.3323, 448 0.3829 / 0.3819, 512 0.3988 / 0.3894, 640 0.5103 / 0.5017, 768 0.5824 / 0.5832, 1 024 0.7144 / 0.7240,
# 1 536 1.1352 / 1.1468, 2 048 1.3998 / 1.4347 - fused below 448 meshes and again from 768 on (where the separate
# launch's re-read of proj and verts is what every CU pays).
FUSE_SKIN_BELOW_B = int(os.environ.get("SMPLR_FUSE_SKIN_BELOW_B", "448"))
FUSE_SKIN_FROM_B = int(os.environ.get("SMPLR_FUSE_SKIN_FROM_B", "768"))


SILH_HINT_MAX_W = 48     # smplr_silh_fwd_hint uses the hint in silh_px_kernel only (W <= 48, raster.hip)


class DecoderFn(torch.autograd.Function):
    """The model.py:108-118 chain as ONE autograd node.

    x (B, 86) -> verts (B,V,3), proj (B,V',3), mask (B,V'), seg (B,W,W,32), silh (B,W,W,2), J_transformed, loss (B,W*W)
    (outputs that were not asked for - DecoderOpts - come back as empty tensors).
    The projection is the skinning kernel's epilogue, the mask is computed in between, and the
    backward fuses d(seg)/d(silh)/d(verts)/d(proj) into one skinning-backward launch.  With `opts.loss` the loss
    head runs as the rasteriser's epilogue and its backward inside the rasteriser's backward: the (B,W,W,32) scores
    and their gradient never exist in memory.  With `opts.seg = False` only the silhouette head is rendered (the
    reference's alternating stage-2 schedule, train_stage2_silhouette.py:262-270).

    Every op is independent per mesh, so the batch may be cut into `nchunk` contiguous chunks
    whose kernel sequences run concurrently on separate HIP streams: at B = 128 most kernels are
    latency-bound (a fraction of a wave per SIMD), and concurrent chunks fill the idle SIMDs.
    Results are identical for any nchunk.
    """

    @staticmethod
    @on_device
    def forward(ctx, x, consts: SMPLConstants, num_cam, img_wh, vertex_sampling, pt: PartTable,
                grid_wh, ref_compat, with_silh, nchunk=1, deterministic=False, opts: Optional[DecoderOpts] = None):
        x = require_cuda(x, "x")
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        opts = opts if opts is not None else DecoderOpts()
        vs, W, B = int(vertex_sampling), int(img_wh), x.shape[0]
        # with_silh: False = no silhouette; True = at img_wh; an int = its own resolution
        # (train_stage2_silhouette.py:72-86 renders the silhouettes at `silhs_output_wh`)
        Ws = 0 if not with_silh else (W if with_silh is True else int(with_silh))
        if not opts.seg and not with_silh:
            raise RuntimeError("DecoderFn: no head asked for (opts.seg = False needs a silhouette)")
        # (smplr_seg_bin reads grid_wh <= 0 as "the mask is an INPUT": the decoder always computes it, so a bad value
        # would rasterise from an uninitialised mask on the two-call path - refuse it here for every path)
        if opts.seg and not 0 < int(grid_wh) <= 128:
            raise RuntimeError("DecoderFn: grid_wh must be in 1..128 (got %r)" % (grid_wh,))
        V, VP = consts.V, (consts.V + vs - 1) // vs
        dev = x.device
        none = lambda: torch.empty(0, device=dev)
        Rs, J = _empty((B, 24, 9), x), _empty((B, 24, 3), x)
        A, Jt = _empty((B, 24, 12), x), _empty((B, 24, 3), x)
        v_posed = _empty((B, V, 3), x)
        want_proj = opts.want_proj or bool(with_silh)
        verts = _empty((B, V, 3), x) if opts.want_verts else None
        proj = _empty((B, VP, 3), x) if want_proj else None
        mask = _empty((B, VP), x) if (opts.want_mask and opts.seg) else None
        loss_spec = opts.loss if opts.seg else None
        labels = class_w = None
        gamma = 0.0
        if loss_spec is not None:
            labels, class_w, gamma = loss_spec
            if pt.P != 31:
                raise RuntimeError("the fused loss head is the 32-class one (P = 31)")
            if labels.numel() != B * W * W or labels.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8):
                raise RuntimeError("fused loss: labels must be an integer class map of %d x %d x %d entries" % (B, W, W))
            labels = require_cuda(labels.to(torch.int32).reshape(B, W, W), "labels", torch.int32)
            if class_w is not None:
                class_w = require_cuda(class_w, "class_w")
                if class_w.numel() != 32:
                    raise RuntimeError("class_w needs 32 entries")
        want_seg = opts.seg and (opts.want_seg or loss_spec is None)
        seg = _empty((B, W, W, pt.P + 1), x) if want_seg else None
        if opts.seg:
            arg = _empty((B, W, W, 32), x, torch.int16)
            rec = _empty((B, lib.smplr_seg_slots(pt.P, pt.K), 4), x)
            vslot = _empty((B, VP), x, torch.int16)
        else:
            arg = rec = vslot = None
        loss = _empty((B, W * W), x) if loss_spec is not None else None
        stats = _empty((B, W * W, 4), x) if loss_spec is not None else None
        if with_silh:
            silh, sarg = _empty((B, Ws, Ws, 2), x), _empty((B, Ws, Ws), x, torch.int32)
        else:
            silh = sarg = None
        # both heads at one resolution: the part rasteriser hands the silhouette rasteriser each pixel's largest part
        # score - an upper bound of the distance to the nearest vertex that spares it its own search for one
        # (only where the silhouette rasteriser reads it: its pixel-per-lane kernel, W <= SILH_HINT_MAX_W - beyond that
        # the step would pay for a (B,W,W) store and the slower `_ex` launch shape for a hint nobody uses)
        vmax = _empty((B, W, W), x) if (with_silh and opts.seg and Ws == W and W <= SILH_HINT_MAX_W and pt.P == 31
                                        and os.environ.get("SMPLR_SILH_HINT", "1") != "0") else None

        # the binning workgroups skin their own vertices (one launch less) when the skinning rows are sparse, every
        # vertex is rasterised and the mesh fits the binning workgroup's LDS; SMPLR_FUSE_SKIN=0 keeps the two calls
        fuse_skin = (opts.seg and consts.lbs_top4 is not None and vs == 1
                     and bool(lib.smplr_skin_vis_seg_fits(V, W, int(grid_wh)))
                     and (B < FUSE_SKIN_BELOW_B or B >= FUSE_SKIN_FROM_B) and os.environ.get("SMPLR_FUSE_SKIN", "1") != "0")
        sl = lambda t, lo, hi: None if t is None else t[lo:hi]

        def run(lo, hi):
            xs = x[lo:hi]
            n = hi - lo
            if consts.blend3_fwd is not None and n < POSE_BLEND_SPLIT_B:
                _pose_blend_fwd(xs, num_cam, consts, out=(Rs[lo:hi], J[lo:hi], A[lo:hi], Jt[lo:hi]), v_posed=v_posed[lo:hi])
            else:
                # (also the bf16x3 constants at large batch: the same bits from two launches, see POSE_BLEND_SPLIT_B)
                coef = _pose_fwd(xs, num_cam, consts, out=(None, Rs[lo:hi], J[lo:hi], A[lo:hi], Jt[lo:hi]))[0]
                _blend_fwd(coef, consts, n, out=v_posed[lo:hi])
            if fuse_skin and (loss_spec is not None or vmax is not None):
                _skin_vis_seg_fwd_ex(
                    v_posed[lo:hi], A[lo:hi], consts, xs, W, pt, sl(labels, lo, hi), class_w, gamma, grid_wh, ref_compat,
                    verts=sl(verts, lo, hi), proj=sl(proj, lo, hi), mask=sl(mask, lo, hi), seg=sl(seg, lo, hi),
                    arg=arg[lo:hi], rec=rec[lo:hi], vslot=vslot[lo:hi], loss=sl(loss, lo, hi), stats=sl(stats, lo, hi),
                    vmax=sl(vmax, lo, hi))
            elif fuse_skin:
                _skin_vis_seg_opt(v_posed[lo:hi], A[lo:hi], consts, xs, W, pt, grid_wh, ref_compat,
                                  sl(verts, lo, hi), sl(proj, lo, hi), sl(mask, lo, hi), seg[lo:hi], arg[lo:hi],
                                  rec[lo:hi], vslot[lo:hi])
            else:
                # the two-call path needs proj (and the mask) in memory whatever the caller wants back
                pj = proj[lo:hi] if proj is not None else _empty((n, VP, 3), x)
                _skin_fwd(v_posed[lo:hi], A[lo:hi], consts, cam=xs, vertex_sampling=vs, want_verts=verts is not None,
                          out=(sl(verts, lo, hi), pj))
                if opts.seg:
                    mk = mask[lo:hi] if mask is not None else _empty((n, VP), x)
                    if loss_spec is not None or vmax is not None:
                        ws_, _ = _seg_bin(pj, mk, W, pt, grid_wh, ref_compat, rec=rec[lo:hi], vslot=vslot[lo:hi])
                        _seg_raster_ex(ws_, rec[lo:hi], n, W, pt, sl(labels, lo, hi), class_w, gamma, seg=sl(seg, lo, hi),
                                       arg=arg[lo:hi], loss=sl(loss, lo, hi), stats=sl(stats, lo, hi), vmax=sl(vmax, lo, hi))
                    else:
                        _vis_seg_fwd(pj, W, pt, grid_wh, ref_compat,
                                     out=(mk, seg[lo:hi], arg[lo:hi], rec[lo:hi]), vslot=vslot[lo:hi])
            if with_silh:
                _silh_fwd(proj[lo:hi], Ws, out=(silh[lo:hi], sarg[lo:hi]), hint=sl(vmax, lo, hi))

        bounds = _chunk_bounds(B, nchunk)
        if B > 0:
            _run_chunks(bounds, x.device, run)
        ctx.consts, ctx.num_cam, ctx.W, ctx.vs, ctx.pt, ctx.with_silh = consts, num_cam, W, vs, pt, bool(with_silh)
        ctx.Ws, ctx.VP = Ws, VP
        ctx.det = bool(deterministic)
        ctx.bounds = bounds
        ctx.has_seg, ctx.has_loss = bool(opts.seg), loss_spec is not None
        E = none
        saved = [x, Rs, J, A, v_posed, proj if with_silh else E(), arg if opts.seg else E(), rec if opts.seg else E(),
                 silh if with_silh else E(), sarg if with_silh else E(), vslot if opts.seg else E(),
                 stats if loss_spec is not None else E()]
        ctx.save_for_backward(*saved)
        outs = [t if t is not None else E() for t in (verts, proj, mask, seg, silh, Jt, loss)]
        ctx.mark_non_differentiable(outs[2])
        if loss_spec is not None:
            ctx.mark_non_differentiable(outs[3])         # with a fused loss the scores are a by-product, not a path
        return tuple(outs)

    @staticmethod
    @on_device
    def backward(ctx, dverts, dproj_in, _dmask, dseg, dsilh, dJt, dloss):
        x, Rs, J, A, v_posed, proj, arg, rec, silh, sarg, vslot, stats = ctx.saved_tensors
        dseg = require_cuda(dseg, "dseg") if (dseg is not None and ctx.has_seg and not ctx.has_loss) else None
        dloss = require_cuda(dloss, "dloss") if (dloss is not None and ctx.has_loss) else None
        dsilh = require_cuda(dsilh, "dsilh") if (ctx.with_silh and dsilh is not None) else None
        dproj_in = require_cuda(dproj_in, "dproj") if dproj_in is not None else None
        dverts = require_cuda(dverts, "dverts") if dverts is not None else None
        dJt = require_cuda(dJt, "dJ_transformed") if dJt is not None else None
        VP = ctx.VP
        dx = _empty(tuple(x.shape), x)

        def run(lo, hi):
            dproj, seg_grad = None, None
            if dseg is not None:
                # the slot sums stay in the workspace; the skinning backward gathers them by vertex
                part, nsplit = _seg_bwd(dseg[lo:hi], arg[lo:hi], rec[lo:hi], VP, ctx.W, ctx.pt, merge=False,
                                        deterministic=ctx.det)
                seg_grad = (part, vslot[lo:hi], nsplit)
            elif dloss is not None:
                part, nsplit = _seg_loss_bwd(dloss[lo:hi], stats[lo:hi], arg[lo:hi], rec[lo:hi], VP, ctx.W, ctx.pt,
                                             merge=False, deterministic=ctx.det)
                seg_grad = (part, vslot[lo:hi], nsplit)
            if dsilh is not None:
                d2 = _silh_bwd(dsilh[lo:hi], silh[lo:hi], sarg[lo:hi], proj[lo:hi], ctx.Ws, ctx.det)
                dproj = d2 if dproj is None else dproj + d2
            if dproj_in is not None:
                dproj = dproj_in[lo:hi] if dproj is None else dproj + dproj_in[lo:hi]
            dv = dverts[lo:hi] if dverts is not None else None
            if dv is None and dproj is None and seg_grad is None:
                dv = torch.zeros_like(v_posed[lo:hi])
            _smpl_bwd(x[lo:hi], ctx.num_cam, ctx.consts, Rs[lo:hi], J[lo:hi], A[lo:hi], v_posed[lo:hi], dv,
                      dproj, dJt[lo:hi] if dJt is not None else None, ctx.vs, out=dx[lo:hi], seg_grad=seg_grad)

        if x.shape[0] > 0:
            _run_chunks(ctx.bounds, x.device, run)
        return (dx,) + (None,) * 11


@on_device
def _skin_vis_seg_opt(v_posed, A, c, cam, W, pt, grid_wh, ref_compat, verts, proj, mask, seg, arg, rec, vslot):
    """smplr_skin_vis_seg_fwd with optional verts / proj / mask (None = not written)."""
    lib = _lib.load()
    B, V = v_posed.shape[0], c.V
    ws = _workspace(lib.smplr_seg_workspace(B, V, W, pt.P, pt.K), v_posed)
    check(lib.smplr_skin_vis_seg_fwd(ptr(v_posed), ptr(c.lbs_top4), ptr(A), ptr(cam), cam.shape[1], B, V, W,
                                     int(grid_wh), 1 if ref_compat else 0, ptr(pt.part_pos), ptr(pt.part_off), pt.P,
                                     pt.K, ptr(ws), ptr(verts), ptr(proj), ptr(mask), ptr(seg), ptr(arg), ptr(rec),
                                     ptr(vslot), stream()), "smplr_skin_vis_seg_fwd")
