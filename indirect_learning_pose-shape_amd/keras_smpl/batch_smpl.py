"""SMPLLayer: drop-in for `keras_smpl/batch_smpl.py:23-153` of the reference, on HIP kernels.

forward = smplr_pose_fwd -> smplr_blend3_fwd (bf16x3 MFMA GEMM; the fp32-MFMA smplr_blend_fwd with
SMPLR_BLEND_GEMM=f32) -> smplr_skin_fwd; the backward is hand-written (smplr_smpl_bwd = skin, blend, pose).  SMPL constants are
non-trainable buffers, exactly as in the reference (`batch_smpl.py:92-94`).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..smpl_model import SMPLModelData, synthetic_smpl_model


def _resolve_model(pkl_path_or_model) -> SMPLModelData:
    if isinstance(pkl_path_or_model, SMPLModelData):
        return pkl_path_or_model
    if pkl_path_or_model is None or pkl_path_or_model == "synthetic":
        return synthetic_smpl_model()
    from ..smpl_pkl import load_smpl_pkl      # real (licensed) SMPL pickle supplied by the user
    return load_smpl_pkl(pkl_path_or_model)


class SMPLLayer(nn.Module):
    """`SMPLLayer(pkl_path, batch_size=8, dtype='float32', joint_type='lsp')` (batch_smpl.py:23-29).

    `pkl_path` may be a path to the real SMPL pickle, an `SMPLModelData`, or None/"synthetic" for
    the seeded SMPL-shaped model.  `batch_size` is accepted for signature compatibility and
    ignored (the reference bakes it into the graph at batch_smpl.py:136-142; here it is dynamic).
    Input x: (B, num_cam + 72 + 10); output verts (B, 6890, 3).  After a call `J_transformed`
    holds the (B,24,3) posed joints (batch_smpl.py:131) and takes part in autograd.
    """

    def __init__(self, pkl_path=None, batch_size=None, dtype=torch.float32, joint_type="lsp",
                 num_cam=4, device=None):
        super().__init__()
        if dtype not in (torch.float32, "float32"):
            raise ValueError("SMPLLayer computes in float32 (the reference's dtype, batch_smpl.py:23)")
        self.pkl_path = pkl_path if isinstance(pkl_path, str) else None
        self.batch_size = batch_size
        self.joint_type = joint_type
        self.num_cam = int(num_cam)
        self.num_joints = 24
        self.num_thetas = 72
        self.num_betas = 10
        self._model = _resolve_model(pkl_path)
        self.size = [self._model.num_verts, 3]
        self._consts = None
        self._consts_device = None
        self.J_transformed = None
        if device is not None:
            self.constants(torch.device(device))

    def constants(self, device) -> ops.SMPLConstants:
        if self._consts is None or self._consts_device != device:
            self._consts = ops.SMPLConstants.from_model(self._model, device, self.joint_type)
            self._consts_device = device
        return self._consts

    def forward(self, x):
        if x.dim() != 2 or x.shape[1] != self.num_cam + self.num_thetas + self.num_betas:
            raise RuntimeError("SMPLLayer expects x of shape (B, %d), got %s"
                               % (self.num_cam + 82, tuple(x.shape)))
        verts, jt = ops.BatchSMPLFn.apply(x, self.constants(x.device), self.num_cam)
        self.J_transformed = jt
        return verts

    def joints(self, verts):
        """cocoplus (19) / lsp (14) joints from posed vertices: the output the reference leaves
        commented out at batch_smpl.py:147-151.  A plain library GEMM (rocBLAS), (B,V,3) -> (B,Jn,3)."""
        jr = self.constants(verts.device).joint_regressor
        if jr is None:
            raise RuntimeError("this SMPL model has no cocoplus_regressor")
        return torch.einsum("bvc,vj->bjc", verts, jr)

    def compute_output_shape(self, input_shape):        # batch_smpl.py:155-159
        return (input_shape[0], self.size[0], self.size[1])

    def get_config(self):                                # batch_smpl.py:161-166
        return {"pkl_path": self.pkl_path, "batch_size": self.batch_size, "dtype": "float32"}

    # Public helpers of the reference class (batch_smpl.py:168, :230, :255).  They are not on the
    # hot path (smplr_pose_fwd fuses all three); provided for API compatibility on device tensors.
    @staticmethod
    def batch_skew(vec, input_size=None):
        z = torch.zeros_like(vec[:, 0])
        return torch.stack([z, -vec[:, 2], vec[:, 1], vec[:, 2], z, -vec[:, 0],
                            -vec[:, 1], vec[:, 0], z], dim=1).reshape(-1, 3, 3)

    def batch_rodrigues(self, theta, batch_size=None):
        """theta (N,3) -> (N,3,3) via the pose kernel (24 joints per row, zero padded)."""
        n = theta.shape[0]
        rows = (n + 23) // 24
        th = torch.zeros(rows * 24, 3, dtype=torch.float32, device=theta.device)
        th[:n] = theta
        x = torch.cat([th.view(rows, 72), th.new_zeros(rows, 10)], dim=1)      # [theta(72) | beta = 0]
        c = self.constants(theta.device)
        _, Rs, _, _, _ = ops._pose_fwd(x, 0, c)
        return Rs.reshape(-1, 3, 3)[:n]

    @staticmethod
    def batch_global_rigid_transformation(Rs, Js, parent, rotate_base=False):
        """batch_smpl.py:168-228 on device tensors (plain torch: the hot path runs smplr_pose_fwd).  rotate_base=True
        (:185-190, never used by the reference's callers) turns the root rotation by diag(1, -1, -1)."""
        N = Rs.shape[0]
        if rotate_base:
            rot_x = Rs.new_tensor([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])
            Rs = torch.cat([(Rs[:, 0] @ rot_x).unsqueeze(1), Rs[:, 1:]], dim=1)
        res = [None] * 24
        bottom = Rs.new_tensor([0, 0, 0, 1.0]).expand(N, 1, 4)
        mk = lambda R, t: torch.cat([torch.cat([R, t.unsqueeze(-1)], 2), bottom], 1)
        res[0] = mk(Rs[:, 0], Js[:, 0])
        for i in range(1, 24):
            p = int(parent[i])
            res[i] = res[p] @ mk(Rs[:, i], Js[:, i] - Js[:, p])
        G = torch.stack(res, 1)
        new_J = G[:, :, :3, 3]
        init = G @ torch.cat([Js, Js.new_zeros(N, 24, 1)], 2).unsqueeze(-1)
        A = G - torch.nn.functional.pad(init, (3, 0))
        return new_J, A
