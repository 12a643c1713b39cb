"""Train-step structure of the reference (train.py:179-242, train_stage2_silhouette.py:226-270) on the
MI355X path: forward (encoder + regressor on stock torch ops, decoder on the HIP kernels), focal
loss, backward, Adam(lr=1e-4).  Data parallelism replaces `keras.utils.multi_gpu_model`
(train.py:205-210): one process per GPU, `DistributedDataParallel` over RCCL (backend "nccl"),
gradients of the encoder/regressor all-reduced in ~25 MiB buckets overlapped with backward; the
decoder has no parameters and exchanges nothing.  BatchNorm stays local per rank,
as Keras towers normalise per tower."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import nn

from .decoder import SMPLDecoder
from .focal_loss import softmax_focal_loss
from .model import SMPLRegressor


class SegTrainer:
    """One optimiser step = `segs_model.fit` on one batch with the focal loss (train.py:207-242)."""

    def __init__(self, smpl_path=None, input_wh=256, output_wh=48, encoder_architecture="enet", use_IEF=True,
                 weight_classes=True, gamma=2.0, lr=1e-4, device=None, ddp=False, bucket_mb=25,
                 with_silhouette=False):
        self.device = (torch.device(device) if device is not None
                       else torch.device("cuda", torch.cuda.current_device()))
        self.output_wh = output_wh
        self.smpl_model = SMPLRegressor(output_wh, encoder_architecture, use_IEF).to(self.device)
        self.decoder = SMPLDecoder(smpl_path, img_wh=output_wh, with_silhouette=with_silhouette)
        self.with_silhouette = with_silhouette
        self.net = self.smpl_model
        if ddp:
            self.net = nn.parallel.DistributedDataParallel(
                self.smpl_model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True)
        self.loss_fn = softmax_focal_loss(gamma, weight_classes)      # softmax + focal loss, one HIP kernel
        self.silh_loss_fn = softmax_focal_loss(0.0, False)            # softmax + categorical CE
        self.opt = torch.optim.Adam(self.smpl_model.parameters(), lr=lr)       # train.py:179

    def step(self, images, labels, silh_labels=None):
        """images (N,3,H,W) or (N,H,W,3); labels (N,W,W) integer class map or (N,W*W,32) one-hot.
        Returns the mean loss (a 0-d tensor; no host sync)."""
        self.opt.zero_grad(set_to_none=True)
        param = self.net(images)
        out = self.decoder(param)
        loss = self.loss_fn(labels, out["seg"]).mean()             # model.py:119-120 + focal_loss.py
        if self.with_silhouette and silh_labels is not None:      # train_stage2_silhouette.py:85-86,226-229
            loss = loss + self.silh_loss_fn(silh_labels, out["silhouette"]).mean()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def state_dict(self):                                           # train.py:302-315 saves smpl_model only
        return self.smpl_model.state_dict()


def init_distributed():
    """Rank/world from torchrun's environment; RCCL over xGMI on GPUs, gloo on CPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group("gloo")
    return int(os.environ.get("RANK", "0")), world
