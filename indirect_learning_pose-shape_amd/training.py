"""Train-step structure of the reference (train.py:179-242, train_stage2_silhouette.py:226-270) on the
MI355X path: forward (encoder + regressor on stock torch ops, decoder on the HIP kernels), focal
loss, backward, Adam(lr=1e-4).  Data parallelism replaces `keras.utils.multi_gpu_model`
(train.py:205-210): one process per GPU, `DistributedDataParallel` over RCCL (backend "nccl"),
gradients of the encoder/regressor all-reduced in ~25 MiB buckets overlapped with backward; the
decoder has no parameters and exchanges nothing.  BatchNorm stays local per rank
(statistics AND running buffers: `broadcast_buffers=False`), as Keras towers normalise per tower.
`with_silhouette` adds the silhouette cross-entropy of train_stage2_silhouette.py:226-229 to the same step - the
reference alternates two separately compiled models over one shared encoder; here both heads come out of one
decoder pass and one backward - at its own resolution `silh_wh` (`silhs_output_wh`, :72-86)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import nn

from .decoder import SMPLDecoder
from .focal_loss import softmax_focal_loss
from .model import SMPLRegressor


def configure_conv_backend():
    """OPT-IN (`SMPLR_CONV_BACKEND=find`): MIOpen find mode with torch's full workspace (`cudnn.benchmark = True`,
    MIOPEN_FIND_MODE=FAST unless set) for the encoder's stock-torch convolutions.  Not the default: on a fresh box (no
    user database, no cached kernels) the first train step then spends MINUTES compiling and timing candidate solvers -
    `tools/conv_backend_ab.sh` on an MI355X: default 28.9 s for the whole run of 8 steps at 26.07 ms per step and no
    workspace warning; find mode silent for over 420 s (killed).  The `GemmFwdRest ... IsEnoughWorkspace` lines a
    `bench.py` run prints (8 of them) come from the batch-1 predict leg's first convolution: immediate mode EVALUATES
    that solver against the workspace torch sized for the solution it picked and logs that it does not fit; the step
    above, at 128 images, prints none."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    torch.backends.cudnn.benchmark = True


class SegTrainer:
    """One optimiser step = `segs_model.fit` on one batch with the focal loss (train.py:207-242)."""

    def __init__(self, smpl_path=None, input_wh=256, output_wh=48, encoder_architecture="enet", use_IEF=True,
                 weight_classes=True, gamma=2.0, lr=1e-4, device=None, ddp=False, bucket_mb=25,
                 with_silhouette=False, silh_wh=None, fused_loss=True):
        self.device = (torch.device(device) if device is not None
                       else torch.device("cuda", torch.cuda.current_device()))
        if self.device.type == "cuda" and os.environ.get("SMPLR_CONV_BACKEND", "default") == "find":
            configure_conv_backend()
        self.output_wh = output_wh
        self.smpl_model = SMPLRegressor(output_wh, encoder_architecture, use_IEF).to(self.device)
        # OPT-IN (`SMPLR_ENCODER_LAYOUT=channels_last`): the encoder's weights and activations in NHWC, the layout
        # MIOpen's implicit-GEMM convolution solvers work in (the default NCHW tensors are transposed around them:
        # `batched_transpose_32x32_dword` in the train step's kernel trace).  The package's fused batch-norm / PReLU
        # kernels read NCHW planes, so this layout runs the stock modules for them.  Measured before being made a default:
        # tools/train_layout_ab.py.
        self.channels_last = os.environ.get("SMPLR_ENCODER_LAYOUT", "").lower() == "channels_last"
        if self.channels_last:
            self.smpl_model = self.smpl_model.to(memory_format=torch.channels_last)
        self.loss_fn = softmax_focal_loss(gamma, weight_classes)      # softmax + focal loss, one HIP kernel
        self.silh_loss_fn = softmax_focal_loss(0.0, False)            # softmax + categorical CE
        # the train pass consumes losses only: verts / projects / mask are not written out, and with an integer class
        # map the segmentation loss runs inside the rasteriser (no (B,W,W,32) score or gradient tensor in memory)
        self.decoder = SMPLDecoder(smpl_path, img_wh=output_wh, with_silhouette=with_silhouette, silh_wh=silh_wh,
                                   outputs=(), loss=self.loss_fn if fused_loss else None)
        self.with_silhouette = with_silhouette
        # what the reference's monitor step reads every 10 trials (train.py:245-300: `verts_model` / `projects_model` /
        # `segs_model` predictions of the monitor images): verts, projects, mask and the raw scores - the train
        # decoder above writes none of them.  Same constants, no second upload; `on_trial_end(trial, trainer)` hooks
        # use `trainer.monitor(images)`.
        self.monitor_decoder = SMPLDecoder(smpl_path, img_wh=output_wh, with_silhouette=with_silhouette,
                                           silh_wh=silh_wh).share_constants(self.decoder)
        # the silhouette-only pass of the reference's alternating schedule (train_stage2_silhouette.py:262-270)
        self.silh_decoder = (SMPLDecoder(smpl_path, img_wh=output_wh, heads=("silhouette",), silh_wh=silh_wh, outputs=())
                             .share_constants(self.decoder)) if with_silhouette else None
        self.net = self.smpl_model
        if ddp:
            self.net = nn.parallel.DistributedDataParallel(
                self.smpl_model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True, broadcast_buffers=False)
        self.opt = torch.optim.Adam(self.smpl_model.parameters(), lr=lr)       # train.py:179

    def step(self, images, labels, silh_labels=None, _marks=None):
        """images (N,3,H,W) or (N,H,W,3); labels (N,W,W) integer class map or (N,W*W,32) one-hot, or None for the
        silhouette-only step of the alternating schedule (then silh_labels is required).
        Returns the mean loss (a 0-d tensor; no host sync).  (`_marks`: see `step_timed`.)"""
        mark = (lambda k: None) if _marks is None else _marks
        mark("start")
        self.opt.zero_grad(set_to_none=True)
        if self.channels_last and images.dim() == 4 and images.shape[1] == 3:
            images = images.contiguous(memory_format=torch.channels_last)
        param = self.net(images)
        mark("encoder_fwd")
        if _marks is not None and param.requires_grad:
            param.register_hook(lambda g: (mark("decoder_bwd"), g)[1])     # fires between the decoder's and the encoder's backward
        if labels is None:
            # `silhouettes_model.fit` (train_stage2_silhouette.py:226-234, the 150 silhouette steps of :262-270):
            # the silhouette head alone - no mask, no binning, no 31-part rasteriser
            if self.silh_decoder is None or silh_labels is None:
                raise RuntimeError("a step without labels is the silhouette-only step: needs with_silhouette and silh_labels")
            out = self.silh_decoder(param)
            loss = self.silh_loss_fn(silh_labels, out["silhouette"]).mean()
        else:
            is_map = labels.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8)
            if self.decoder.loss is not None and is_map:           # model.py:119-120 + focal_loss.py, in the rasteriser
                loss = self.decoder(param, labels)
                out, loss = loss, loss["seg_loss"].mean()
            else:
                out = self.decoder(param)
                loss = self.loss_fn(labels, out["seg"]).mean()
            if self.with_silhouette and silh_labels is not None:  # train_stage2_silhouette.py:85-86,226-229
                loss = loss + self.silh_loss_fn(silh_labels, out["silhouette"]).mean()
        mark("decoder_fwd")
        loss.backward()
        mark("backward")
        self.opt.step()
        mark("optimizer")
        return loss.detach()

    def step_timed(self, images, labels, silh_labels=None):
        """One `step` with HIP events between its sections -> dict(encoder_ms, decoder_ms, optimizer_ms, total_ms):
        encoder = regressor forward + its backward (under DDP that includes the all-reduce buckets it overlaps),
        decoder = decoder + loss forward and backward (up to the gradient of the 86-vector), optimizer = zero_grad +
        Adam.  Synchronises; a measurement aid for bench.py's train leg (train.py:179-242 is the step reproduced)."""
        ev = {}

        def mark(k):
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            ev[k] = e
        self.step(images, labels, silh_labels, _marks=mark)
        torch.cuda.synchronize(self.device)
        dt = lambda a, b: ev[a].elapsed_time(ev[b])
        dec_bwd_end = "decoder_bwd" if "decoder_bwd" in ev else "backward"
        return {"encoder_ms": dt("start", "encoder_fwd") + dt(dec_bwd_end, "backward"),
                "decoder_ms": dt("encoder_fwd", "decoder_fwd") + dt("decoder_fwd", dec_bwd_end),
                "optimizer_ms": dt("backward", "optimizer"), "total_ms": dt("start", "optimizer")}

    @torch.no_grad()
    def monitor(self, images):
        """The monitor step's predictions (train.py:245-262) for a batch of images: dict(smpl (N,86), verts, projects,
        mask, seg (N,W,W,32) raw scores [, silhouette], J_transformed) from the full-output decoder.  `step()`'s own
        decoder returns losses only (outputs=()), so monitor hooks must not read `trainer.decoder(...)['verts']`."""
        was_training = self.smpl_model.training
        self.smpl_model.eval()
        try:
            param = self.smpl_model(images)
            out = dict(self.monitor_decoder(param))
        finally:
            self.smpl_model.train(was_training)
        out["smpl"] = param
        return out

    def state_dict(self):                                           # train.py:302-315 saves smpl_model only
        return self.smpl_model.state_dict()

    # ---- save / resume (train.py:181-193 `resume_from`, :300-315 `smpl_model.save`) -------------------------
    def save(self, path, trial=0):
        """The regressor's weights (what the reference saves), plus what a bit-exact resume of THIS trainer also
        needs and Keras' `model.save` would have kept: the Adam state, and the trial counter."""
        torch.save({"smpl_model": self.smpl_model.state_dict(), "optimizer": self.opt.state_dict(),
                    "trial": int(trial), "output_wh": int(self.output_wh)}, path)

    def resume(self, path):
        """-> the trial to continue from.  A file holding only a state dict (weights) is accepted too."""
        ck = torch.load(path, map_location=self.device)
        if "smpl_model" not in ck:
            self.smpl_model.load_state_dict(ck)
            return 0
        if int(ck.get("output_wh", self.output_wh)) != int(self.output_wh):
            raise RuntimeError("checkpoint was trained at %dx%d, this trainer renders %dx%d"
                               % (ck["output_wh"], ck["output_wh"], self.output_wh, self.output_wh))
        self.smpl_model.load_state_dict(ck["smpl_model"])
        self.opt.load_state_dict(ck["optimizer"])
        return int(ck.get("trial", -1)) + 1


def save_name(dataset, output_wh, use_IEF=True, scaledown=0.005, vertex_sampling=None, weight_classes=True, trial=0,
              encoder="resnet"):
    """The reference's checkpoint file name (train.py:302-313), with torch's extension."""
    name = "%s_%dx%d_%s" % (dataset, output_wh, output_wh, encoder)
    if use_IEF:
        name += "_ief"
    name += "_scaledown" + str(scaledown).replace(".", "")
    if vertex_sampling is not None:
        name += "_vs%d" % vertex_sampling
    if weight_classes:
        name += "_arms_weighted_2_bg_weighted_0point3_gamma2_multigpu"
    return name + "_%d.pt" % trial


def fit(trainer, batches, trials, steps_per_trial, save_dir=None, save_every=10, name_fn=None, start_trial=0,
        on_trial_end=None):
    """The loop of train.py:221-315: `trials` rounds of `steps_per_trial` optimiser steps (`fit_generator(...,
    steps_per_epoch, nb_epoch=1)`) over `batches` - an iterator of (images, labels[, silhouette labels]) already
    on the device, the data generators being the caller's - and every `save_every` trials (on rank 0) the monitor
    hook `on_trial_end(trial, trainer)` (use `trainer.monitor(images)` for verts / projects / seg: the train decoder
    `trainer.decoder` writes losses only) and a checkpoint named like the reference's.  -> list of per-trial mean losses (python floats; the one
    host sync per trial)."""
    it = iter(batches)
    rank0 = (not dist.is_initialized()) or dist.get_rank() == 0
    history = []
    for trial in range(start_trial, trials):
        total = None
        for _ in range(steps_per_trial):
            batch = next(it)
            loss = trainer.step(*batch)
            total = loss if total is None else total + loss
        history.append(float(total) / max(1, steps_per_trial))
        if trial % save_every == 0:
            if on_trial_end is not None and rank0:
                on_trial_end(trial, trainer)
            if save_dir is not None and rank0:
                os.makedirs(save_dir, exist_ok=True)
                fname = name_fn(trial) if name_fn is not None else "smpl_model_%d.pt" % trial
                trainer.save(os.path.join(save_dir, fname), trial)
    return history


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
