"""categorical_focal_loss: counterpart of `focal_loss.py:10-46` of the reference (gamma = 2 by default,
optional fixed class weights).  y_pred are softmax probabilities (N, W*W, 32) as produced by the
'segs' model; labels may be one-hot (the reference's format, train.py:18-31) or integer class maps."""
from __future__ import annotations

import torch

_EPS = 1e-7        # K.epsilon()

# focal_loss.py:22-40: up-weight hands, elbows, knees, ankles; down-weight background
CLASS_WEIGHTS = [1.0] * 32
CLASS_WEIGHTS[0] = 0.3
for _c in (1, 2, 3, 4, 10, 12, 14, 15, 16, 17, 23, 25):
    CLASS_WEIGHTS[_c] = 2.0


_class_w_cache = {}


def class_weights(device, dtype=torch.float32):
    """The fixed weight vector on `device`, uploaded once per (device, dtype): a fresh `torch.tensor(list)` per
    step is a pageable host-to-device copy, i.e. a host sync in every train step and illegal under graph capture."""
    key = (str(device), dtype)
    w = _class_w_cache.get(key)
    if w is None:
        w = _class_w_cache[key] = torch.tensor(CLASS_WEIGHTS, device=device, dtype=dtype)
    return w


def categorical_focal_loss(gamma=2.0, weight_classes=False):
    def categorical_focal_loss_fixed(y_true, y_pred):
        """Returns the per-pixel loss (N, W*W), like the Keras loss function (Keras then averages)."""
        p = y_pred.clamp(_EPS, 1.0 - _EPS)                                  # :17
        if y_true.dtype in (torch.int64, torch.int32, torch.uint8, torch.int16):
            idx = y_true.reshape(p.shape[0], -1, 1).long()
            pt = p.gather(2, idx).squeeze(2)
            w = class_weights(p.device, p.dtype)[idx.squeeze(2)] if weight_classes else 1.0
            return w * (1.0 - pt) ** gamma * (-torch.log(pt))
        ce = -y_true * torch.log(p)                                         # :18
        if weight_classes:
            ce = ce * class_weights(p.device, p.dtype)                      # :20-41
        return ((1.0 - p) ** gamma * ce).sum(dim=2)                         # :43-44
    return categorical_focal_loss_fixed


def softmax_focal_loss(gamma=2.0, weight_classes=False):
    """The fused loss head on the HIP path: `loss(y_true, scores)` with RAW rasteriser scores
    (N,W,W,C) or (N,W*W,C) -- Reshape + softmax (model.py:119-120) and the focal loss above in one
    kernel forward, one backward (smplr_focal_fwd/bwd).  y_true: integer class map or one-hot.
    gamma=0, weight_classes=False is Keras' categorical_crossentropy (the silhouette head)."""
    def loss(y_true, scores):
        from . import ops
        w = class_weights(scores.device)[: scores.shape[-1]].contiguous() if weight_classes else None
        return ops.SoftmaxFocalFn.apply(scores, y_true, w, float(gamma))
    # what `SMPLDecoder(loss=...)` reads to run this head inside the rasteriser instead (smplr_skin_vis_seg_fwd_ex)
    loss.gamma, loss.weight_classes = float(gamma), bool(weight_classes)
    return loss


def classlab(labels, num_classes=32):
    """train.py:18-31 (`classlab`): (..., H, W[,1]) integer label image -> one-hot (..., H, W, C).
    The reference does this with a Python double loop per image; here it is one scatter on device."""
    if labels.dim() >= 3 and labels.shape[-1] == 1:
        labels = labels[..., 0]
    return torch.nn.functional.one_hot(labels.long(), num_classes).to(torch.float32)
