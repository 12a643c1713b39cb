"""concat_mean_param: `keras_smpl/concat_mean_param.py:8-31` (IEF start state)."""
import torch

from .set_cam_params import _row


def concat_mean_param(img_features, img_wh):
    mean = _row(img_wh, True, img_features.device).expand(img_features.shape[0], 86)
    return torch.cat([img_features, mean], dim=1)
