"""set_cam_params / load_mean_set_cam_params: `keras_smpl/set_cam_params.py:13-51`.

A constant row added to the regressed 86-vector; plain torch on the device (SURVEY.md §2 #6:
"a constant add - no kernel").
"""
import numpy as np
import torch

from ..smpl_model import mean86

_cache = {}


def _row(img_wh, with_mean, device):
    key = (float(img_wh), with_mean, str(device))
    if key not in _cache:
        row = mean86(img_wh) if with_mean else np.r_[img_wh / 2.0, img_wh / 2.0, img_wh / 2.0,
                                                      img_wh / 1.6, np.zeros(82)]
        _cache[key] = torch.as_tensor(row, dtype=torch.float32).to(device).unsqueeze(0)
    return _cache[key]


def set_cam_params(smpl, img_wh):
    return smpl + _row(img_wh, False, smpl.device)


def load_mean_set_cam_params(smpl, img_wh):
    return smpl + _row(img_wh, True, smpl.device)
