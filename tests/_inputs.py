"""Seeded synthetic inputs: x = mean86(W) + [cam | theta | beta] noise."""
import numpy as np


def make_x(B, W, seed=0, theta_sigma=0.2, beta_sigma=1.0):
    from ilps_amd.smpl_model import mean86
    rng = np.random.default_rng(seed)
    x = np.tile(mean86(W), (B, 1))
    x[:, 0:2] += rng.normal(0.0, 1.0, (B, 2))
    x[:, 2:4] += rng.normal(0.0, 0.05 * W, (B, 2))
    x[:, 4:76] += rng.normal(0.0, theta_sigma, (B, 72))
    x[:, 76:86] += rng.normal(0.0, beta_sigma, (B, 10))
    return x.astype(np.float32)
