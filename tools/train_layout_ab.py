"""The train step (ENet + IEF + decoder + losses + Adam, BASELINE configs[3]/[4]) with the encoder in NCHW (default, the
package's fused batch-norm / PReLU kernels) against SMPLR_ENCODER_LAYOUT=channels_last (stock modules, MIOpen's NHWC
solvers without the transposes): ms per step and the encoder / decoder / optimizer split, same process, same inputs.
    python tools/train_layout_ab.py [--batch 128] [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilps_amd  # noqa: E402,F401
from ilps_amd.smpl_model import synthetic_smpl_model  # noqa: E402
from ilps_amd.training import SegTrainer  # noqa: E402


def run(layout, B, steps, model, dev):
    if layout:
        os.environ["SMPLR_ENCODER_LAYOUT"] = layout
    else:
        os.environ.pop("SMPLR_ENCODER_LAYOUT", None)
    torch.manual_seed(1234)
    tr = SegTrainer(model, output_wh=48, encoder_architecture="enet", use_IEF=True, device=dev, with_silhouette=True)
    tr.smpl_model.train()
    g = torch.Generator(device=dev).manual_seed(100)
    data = (torch.rand(B, 3, 256, 256, device=dev, generator=g), torch.randint(0, 32, (B, 48, 48), device=dev, generator=g),
            torch.randint(0, 2, (B, 48, 48), device=dev, generator=g))
    t0 = time.perf_counter()
    for _ in range(3):
        loss = tr.step(*data)
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(*data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    sp = [tr.step_timed(*data) for _ in range(5)]
    med = {k: float(np.median([d[k] for d in sp])) for k in ("encoder_ms", "decoder_ms", "optimizer_ms")}
    print("layout=%-14s B=%d  %.2f ms per step (%.0f images/s)  encoder %.2f  decoder %.3f  optimizer %.3f  loss %.5f  "
          "(first 3 steps %.1f s)" % (layout or "NCHW (default)", B, ms, B / ms * 1e3, med["encoder_ms"], med["decoder_ms"],
                                      med["optimizer_ms"], float(loss), t_warm), flush=True)
    del tr
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = synthetic_smpl_model(1234)
    for layout in (None, "channels_last", None, "channels_last"):
        run(layout, a.batch, a.steps, model, dev)


if __name__ == "__main__":
    main()
