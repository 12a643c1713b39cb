"""Reads the in-kernel clock stamps of a timeline build of skin_bwd_kernel (skin.hip with the SMPLR_TL edits, lib_tlskin.so
copied over the library): per wave the shader clock at entry, before / after the first barrier (operands landed), after
the dv_posed store, after the second barrier (weights operand landed), after the MFMA loop, after the third barrier and
at exit; the 100 MHz wall clock at entry gives the launch's rounds of workgroups.  GPU only."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ilps_amd import _lib, ops  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402


def main():
    B, W = 128, 48
    dev = torch.device("cuda", 0)
    dec = SMPLDecoder(None, img_wh=W)
    x = torch.tensor(bench.make_x(B, W, 11), device=dev, requires_grad=True)
    for it in range(3):
        out = dec(x)
        loss = (out["seg"] ** 2).sum() + (out["verts"] ** 2).sum()
        loss.backward()
    torch.cuda.synchronize()
    lib = _lib.load()
    n = 3456 * 4 * 16
    host = (ctypes.c_uint32 * n)()
    f = lib.smplr_tl_read_skin if hasattr(lib, "smplr_tl_read_skin") else ctypes.CDLL(_lib.LIB_PATH).smplr_tl_read_skin
    rc = f(host, n)
    assert rc == 0, rc
    t = np.frombuffer(host, dtype=np.uint32).reshape(3456, 4, 16).astype(np.int64)
    d = (t - t[..., :1]) & 0xFFFFFFFF
    names = ["entry", "operands requested", "barrier 1 (landed)", "T + dv_posed + weights requested", "barrier 2", "mfma done",
             "barrier 3", "exit"]
    print("median / p90 over waves, clocks since the wave's entry:")
    for i in range(1, 8):
        v = d[..., i].reshape(-1)
        print("  %-34s %8.0f %8.0f" % (names[i], np.median(v), np.percentile(v, 90)))
    wall = t[:, 0, 8]
    w0 = wall.min()
    rel = (wall - w0) & 0xFFFFFFFF
    print("workgroup entry times, 100 MHz ticks after the first: percentiles 10/25/50/75/90/100:",
          [int(np.percentile(rel, q)) for q in (10, 25, 50, 75, 90, 100)])
    hw, xcc = t[:, 0, 9], t[:, 0, 10] & 0xF
    key = (xcc * 8 + ((hw >> 13) & 7)) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
    uniq, cnt = np.unique(key, return_counts=True)
    print("workgroups %d on %d CUs; per CU min %d max %d" % (key.size, uniq.size, cnt.min(), cnt.max()))


if __name__ == "__main__":
    main()
