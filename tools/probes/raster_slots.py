"""Which hardware wave slots raster blocks get (timeline build): HW_ID of every wave of every block.  GPU only.
HW_ID (s_getreg 4): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timeline import read_stamps  # noqa: E402
import bench  # noqa: E402
from ilps_amd.decoder import SMPLDecoder  # noqa: E402

B, W = 128, 48
dev = torch.device("cuda", 0)
x = torch.tensor(bench.make_x(B, W, 11), device=dev)
dec = SMPLDecoder(None, img_wh=W)
for it in range(3):
    dec(x)
torch.cuda.synchronize()
t = read_stamps("raster", 1152, 16)
hw = t[..., 29]
wid, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xcc = t[..., 30] & 15
print("wave_id per block: min..max histogram of (min, max):")
mn, mx = wid.min(1), wid.max(1)
u, c = np.unique(np.stack([mn, mx], 1), axis=0, return_counts=True)
for (a, b), k in zip(u, c):
    print("  slots %d..%d: %d blocks" % (a, b, k))
key = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
entry = t[:, 0, 28]
order = np.argsort(entry)
first = {}
for b in order[:512]:
    first.setdefault(int(key[b]), []).append((int(b), int(mn[b]), int(entry[b] - entry.min())))
print("first-round blocks per CU (block id, min slot, entry tick): sample")
for k in list(first)[:8]:
    print("  CU", k, first[k])
par = [tuple(sorted(s[1] >= 4 for s in v)) for v in first.values() if len(v) == 2]
print("first-round pairs by (slot >= 4) of the two blocks:", {p: par.count(p) for p in set(par)})
print("block id difference of first-round pairs:", np.unique([abs(v[0][0] - v[1][0]) for v in first.values() if len(v) == 2], return_counts=True))
