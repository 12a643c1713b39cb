"""Shared reader of the -DSMPLR_TL timeline buffers (csrc/common.h)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ilps_amd import _lib  # noqa: E402


def read_stamps(name, nwg, nwaves):
    """(nwg, nwaves, 32) int64 of the timeline buffer of one kernel."""
    n = nwg * nwaves * 32
    host = (ctypes.c_uint32 * n)()
    lib = _lib.load()
    if not hasattr(lib, "smplr_tl_read_" + name):
        raise SystemExit("this library is not a timeline build: bash tools/build_tl.sh, then run under tools/ab_run.sh ... tl")
    rc = getattr(lib, "smplr_tl_read_" + name)(host, n)
    assert rc == 0, rc
    return np.frombuffer(host, dtype=np.uint32).reshape(nwg, nwaves, 32).astype(np.int64)


def placement(t):
    hw, xcc = t[:, 0, 29], t[:, 0, 30] & 0xF
    key = (xcc * 8 + ((hw >> 13) & 7)) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
    uniq, cnt = np.unique(key, return_counts=True)
    wall = t[:, 0, 28]
    rel = (wall - wall.min()) & 0xFFFFFFFF
    print("workgroups %d on %d CUs (per CU %d..%d); entry, 10 ns ticks after the first: p10/25/50/75/90/100 = %s"
          % (key.size, uniq.size, cnt.min(), cnt.max(), [int(np.percentile(rel, q)) for q in (10, 25, 50, 75, 90, 100)]))
