"""The part rasteriser's launch forms give the same bits: the two-pixel kernel in both block shapes (128 pair-lanes x 8
part ranges, 64 x 10: `raster2_shape` picks by batch) against the one-pixel kernel of rounds 1-3, over the probe's
twelve (W, B, vertex sampling, mesh scale) cases, visibility path and explicit all-visible / none-visible masks.
The forms are chosen by environment variables the library reads once, so each runs in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hashes(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "seg_hash.py")], env=e, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("W=")]
    assert len(lines) == 12, r.stdout
    return lines


@pytest.mark.gpu
def test_raster_forms_are_bit_identical():
    ref = _hashes({"SMPLR_RASTER": "1"})
    for env in ({"SMPLR_RASTER_SHAPE": "1"}, {"SMPLR_RASTER_SHAPE": "2"}, {"SMPLR_RASTER_SHAPE": "3"}, {}):
        assert _hashes(env) == ref, env
