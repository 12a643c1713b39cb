"""The rasteriser's part draw is a bare `ds_add_rtn_u32` in inline assembly (csrc/raster.hip, scan2_parts): the compiler
does not know that its destination register is still in flight.  This test compiles raster.hip to gfx950 assembly (no
GPU needed) and checks, for every draw of every raster2_fwd_kernel instantiation, that nothing reads or writes that
register before a full `s_waitcnt lgkmcnt(0)` - the condition the source relies on."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "indirect_learning_pose-shape_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_part_draw_result_is_not_touched_before_its_wait(tmp_path):
    out = tmp_path / "raster.s"
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                        "raster.hip", "-o", str(out)], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().split("\n")
    name, sites = None, 0
    for i, l in enumerate(lines):
        if l.startswith("_ZN5smplr") and ":" in l:
            name = l.split(":")[0]
        m = re.search(r"ds_add_rtn_u32 (v\d+), ", l)
        if not (m and name and "raster2_fwd_kernel" in name and "ASMSTART" in lines[i - 1]):
            continue
        sites += 1
        reg = m.group(1)
        for j in range(i + 1, min(len(lines), i + 5000)):
            lj = lines[j]
            if "s_waitcnt" in lj and "lgkmcnt(0)" in lj:
                break
            assert not (re.search(r"\b" + reg + r"\b", lj) and not lj.strip().startswith(";")), \
                "%s: %s used before the draw's wait: %s" % (name, reg, lj.strip())
        else:
            raise AssertionError("%s: no s_waitcnt lgkmcnt(0) after the draw" % name)
    assert sites >= 4, "no inline draw found (%d): has scan2_parts changed?" % sites
