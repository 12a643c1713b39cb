"""What the built code objects say about the kernels' registers, LDS and scratch (tools/kernel_resources.py reads the
metadata notes inside libsmplraster_hip.so - no GPU): the launch geometry the design counts on has to fit the hardware's
allotment rules, one of which the compiler's own occupancy remark does not know (above 80 SGPRs a
SIMD holds 7 waves, and `raster_fwd_kernel`'s two 16-wave blocks per CU became one: 46.6 us instead of 37.2)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import kernel_resources as kr  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    ks = kr.kernels()
    assert len(ks) > 30, "no kernel metadata found in the library"
    return ks


def _match(ks, pat):
    hit = {n: k for n, k in ks.items() if pat in n}
    assert hit, "no kernel named *%s*" % pat
    return hit


def test_raster_fwd_keeps_two_blocks_per_cu(kernels):
    for name, k in _match(kernels, "raster_fwd_kernel").items():
        assert k["sgpr"] <= 80, "%s: %d SGPRs - 7 waves per SIMD, one block per CU" % (name, k["sgpr"])
        assert kr.waves_per_simd(k) == 8, name
        assert 2 * k["lds"] <= 160 * 1024, "%s: %d B of LDS per block" % (name, k["lds"])
        assert k["scratch"] == 0, name


@pytest.mark.parametrize("pat,threads,blocks", [
    ("seg_bin_kernel", 1024, 1), ("silh_px_kernel", 1024, 1), ("seg_bwd_kernel", 768, 1), ("skin_bwd_kernelILb1E", 256, 7),
    ("pose_blend3_fwd_kernel", 512, 1), ("blend3_bwd_kernelILi3ELi1ELi2E", 256, 1), ("pose_bwd_kernel", 512, 1),
    # the GEMMs' small-batch forms count on TWO workgroups per CU (one's first round trip under the other's loop, and
    # two mesh tiles of a column slice sharing the constant's lines in L1)
    ("blend3_bwd_kernelILi2ELi2ELi1E", 256, 2), ("_ZN5smplr17blend3_fwd_kernel", 256, 2)])
def test_hot_path_kernels_fit_their_launch(kernels, pat, threads, blocks):
    """`blocks` workgroups of `threads` threads fit a CU's registers (4 SIMDs), without scratch."""
    for name, k in _match(kernels, pat).items():
        assert k["scratch"] == 0, "%s spills %d B per lane" % (name, k["scratch"])
        assert k["max_threads"] >= threads, name
        need = -(-blocks * (threads // 64) // 4)
        assert kr.waves_per_simd(k) >= need, "%s: %d waves per SIMD, the launch needs %d" % (name, kr.waves_per_simd(k), need)
