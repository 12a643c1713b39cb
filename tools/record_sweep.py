"""Record-count sweep of the part rasteriser (VERDICT r04 #1d): B meshes of the headline's inputs with a FORCED mask of
exactly `nfar` far-reaching (weight-1) vertices each - the trick of tests/test_gpu_parity.py::test_seg_record_list_regimes
- through binning (mask given), the rasteriser and the segmentation backward.  Per (W, nfar): the rasteriser's own
launch time (events ON the launch), the three launches replayed from a graph, the passes a block makes over its LDS
table, and the time per record.  Also the reference's rand x 80 recipe (profiling_renderer.py:28).
    python tools/record_sweep.py [--batch 128] [--wh 48 64] [--nfar 400 600 800 1000 1500 3000]
Run it under SMPLR_LIB_PATH=<older build> for the curve before round 5's chunked table passes."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ilps_amd import _lib, ops  # noqa: E402
from ilps_amd.smpl_model import synthetic_smpl_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--wh", type=int, nargs="+", default=[48, 64])
    ap.add_argument("--nfar", type=int, nargs="+", default=[400, 600, 800, 1000, 1500, 3000])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
    pt = ops.get_part_table(1, dev, consts.V)
    lib, B = _lib.load(), a.batch
    st = torch.cuda.current_stream()
    off = pt.part_off.cpu().numpy()
    print("# build %s%s  B=%d" % (_lib.build_id()[:12], " (SMPLR_LIB_PATH)" if _lib.LIB_OVERRIDE else "", B))
    print("# W nfar | padded list mean/max | table records per tile | passes max, blocks multi-pass / all | raster us | "
          "bin + raster + seg_bwd us (graph) | raster ns per (mesh, record)")
    for W in a.wh:
        x = torch.tensor(bench.make_x(B, W, 1000), device=dev)
        coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
        proj = ops._skin_fwd(ops._blend_fwd(coef, consts, B), A, consts, cam=x)[1]
        dseg = torch.randn(B, W, W, 32, device=dev)
        prev = None
        cases = [("%d" % n, n) for n in a.nfar] + [("rand*80", -1)]
        for name, nfar in cases:
            rng = np.random.default_rng(W + max(nfar, 0))
            if nfar >= 0:
                pj = proj
                m = np.full((B, pt.VP), 500.0, np.float32)
                pos = pt.part_pos.cpu().numpy()
                for b in range(B):
                    m[b, rng.choice(pos, size=nfar, replace=False)] = 1.0
                mask = torch.tensor(m, device=dev)
            else:
                g = torch.Generator(device="cpu").manual_seed(5)
                pj = (torch.rand(B, 6890, 3, generator=g) * 80.0).to(dev)
                mask = ops.visibility(pj)
            # the padded list length from the mask (as seg_bin_kernel pads: each part to a multiple of 4)
            vis = (mask[:, pt.part_pos.long()] <= 208.0).cpu().numpy()
            cnt = np.stack([vis[:, off[p]:off[p + 1]].sum(1) for p in range(pt.P)], 1)
            far = ((cnt + 3) // 4 * 4).sum(1)
            ws, rec = ops._seg_bin(pj, mask, W, pt)
            seg, arg = ops._seg_raster(ws, rec, B, W, pt)
            tr, passes, multi, blocks = "-", "-", "-", "-"
            if hasattr(lib, "smplr_seg_raster_plan"):
                pl = ops.raster_plan(rec, W, pt)
                assert np.array_equal(pl["far_records"], far), "list lengths: header %s, mask %s" % (pl["far_records"][:4], far[:4])
                tr = "/".join(str(v) for v in sorted(set(pl["tile_records"].tolist())))
                passes, multi, blocks = pl["passes_max"], pl["blocks_multi_pass"], pl["blocks"]
            kms, kern = ctypes.c_float(0.0), []
            for i in range(25):
                ops._seg_bin(pj, mask, W, pt, rec=rec, ws=ws)
                _lib.check(lib.smplr_seg_raster_timed(B, W, pt.P, pt.K, _lib.ptr(ws), _lib.ptr(rec), _lib.ptr(seg),
                                                      _lib.ptr(arg), ctypes.byref(kms), _lib.stream()), "raster_timed")
                if i >= 5:
                    kern.append(float(kms.value))
            t_ras = float(np.median(kern)) * 1e3

            def three():
                ops._seg_bin(pj, mask, W, pt, rec=rec, ws=ws)
                ops._seg_raster(ws, rec, B, W, pt, out=(seg, arg))
                ops._seg_bwd(dseg, arg, rec, pt.VP, W, pt)
            three()
            t3 = bench.graph_time_ms(three, 10, st) * 1e3
            slope = ""
            if prev is not None and nfar >= 0 and far.mean() > prev[0]:
                slope = "  d(raster us)/d(records) x mean = %.2f" % ((t_ras - prev[1]) / (far.mean() - prev[0]) * far.mean() / t_ras)
            print("W=%d nfar=%s | %.0f / %d | %s | %s, %s / %s | %.1f | %.1f | %.3f%s"
                  % (W, name, far.mean(), far.max(), tr, passes, multi, blocks, t_ras, t3, t_ras * 1e3 / B / far.mean(), slope),
                  flush=True)
            if nfar >= 0:
                prev = (far.mean(), t_ras)


if __name__ == "__main__":
    main()
