#!/usr/bin/env python3
"""bench.py - meshes/s of the hot path: SMPL(86) -> verts(6890x3) -> 48x48 31-part seg, fwd+bwd.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL) - and when it is NOT under a launcher (no WORLD_SIZE) it starts those N ranks itself as a fresh child
process and relays rank 0's line; `--gpus` and WORLD_SIZE disagreeing is an error.  A "step" is one forward + backward of the full decoder (BASELINE.json configs[2]: batch_smpl +
projection + compute_mask + projects_to_seg, B=128 meshes per GPU, W=48) on seeded synthetic
parameters that are already resident in HBM.  The path shards by mesh with no data-path collective
(SURVEY.md §8(e)): every rank processes its own B meshes ("weak" scaling), the timed region is
bracketed by barrier + synchronize and the MAX over ranks is taken.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      - the dominant kernel (raster_fwd_kernel alone, smplr_seg_raster), timed live with HIP events on
                  the launch stream; achieved = FLOPs of the pairs it evaluates (counted on the device) / mean launch
                  time, frac = achieved / fp32 vector peak (< 1); the brute-force count of SURVEY 8(d) is carried as
                  algorithmic_speedup; valu_* and traffic come from the committed PMC passes (profiles/)
  cpu_baseline  - the oracle's fp32 reference-shaped torch-CPU restatement timed on this box's host
                  cores on a bounded sample (rank 0, N=1 only); a reported baseline, not the target
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ilps_amd  # noqa: E402,F401
from ilps_amd import ops, _lib  # noqa: E402
from ilps_amd.smpl_model import synthetic_smpl_model, mean86, load_part_tables  # noqa: E402

# algorithmic work of the segmentation raster forward per mesh at W=48:
# 2304 px x 6879 part vertices = 15.85 M pair-evals x 7 FLOP + 71,424 (pixel,part) x (sqrt, exp).
SEG_FWD_FLOP_PER_MESH = 2304 * 6879 * 7 + 2304 * 31 * 2
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector peak == fp32 MFMA peak
HBM_PEAK_GBS = 8000.0


def make_x(B, W, seed):
    rng = np.random.default_rng(seed)
    x = np.tile(mean86(W), (B, 1))
    x[:, 0:2] += rng.normal(0.0, 1.0, (B, 2))
    x[:, 2:4] += rng.normal(0.0, 0.05 * W, (B, 2))
    x[:, 4:76] += rng.normal(0.0, 0.2, (B, 72))
    x[:, 76:86] += rng.normal(0.0, 1.0, (B, 10))
    return x.astype(np.float32)


def event_time_ms(fn, reps, stream):
    """Mean duration of fn() in ms, HIP events recorded on the stream fn launches on."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def graph_time_ms(fn, reps, stream):
    """Mean duration of fn() in ms with `reps` back-to-back calls replayed from one HIP graph (no host launch time
    in the figure; HIP events on the replay stream).  Falls back to event_time_ms if capture is refused."""
    try:
        s = torch.cuda.Stream()
        s.wait_stream(stream)
        with torch.cuda.stream(s):
            fn()
        stream.wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        for _ in range(max(2, 100 // reps)):             # (>= 100 untimed calls: a settled clock, as for the headline)
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = max(3, 60 // reps)
        e0.record(stream)
        for _ in range(nrep):
            g.replay()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / (nrep * reps)
    except Exception:
        torch.cuda.synchronize()
        return event_time_ms(fn, reps, stream)


def stage_breakdown(x, consts, pt, W, reps=20):
    """Per-C-ABI-call mean times (us) on the current stream: the calls of the timed step (pose_blend_fwd(fused), skin_fwd,
    vis_seg_fwd, seg_bwd, smpl_bwd) plus the two stand-alone calls the fused forward launch replaces (pose_fwd, blend_fwd)."""
    st = torch.cuda.current_stream()
    B = x.shape[0]
    VP = consts.V
    res = {}
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    res["pose_fwd"] = graph_time_ms(lambda: ops._pose_fwd(x, 4, consts), reps, st)
    v_posed = ops._blend_fwd(coef, consts, B)
    res["blend_fwd"] = graph_time_ms(lambda: ops._blend_fwd(coef, consts, B), reps, st)
    if consts.blend3_fwd is not None:       # what the step runs instead of the two above: one launch
        bufs = (Rs, J, A, Jt)
        res["pose_blend_fwd(fused)"] = graph_time_ms(lambda: ops._pose_blend_fwd(x, 4, consts, out=bufs, v_posed=v_posed), reps, st)
    verts, proj = ops._skin_fwd(v_posed, A, consts, cam=x)
    res["skin_fwd"] = graph_time_ms(lambda: ops._skin_fwd(v_posed, A, consts, cam=x), reps, st)
    vslot = torch.empty(B, VP, dtype=torch.int16, device=x.device)
    mask, seg, arg, rec = ops._vis_seg_fwd(proj, W, pt, vslot=vslot)
    res["vis_seg_fwd"] = graph_time_ms(lambda: ops._vis_seg_fwd(proj, W, pt, vslot=vslot), reps, st)
    dseg = torch.randn_like(seg)
    part, nsplit = ops._seg_bwd(dseg, arg, rec, VP, W, pt, merge=False)
    res["seg_bwd"] = graph_time_ms(lambda: ops._seg_bwd(dseg, arg, rec, VP, W, pt, merge=False), reps, st)
    res["smpl_bwd(skin+blend+pose)"] = graph_time_ms(
        lambda: ops._smpl_bwd(x, 4, consts, Rs, J, A, v_posed, None, None, None, seg_grad=(part, vslot, nsplit)),
        reps, st)
    return {k: round(v * 1e3, 2) for k, v in res.items()}


def stage_rooflines(stages, B, W, V, gemm):
    """Each stage against the roof that bounds it, from ALGORITHMIC work per mesh
    and the live time of the call replayed from a HIP graph (HIP events; includes the ~2 us kernel-to-kernel gap;
    rocprofv3 kernel times are in profiles/).  Bytes are compulsory traffic: inputs read once + outputs written once + constants once per batch."""
    npx = W * W
    n3 = 3 * V
    const_fwd = (n3 * 224 * 6) if gemm == "bf16x3" else (n3 * 220 * 4)     # packed bf16x3 constant: 6 B per entry
    rows = [
        # name, bound, work per launch, unit scale
        ("blend_fwd", "mfma", 2.0 * 220 * n3 * B, None),
        ("blend_fwd", "hbm", const_fwd + n3 * 4 * B, None),
        ("skin_fwd", "hbm", (3 * n3 * 4 + 288 * 4) * B + V * 32, None),            # v_posed in, verts + proj out
        ("seg_bwd", "hbm", npx * (32 * 4 + 32 * 2) * B, None),                     # dseg + arg in (slot sums are small)
    ]
    out = []
    for name, bound, work, _ in rows:
        t = stages[name] * 1e-6
        if bound == "mfma":
            ach, peak, unit = work / t / 1e12, FP32_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work / t / 1e9, HBM_PEAK_GBS, "GB/s"
        out.append({"stage": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "us": stages[name]})
    return out


def plan_keys(rec, W, pt):
    """The rasteriser's record-list regime for a binned batch (ops.raster_plan: list headers + smplr_seg_raster_plan):
    how long the far-reaching lists are against what one pass over a block's LDS table takes, and how many blocks
    needed more than one pass (round 5: chunked table passes; rounds 1-4: a scalar-load walk of the whole list)."""
    pl = ops.raster_plan(rec, W, pt)
    far = pl["far_records"]
    edges = [0, 400, 600, 800, 1000, 1500, 2000, 3000, 1 << 30]
    hist = np.histogram(far, bins=edges)[0]
    return {"far_records_padded_mean": round(float(far.mean()), 1), "far_records_padded_max": int(far.max()),
            "far_records_hist": {("%d-%d" % (a, b - 1) if b < (1 << 30) else ">=%d" % a): int(h)
                                 for a, b, h in zip(edges[:-1], edges[1:], hist) if h},
            "tile_records": [int(v) for v in sorted(set(pl["tile_records"].tolist()))],
            "blocks": pl["blocks"], "blocks_multi_pass": pl["blocks_multi_pass"],
            "blocks_scalar_walk": pl["blocks_scalar_walk"], "passes_max": pl["passes_max"],
            "block_shape": "%d pair-lanes x %d part ranges" % (pl["pair_lanes"], pl["part_ranges"])}


def raster_recipe_leg(B, W, vs, dev, consts, cpu_ms=False):
    """The reference's own rasteriser workload (profiling_renderer.py:19-39) at batch B: projects = rand(B, 6890, 3) x 80,
    compute_mask over all 6890 of them, projects_to_seg at W x W with `vertex_sampling` (which, there, gathers positions
    ids // vs of the 6890-long list) - forward, and backward of a random cotangent (the reference only predicts).
    Uniform vertices put a winner in most cells of the 64 x 64 visibility grid: ~2 700 far-reaching records per mesh
    at vertex_sampling = None, several passes over the rasteriser's LDS table."""
    import ctypes
    pt = ops.get_part_table(vs, dev, consts.V)
    g = torch.Generator(device="cpu").manual_seed(5 + B)
    pr = (torch.rand(B, 6890, 3, generator=g) * 80.0).to(dev)
    st = torch.cuda.current_stream()
    dseg = torch.randn(B, W, W, 32, device=dev)
    prv = pr[:, :pt.VP].contiguous()

    def fwd():
        if vs in (None, 1):
            mk, seg, arg, rec = ops._vis_seg_fwd(pr, W, pt)
        else:
            mk = ops.visibility(pr)
            seg, arg, rec = ops._seg_fwd(prv, mk[:, :pt.VP].contiguous(), W, pt)[:3]
        return mk, seg, arg, rec

    def fwd_bwd():
        mk, seg, arg, rec = fwd()
        return ops._seg_bwd(dseg, arg, rec, pt.VP, W, pt)
    mk, seg, arg, rec = fwd()
    fwd_bwd()
    t_f = graph_time_ms(fwd, 10, st)
    t_fb = graph_time_ms(fwd_bwd, 10, st)
    mkv = mk[:, :pt.VP].contiguous()
    ws, rec2 = ops._seg_bin(prv, mkv, W, pt)
    lib_, kms, kern = _lib.load(), ctypes.c_float(0.0), []
    for i in range(25):
        ops._seg_bin(prv, mkv, W, pt, rec=rec2, ws=ws)
        _lib.check(lib_.smplr_seg_raster_timed(B, W, pt.P, pt.K, _lib.ptr(ws), _lib.ptr(rec2), _lib.ptr(seg), _lib.ptr(arg),
                                               ctypes.byref(kms), _lib.stream()), "smplr_seg_raster_timed")
        if i >= 5:
            kern.append(float(kms.value))
    t_ras = float(np.median(kern)) * 1e-3
    far = (mkv[:, pt.part_pos.long()] <= 208.0).sum(dim=1).double()
    pairs = float(far.sum().item()) * W * W
    out = {"batch": B, "img_wh": W, "vertex_sampling": vs, "fwd_ms": round(t_f, 4), "fwd_bwd_ms": round(t_fb, 4),
           "meshes_per_s_fwd_bwd": round(B / (t_fb * 1e-3), 1), "raster_us": round(t_ras * 1e6, 2),
           "far_records_per_mesh": round(float(far.mean().item()), 1), "far_records_max": int(far.max().item()),
           "raster_tflops": round(pairs * 7.0 / t_ras / 1e12, 3),
           "raster_frac": round(pairs * 7.0 / t_ras / 1e12 / FP32_PEAK_TFLOPS, 4), "executed_pairs_per_launch": int(pairs)}
    out.update(plan_keys(rec2, W, pt))
    return out


def raster_roofline(x, consts, pt, W, stages):
    """`roofline` of the dominant kernel, raster_fwd_kernel (the pair loop of projects_to_seg.py:41-56), against
    the fp32 VECTOR peak - the loop runs on the VALU, not on the matrix cores; the schema's `bound` offers hbm|mfma
    and the two fp32 peaks are equal (157.3 TFLOP/s), so `bound` says "mfma" and `pipe` says what it is.
      achieved = pairs the kernel EVALUATES (the mesh's far-reaching records, counted on the device from this run's
                 masks, x W*W pixels x B) x 7 FLOP (2 sub, mul, fma, weight, min) / launch time
      frac     = achieved / 157.3: a true fraction (< 1)
      algorithmic_* = the brute-force count SURVEY 8(d) prices (W*W x 6879 pairs per mesh) over the same time: how
                 much arithmetic the reach split makes unnecessary - a speed-up factor, not a utilisation
    Launch time: HIP events on the launch stream around the kernel's launch (smplr_seg_raster) inside an eager step of
    the step's own kernels (includes the event records' dispatch gap; the rocprofv3 kernel average is in profiles/).  valu_* come from the
    committed SQ counter pass of the same kernel (profiles/raster_sq.json), traffic from the FETCH/WRITE passes."""
    B, V = x.shape[0], consts.V
    st = torch.cuda.current_stream()
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    proj = ops._skin_fwd(ops._blend_fwd(coef, consts, B), A, consts, cam=x)[1]
    mask = torch.empty(B, pt.VP, device=x.device)
    ws, rec = ops._seg_bin(proj, mask, W, pt, grid_wh=64)
    seg, arg = ops._seg_raster(ws, rec, B, W, pt)
    t_bin = graph_time_ms(lambda: ops._seg_bin(proj, mask, W, pt, grid_wh=64, rec=rec, ws=ws), 20, st) * 1e-3
    t_iso = graph_time_ms(lambda: ops._seg_raster(ws, rec, B, W, pt, out=(seg, arg)), 20, st) * 1e-3
    # the kernel's launch time INSIDE the step it belongs to: an eager step made of the step's own kernels (pose + blend,
    # skinning, binning, [raster], seg_bwd, smpl_bwd) with HIP events ON the raster launch (smplr_seg_raster_timed:
    # hipExtLaunchKernel's start / stop events = the kernel begin to end, what the rocprofv3 trace of `bench.py --mode
    # eager` averages, profiles/).  An event PAIR recorded around the launch reads 3-4 us more (dispatch gaps)
    dseg_r = torch.randn_like(seg)
    vslot_r = torch.empty(B, pt.VP, dtype=torch.int16, device=x.device)
    nrep = 60
    import ctypes
    lib_, check_ = _lib.load(), _lib.check
    kms, kern_ms = ctypes.c_float(0.0), []
    for i in range(nrep + 10):
        Rs_, J_, A_, Jt_, vp_ = ops._pose_blend_fwd(x, 4, consts) if consts.blend3_fwd is not None else (None,) * 5
        if vp_ is None:
            coef_, Rs_, J_, A_, Jt_ = ops._pose_fwd(x, 4, consts)
            vp_ = ops._blend_fwd(coef_, consts, B)
        pj_ = ops._skin_fwd(vp_, A_, consts, cam=x)[1]
        ops._seg_bin(pj_, mask, W, pt, grid_wh=64, rec=rec, ws=ws, vslot=vslot_r)
        if i >= 10:
            check_(lib_.smplr_seg_raster_timed(B, W, pt.P, pt.K, _lib.ptr(ws), _lib.ptr(rec), _lib.ptr(seg), _lib.ptr(arg),
                                               ctypes.byref(kms), _lib.stream()), "smplr_seg_raster_timed")
            kern_ms.append(float(kms.value))
        else:
            ops._seg_raster(ws, rec, B, W, pt, out=(seg, arg))
        part_, ns_ = ops._seg_bwd(dseg_r, arg, rec, pt.VP, W, pt, merge=False)
        ops._smpl_bwd(x, 4, consts, Rs_, J_, A_, vp_, None, None, None, seg_grad=(part_, vslot_r, ns_))
    torch.cuda.synchronize()
    t_ras = float(np.median(kern_ms)) * 1e-3                                  # the kernel itself, begin to end
    # far-reaching records per mesh, counted on the device: part-table vertices whose mask is <= 208 (here: == 1)
    far = (mask[:, pt.part_pos.long()] <= 208.0).sum(dim=1).double()
    n_far = float(far.mean().item())
    pairs = float(far.sum().item()) * W * W                          # evaluated pairs per launch (pads excluded)
    flop = pairs * 7.0
    ach = flop / t_ras / 1e12
    brute = float(W * W) * pt.K * B                                  # every pixel x every part vertex
    out = {"kernel": "raster2_fwd_kernel (smplr_seg_raster; two pixels per lane since round 4)", "bound": "mfma", "pipe": "fp32 VALU (vector peak = fp32 matrix peak)",
           "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_PEAK_TFLOPS, 4),
           "traffic": None, "launch_us": round(t_ras * 1e6, 2), "launch_us_isolated_replay": round(t_iso * 1e6, 2),
           "launch_us_how": "median over 60 launches inside an eager step of the step's own seven kernels, HIP events ON the "
                            "launch (hipExtLaunchKernel start / stop: the kernel begin to end, as rocprofv3 reports it); "
                            "isolated_replay = 20 back-to-back copies replayed from one HIP graph, event pair around them "
                            "(adds the dispatch gaps)",
           "bin_launch_us": round(t_bin * 1e6, 2),
           "far_records_per_mesh": round(n_far, 1), "far_records_max": int(far.max().item()),
           "executed_pairs_per_launch": int(pairs), "flop_per_pair": 7,
           "algorithmic_pairs_per_launch": int(brute), "algorithmic_speedup": round(brute / pairs, 2),
           "algorithmic_equiv_tflops": round((brute * 7.0 + W * W * pt.P * 2.0 * B) / t_ras / 1e12, 2),
           "note": "achieved counts only evaluated pairs; the brute-force figure of SURVEY 8(d) is reported as "
                   "algorithmic_speedup / algorithmic_equiv_tflops, never as a utilisation"}
    out.update(plan_keys(rec, W, pt))
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf) and B == 128 and W == 48:
        try:
            tj = json.load(open(tf))
            k = next((v for kn, v in tj["kernels"].items() if kn.startswith(("raster2_fwd_kernel", "raster_fwd_kernel"))), {})
            # the committed counter passes are stamped with the library they were taken on
            out["profile_build_id_matches"] = tj.get("build_id") == _lib.build_id()
            out["traffic"] = k.get("hbm_bytes_per_launch")
            out["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 read-side x2)"
        except Exception:
            pass
    sq = os.path.join(ROOT, "profiles", "raster_sq.json")
    if os.path.exists(sq) and B == 128 and W == 48:
        try:
            q = json.load(open(sq))
            cyc = q["kernel_us"] * 1e-6 * q["clock_hz"] * q["simds"]      # SIMD-cycles the kernel had
            out["valu_active_frac"] = round(4.0 * q["SQ_ACTIVE_INST_VALU"] / cyc, 3)
            out["valu_issue_frac"] = [round(q["issue_cycles_plain"] * q["SQ_INSTS_VALU"] / cyc, 3),
                                      round(q["issue_cycles_packed"] * q["SQ_INSTS_VALU"] / cyc, 3)]
            out["valu_source"] = ("profiles/raster_sq.json: valu_active_frac = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) and "
                                  "valu_issue_frac = SQ_INSTS_VALU x [%.2f, %.2f] cycles per wave64 instruction (measured issue "
                                  "cost of a plain and of a packed / three-operand fp32 instruction at 8 waves per SIMD, %s; "
                                  "the kernel mixes both), each over %d SIMDs x %.2f us x %.1f GHz nominal"
                                  % (q["issue_cycles_plain"], q["issue_cycles_packed"], "profiles/r04_valu_issue_probe.txt",
                                     q["simds"], q["kernel_us"], q["clock_hz"] / 1e9))
        except Exception:
            pass
    return out


def size_leg(x, consts, W, vs, dev):
    """One of the reference's OTHER shipped sizes through the headline's step (decoder fwd+bwd at the batch of `x`,
    ten steps per HIP-graph replay, HIP events): W = 64 is what train.py:320-321 and predict.py:129-136 call,
    vertex_sampling 2 / 5 what projects_to_seg.py:18-24 and profiling_renderer.py:26-28 offer.  Adds the rasteriser's
    own launch time (hipExtLaunchKernel events ON the launch, median of 30, inside bin -> raster pairs), the
    far-reaching records per mesh it walks and its fraction of the fp32 vector peak, as `roofline` does for the headline."""
    import ctypes
    B = x.shape[0]
    pt = ops.get_part_table(vs, dev, consts.V)
    st = torch.cuda.current_stream()
    dseg = torch.randn(B, W, W, 32, device=dev)

    def step():
        xg = x.detach().requires_grad_(True)
        o_ = ops.DecoderFn.apply(xg, consts, 4, W, vs, pt, 64, True, False, 1)
        o_[3].backward(dseg)
    step()
    ms = graph_time_ms(step, 10, st)
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    proj = ops._skin_fwd(ops._blend_fwd(coef, consts, B), A, consts, cam=x, vertex_sampling=vs)[1]
    mask = torch.empty(B, pt.VP, device=dev)
    ws, rec = ops._seg_bin(proj, mask, W, pt, grid_wh=64)
    seg, arg = ops._seg_raster(ws, rec, B, W, pt)
    lib_, kms, kern = _lib.load(), ctypes.c_float(0.0), []
    for i in range(35):
        ops._seg_bin(proj, mask, W, pt, grid_wh=64, rec=rec, ws=ws)
        _lib.check(lib_.smplr_seg_raster_timed(B, W, pt.P, pt.K, _lib.ptr(ws), _lib.ptr(rec), _lib.ptr(seg), _lib.ptr(arg),
                                               ctypes.byref(kms), _lib.stream()), "smplr_seg_raster_timed")
        if i >= 5:
            kern.append(float(kms.value))
    t_ras = float(np.median(kern)) * 1e-3
    far = (mask[:, pt.part_pos.long()] <= 208.0).sum(dim=1).double()
    pairs = float(far.sum().item()) * W * W
    ach = pairs * 7.0 / t_ras / 1e12
    out = {"img_wh": W, "vertex_sampling": vs, "meshes": B, "ms_per_step": round(ms, 4),
           "meshes_per_s": round(B / (ms * 1e-3), 1), "far_records_per_mesh": round(float(far.mean().item()), 1),
           "far_records_max": int(far.max().item()), "part_vertices": int(pt.K),
           "raster_us": round(t_ras * 1e6, 2), "raster_tflops": round(ach, 3), "raster_frac": round(ach / FP32_PEAK_TFLOPS, 4),
           "executed_pairs_per_launch": int(pairs), "algorithmic_pairs_per_launch": int(float(W * W) * pt.K * B)}
    out.update(plan_keys(rec, W, pt))
    return out


def _cpu_info():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(avail, 16))          # the GPU box's CPU share for one GPU is 16 cores
    cpu_model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return ncores, cpu_model


def _timed(one, budget_s, max_n=2000):
    one()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_n:
            return n, el


def cpu_baseline(model, W, budget_s=14.0):
    """The oracle's fp32 torch-CPU restatement timed on this box's host cores (kind "port"), three legs on bounded
    samples:
      value      dense: the reference's formulation with materialised (N, W^2, n_p, 2) tiles under autograd, full decoder
                 fwd+bwd at B = 1 (what fits: ~0.6 GB of intermediates per mesh); visibility by the vectorised sort
      streaming  the same decoder fwd+bwd at B = 128 (the GPU workload's batch) without the materialised tiles or a tape
                 for the rasteriser (per-part distance blocks, min over the part, hand-written gradient): separates the
                 algorithmic reformulation from the hardware
      smpl_only  batch_smpl fwd+bwd at B = 256 (BASELINE configs[1])"""
    from oracle import torch_oracle as to
    from oracle import np_oracle as no
    ncores, cpu_model = _cpu_info()
    torch.set_num_threads(ncores)
    ids, off = load_part_tables(1)
    smpl = to.TorchSMPL(model, dtype=torch.float32)
    mask_fn = lambda p: torch.tensor(no.compute_mask_sorted(p.numpy().astype(np.float64)), dtype=torch.float32)
    # ---- dense, B = 1
    x = torch.tensor(make_x(1, W, 123), requires_grad=True)
    g = torch.randn(1, W, W, 32)

    def dense():
        x.grad = None
        _v, _p, _m, seg = to.decoder_forward(smpl, x, mask_fn, W, ids, off)
        (seg * g).sum().backward()
    n, el = _timed(dense, budget_s)
    out = {"value": round(n / el, 3), "unit": "meshes/s", "cores": int(torch.get_num_threads()), "cpu": cpu_model,
           "os_cpu_count": os.cpu_count(), "kind": "port",
           "sample": "%d fwd+bwd passes of the full decoder at B=1, W=%d (%.1f s): fp32 torch-CPU restatement of the "
                     "reference's dense formulation (oracle/torch_oracle.py; TensorFlow itself is not installable "
                     "here), visibility by one sort per mesh (oracle/np_oracle.compute_mask_sorted)" % (n, W, el)}
    # ---- streaming, B = 128
    try:
        Bs = 128
        xs = torch.tensor(make_x(Bs, W, 321), requires_grad=True)
        gs = torch.randn(Bs, W, W, 32)

        def streaming():
            xs.grad = None
            verts = smpl(xs)
            proj = to.orthographic_project(verts, xs)
            with torch.no_grad():
                pd = proj.detach()
                _seg, dproj = to.seg_streaming_fwd_bwd(pd, mask_fn(pd), gs, W, ids, off)
            proj.backward(dproj)
        n2, el2 = _timed(streaming, budget_s * 0.8, 50)
        out["streaming"] = {"value": round(Bs * n2 / el2, 2), "unit": "meshes/s", "batch": Bs, "passes": n2,
                            "seconds": round(el2, 1),
                            "what": "non-materialising CPU variant (oracle/torch_oracle.seg_streaming_fwd_bwd + "
                                    "autograd through the SMPL layer), same threads"}
    except Exception as e:
        out["streaming"] = {"error": str(e)}
    # ---- batch_smpl only, B = 256
    try:
        x2 = torch.tensor(make_x(256, W, 7), requires_grad=True)
        gv = torch.randn(256, model.v_template.shape[0], 3)

        def smpl_only():
            x2.grad = None
            (smpl(x2) * gv).sum().backward()
        n3, el3 = _timed(smpl_only, budget_s * 0.4, 200)
        out["smpl_only_B256"] = {"value": round(256 * n3 / el3, 1), "unit": "meshes/s", "passes": n3,
                                 "seconds": round(el3, 1)}
    except Exception as e:
        out["smpl_only_B256"] = {"error": str(e)}
    return out


def parity_sample(model, consts, pt, W, dev, Bp=2):
    """The other half of BASELINE's metric ("vert Linf vs ref"): the HIP path against the float64 oracle on Bp
    seeded meshes - vertex L-inf, and the largest |seg - ref| / (1e-3 |ref| + 1e-6) (<= 1 passes the bar)."""
    from oracle import np_oracle as no
    xs = make_x(Bp, W, 4242)
    x = torch.tensor(xs, device=dev)
    verts, proj, mask, seg, _silh, _jt, _ls = ops.DecoderFn.apply(x, consts, 4, W, 1, pt, 64, True, False, 1)
    ref_v = no.smpl_layer_call(xs.astype(np.float64), model)
    pj = proj.cpu().numpy().astype(np.float64)
    ref_m = no.compute_mask(pj)
    ids, off = load_part_tables(1)
    ref_s = no.projects_to_seg(pj, ref_m, W, ids, off)
    got_s = seg.cpu().numpy()
    return {"meshes": Bp, "vert_linf": float(np.abs(verts.cpu().numpy() - ref_v).max()), "vert_bar": 1e-4,
            "mask_equal": bool(np.array_equal(ref_m, mask.cpu().numpy())),
            "seg_err_over_bar": float((np.abs(got_s - ref_s) / (1e-3 * np.abs(ref_s) + 1e-6)).max()),
            "seg_bar": "|d| <= 1e-3 |ref| + 1e-6", "oracle": "oracle/np_oracle.py (float64)"}


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a FRESH child process
    (`python -m torch.distributed.run`, one rank per GPU) before this process has touched the GPU, relay rank 0's
    JSON line and exit with the child's code.  Never exec: replacing a process that has initialised HIP takes the
    machine down on this pool, and a child keeps the parent free of any GPU state."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    out = proc.stdout.decode("utf-8", "replace")
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        sys.stderr.write("bench: the %d-rank child printed no JSON line\n" % n)
        return 1
    return proc.returncode


def _host_placement():
    """Where this rank runs on the host: the devices it may see, the cores it may run on and their NUMA nodes - eight
    graph-replaying ranks on a 16-core cgroup is the first thing that bends a scaling curve."""
    out = {"HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"),
           "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"), "pid": os.getpid()}
    try:
        cpus = sorted(os.sched_getaffinity(0))
        out["cpu_affinity"] = len(cpus)
        out["cpu_first_last"] = [cpus[0], cpus[-1]] if cpus else None
        nodes = set()
        for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
            try:
                ids = set()
                for part in open(os.path.join(nd, "cpulist")).read().strip().split(","):
                    if part:
                        a, _, b = part.partition("-")
                        ids.update(range(int(a), int(b or a) + 1))
                if ids & set(cpus):
                    nodes.add(int(os.path.basename(nd)[4:]))
            except (OSError, ValueError):
                pass
        out["numa_nodes"] = sorted(nodes) if nodes else None
    except (AttributeError, OSError):
        out["cpu_affinity"] = None
    return out


def _ranks_seen(dist, rank, world, dev, backend, own_ms):
    """Every rank's (rank, device index, device name, its OWN ms per step before the MAX reduction), gathered on all
    ranks: the N > 1 line shows that N distinct devices took part (`rccl_ranks_seen`) and how far apart they ran."""
    if dev is not None and getattr(dev, "type", "cpu") == "cuda":
        mine = {"rank": rank, "device": int(torch.cuda.current_device()), "name": torch.cuda.get_device_name(dev),
                "backend": backend if world > 1 else None, "ms_per_step": round(own_ms, 4)}
    else:
        mine = {"rank": rank, "device": "cpu", "name": "cpu (pid %d)" % os.getpid(),
                "backend": backend if world > 1 else None, "ms_per_step": round(own_ms, 4)}
    mine["host"] = _host_placement()
    if dist is None or world == 1:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return sorted(got, key=lambda d: d["rank"])


def _max_over_ranks(dist, v, dev, backend):
    if dist is None:
        return v
    tt = torch.tensor([v], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def _timed_steps(fn, warm, steps, dist, dev, backend, sync):
    """`steps` calls of fn bracketed by barrier + synchronize on both sides, MAX over ranks -> seconds."""
    for _ in range(warm):
        fn()
    sync()
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist:
        dist.barrier()
    sync()
    return _max_over_ranks(dist, time.perf_counter() - t0, dev, backend)


def train_leg(args, rank, world, dev, dist, backend, model, W, per=None):
    """The data-parallel TRAIN step of BASELINE configs[3]/[4] (train.py:205-215,
    train_stage2_silhouette.py:226-234): ENet(256x256x3) + IEF + HIP decoder with both heads + softmax-focal loss +
    silhouette cross-entropy + Adam, `--train-batch` images per GPU, DistributedDataParallel over RCCL when world > 1
    - the design's ONE collective (the gradient all-reduce of the encoder + regressor: `allreduce_MiB` per step).
    Every rank runs it (rank-synchronous); rank 0 gets the dict.  Legs, each barrier-bracketed and MAX-reduced:
      ddp                 the step as trained (all-reduce overlapped with backward by DDP's buckets)
      no_sync             the same step inside `no_sync()`: no all-reduce - the difference is what the collective costs
      strong_global_G     the step at G / world images per GPU (G = --train-global-batch, 1024: the strong-scaling point)"""
    import contextlib
    from ilps_amd.training import SegTrainer
    per = int(args.train_batch) if per is None else int(per)
    steps, warm = int(args.train_steps), 3
    sync = torch.cuda.synchronize
    torch.manual_seed(1234)                                  # same initial weights on every rank (DDP broadcasts anyway)
    tr = SegTrainer(model, output_wh=W, encoder_architecture="enet", use_IEF=True, device=dev, ddp=world > 1,
                    with_silhouette=True)
    tr.smpl_model.train()
    nparam = sum(p.numel() for p in tr.smpl_model.parameters() if p.requires_grad)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)

    def batch(n):
        return (torch.rand(n, 3, 256, 256, device=dev, generator=gen),
                torch.randint(0, 32, (n, W, W), device=dev, generator=gen),
                torch.randint(0, 2, (n, W, W), device=dev, generator=gen))

    out = {"workload": "train step: ENet(256x256x3) + IEF + decoder(W=%d, seg + silhouette) + softmax-focal loss + "
                       "silhouette CE + Adam; BASELINE configs[3]/[4]" % W,
           "images_per_gpu": per, "global_batch": per * world, "n_gpus": world, "steps": steps, "warmup": warm,
           "ddp": world > 1, "backend": (backend if world > 1 else None),
           "params": int(nparam), "allreduce_MiB": round(nparam * 4 / 2 ** 20, 1) if world > 1 else 0.0,
           "bucket_MiB": 25, "scaling": "weak"}
    data = batch(per)
    el = _timed_steps(lambda: tr.step(*data), warm, steps, dist, dev, backend, sync)
    out["ddp_step" if world > 1 else "step"] = {"ms_per_step": round(el / steps * 1e3, 3),
                                                "images_per_s": round(world * per * steps / el, 1)}
    out["images_per_s"] = round(world * per * steps / el, 1)
    out["ms_per_step"] = round(el / steps * 1e3, 3)
    # where the step's time goes: HIP events between its sections (median of 5 steps; every rank runs them, rank 0 reports)
    try:
        sp = [tr.step_timed(*data) for _ in range(5)]
        for k in ("encoder_ms", "decoder_ms", "optimizer_ms"):
            out[k] = round(float(np.median([d[k] for d in sp])), 3)
        out["split_note"] = ("encoder = ENet + IEF forward and backward (stock torch / MIOpen; under DDP with the all-reduce "
                             "it overlaps), decoder = HIP decoder + losses forward and backward, optimizer = zero_grad + Adam")
        out["conv_backend"] = {"cudnn_benchmark": bool(torch.backends.cudnn.benchmark),
                               "MIOPEN_FIND_MODE": os.environ.get("MIOPEN_FIND_MODE")}
    except Exception as e:
        out["split_error"] = "%s: %s" % (type(e).__name__, e)
    if world > 1:
        def nosync_step():
            with tr.net.no_sync():
                tr.step(*data)
        el2 = _timed_steps(nosync_step, 1, steps, dist, dev, backend, sync)
        out["no_sync_step"] = {"ms_per_step": round(el2 / steps * 1e3, 3),
                               "images_per_s": round(world * per * steps / el2, 1),
                               "note": "same step without the gradient all-reduce (DDP.no_sync)"}
        out["allreduce_exposed_ms"] = round((el - el2) / steps * 1e3, 3)
        G = int(args.train_global_batch)
        if G % world == 0 and G // world <= 512 and G // world != per:
            try:
                ps = G // world
                del data
                data_s = batch(ps)
                el3 = _timed_steps(lambda: tr.step(*data_s), 2, steps, dist, dev, backend, sync)
                out["strong_global_%d" % G] = {"images_per_gpu": ps, "ms_per_step": round(el3 / steps * 1e3, 3),
                                               "images_per_s": round(G * steps / el3, 1), "scaling": "strong"}
                del data_s
            except Exception as e:          # (an out-of-memory on one rank would desynchronise the others: reported, not hidden)
                out["strong_global_%d" % G] = {"error": str(e)}
        elif G // world == per:
            out["strong_global_%d" % G] = {"images_per_gpu": per, "same_as": "ddp_step", "scaling": "strong"}
    del tr
    torch.cuda.empty_cache()
    return out


def dry_run(args, rank, world):
    """`--dry-run`: the N-rank protocol of the bench without a GPU (gloo): per-rank seeded inputs, a CPU stand-in
    step (parameter conditioning of the rank's own meshes + a rank-dependent sleep), barrier-bracketed timing,
    MAX over ranks, ONE JSON line from rank 0.  What tests/test_bench_launcher.py checks at world size 2."""
    import torch.distributed as dist
    from ilps_amd.keras_smpl.set_cam_params import load_mean_set_cam_params
    if world > 1:
        dist.init_process_group("gloo")
    B, W = args.batch, args.wh
    seed = 1000 + rank
    x = torch.tensor(make_x(B, W, seed))
    # fault rehearsal: BENCH_DRY_FAIL_RANK=r makes rank r die after the rendezvous - the launcher must then end the
    # other ranks and the fresh child this was started as must exit non-zero, with no JSON line (tests/test_bench_launcher.py)
    if os.environ.get("BENCH_DRY_FAIL_RANK", "") == str(rank):
        sys.stderr.write("bench: rank %d fails on purpose (BENCH_DRY_FAIL_RANK)\n" % rank)
        os._exit(3)

    def step():
        load_mean_set_cam_params(x, W)
        time.sleep(0.002 * (rank + 1))

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    work = time.perf_counter() - t0                    # this rank's own steps, before it waits for the others
    if world > 1:
        dist.barrier()
    mine = time.perf_counter() - t0
    elapsed, seeds, times, sums, works = mine, [seed], [mine], [float(x.double().sum())], [work]
    if world > 1:
        tt = torch.tensor([mine], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rows = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(rows, torch.tensor([seed, mine, sums[0], work], dtype=torch.float64))
        seeds, times, sums, works = ([float(r[i]) for r in rows] for i in range(4))
    # the train leg's protocol on CPU: a stand-in regressor (the real one's IEF head on random features) under
    # DistributedDataParallel over gloo - ddp step, no_sync step, all-reduce volume - same keys as the GPU leg
    tleg = None
    if not args.no_train_leg:
        import contextlib
        from torch import nn
        torch.manual_seed(7)
        net = nn.Sequential(nn.Linear(64, 128), nn.ReLU(), nn.Linear(128, 86))
        ddp = nn.parallel.DistributedDataParallel(net) if world > 1 else net
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        feats = torch.randn(int(args.train_batch), 64, generator=torch.Generator().manual_seed(100 + rank))

        def tstep(ctx=contextlib.nullcontext):
            with ctx():
                opt.zero_grad(set_to_none=True)
                ddp(feats).square().mean().backward()
                opt.step()
        dmod = dist if world > 1 else None
        el1 = _timed_steps(tstep, 1, args.train_steps, dmod, "cpu", "gloo", lambda: None)
        nparam = sum(p.numel() for p in net.parameters())
        tleg = {"workload": "DRY RUN stand-in (CPU MLP under DDP/gloo): protocol only", "stand_in": True,
                "images_per_gpu": int(args.train_batch), "global_batch": int(args.train_batch) * world, "n_gpus": world,
                "steps": args.train_steps, "ddp": world > 1, "backend": "gloo" if world > 1 else None,
                "params": nparam, "allreduce_MiB": round(nparam * 4 / 2 ** 20, 4) if world > 1 else 0.0, "scaling": "weak",
                "images_per_s": round(world * args.train_batch * args.train_steps / el1, 1),
                "ms_per_step": round(el1 / args.train_steps * 1e3, 3)}
        tleg["ddp_step" if world > 1 else "step"] = {"ms_per_step": tleg["ms_per_step"], "images_per_s": tleg["images_per_s"]}
        if world > 1:
            el2 = _timed_steps(lambda: tstep(ddp.no_sync), 1, args.train_steps, dmod, "cpu", "gloo", lambda: None)
            tleg["no_sync_step"] = {"ms_per_step": round(el2 / args.train_steps * 1e3, 3),
                                    "images_per_s": round(world * args.train_batch * args.train_steps / el2, 1)}
            tleg["allreduce_exposed_ms"] = round((el1 - el2) / args.train_steps * 1e3, 3)
    seen = _ranks_seen(dist if world > 1 else None, rank, world, None, "gloo", mine / args.steps * 1e3)
    if rank == 0:
        print(json.dumps({
            "rccl_ranks_seen": seen, "rank_ms_per_step": [d["ms_per_step"] for d in seen],
            "metric": "meshes/sec fwd+bwd (SMPL->48x48 31-part seg)", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "meshes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "dry_run": True,
            "config": {"workload": "DRY RUN (no GPU): launcher / sharding / timing protocol only",
                       "meshes_per_gpu": B, "global_batch": B * world, "img_wh": W},
            "train_step": tleg,
            "rank_seeds": [int(v) for v in seeds], "rank_elapsed_s": times, "rank_work_s": works,
            "rank_input_checksums": sums}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="meshes per GPU")
    ap.add_argument("--wh", type=int, default=48)
    ap.add_argument("--vertex-sampling", type=int, default=1, choices=[1, 2, 5],
                    help="the default step at the reference's vertex_sampling 2 / 5 (profiling passes; 1 = the headline)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph")
    ap.add_argument("--step", default="default",
                    choices=["default", "seg_only", "fused_loss", "unfused_loss", "both_heads", "silhouette_only"],
                    help="time a variant of the decoder step (profiling); the headline is `default`")
    ap.add_argument("--min-warmup", type=int, default=200,
                    help="floor of the untimed warm-up steps (0 for counter passes, where every launch is serialised)")
    ap.add_argument("--graph-steps", type=int, default=10,
                    help="graph mode: whole steps captured per graph launch (cut to a divisor of --steps)")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent mesh chunks per step (HIP streams / parallel graph branches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch meshes per GPU (default); strong: --global-batch meshes split over the ranks")
    ap.add_argument("--global-batch", type=int, default=1024, help="strong scaling: total meshes, split evenly")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the (data-parallel) train-step leg")
    ap.add_argument("--train-batch", type=int, default=128, help="train leg: images per GPU")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--train-global-batch", type=int, default=1024,
                    help="train leg, world > 1: the fixed global batch of its strong-scaling point (configs[4]: 1024)")
    ap.add_argument("--windows", type=int, default=10, help="timed windows of --steps for the min/median/max spread")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the N-rank launch / timing protocol over gloo with a CPU stand-in step")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks ourselves, BEFORE anything touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    if args.scaling == "strong":
        if args.global_batch % world:
            raise SystemExit("bench.py: --global-batch %d does not split over %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    # one rank per GPU; the modulo only matters when several ranks are rehearsed on a one-GPU box
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")      # "nccl" is RCCL; "gloo" for the rehearsal above
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        import datetime
        to = datetime.timedelta(minutes=5)          # (a rank that dies mid-collective must not hang the others for 10 min)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=to)
        else:
            dist.init_process_group(backend, timeout=to)

    B, W = args.batch, args.wh
    model = synthetic_smpl_model(1234)
    consts = ops.SMPLConstants.from_model(model, dev)
    vs_main = int(args.vertex_sampling)
    pt = ops.get_part_table(vs_main, dev, consts.V)
    x = torch.tensor(make_x(B, W, 1000 + rank), device=dev)      # resident in HBM before timing
    gen = torch.Generator(device="cpu").manual_seed(rank)
    dseg = torch.randn(B, W, W, 32, generator=gen).to(dev)

    def step():
        xg = x.detach().requires_grad_(True)
        verts, proj, mask, seg, silh, jt, _ls = ops.DecoderFn.apply(xg, consts, 4, W, vs_main, pt, 64, True, False,
                                                               args.streams)
        seg.backward(dseg)
        return xg.grad

    # `--step` other than "default" times a VARIANT of the decoder step instead (for the profile passes, which trace
    # whatever this process runs): the line then says so in `config.workload` and is not the headline
    variant_note = None
    if args.step != "default":
        from ilps_amd.focal_loss import class_weights
        labs_v = torch.randint(0, 32, (B, W, W), device=dev, dtype=torch.int32)
        dlp_v = torch.full((B, W * W), 1.0 / (B * W * W), device=dev)
        dsl_v = torch.randn(B, W, W, 2, device=dev)
        quiet = dict(want_verts=False, want_proj=False, want_mask=False)
        v_opts = {"seg_only": ops.DecoderOpts(**quiet),
                  "fused_loss": ops.DecoderOpts(loss=(labs_v, class_weights(dev), 2.0), want_seg=False, **quiet),
                  "unfused_loss": None, "both_heads": None,
                  "silhouette_only": ops.DecoderOpts(want_verts=False, want_mask=False, seg=False)}[args.step]
        silh_v = args.step in ("both_heads", "silhouette_only")
        variant_note = {"seg_only": "decoder fwd+bwd without writing verts / proj / mask",
                        "fused_loss": "decoder + softmax-focal loss fwd+bwd, loss head inside the rasteriser, no verts / proj / mask / seg written",
                        "unfused_loss": "decoder + softmax-focal loss fwd+bwd, scores written, smplr_focal_fwd / bwd",
                        "both_heads": "decoder fwd+bwd with the 31-part head and the silhouette head",
                        "silhouette_only": "decoder fwd+bwd with the silhouette head alone"}[args.step]

        def step():                                               # noqa: F811
            xg = x.detach().requires_grad_(True)
            o_ = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, silh_v, args.streams, False, v_opts)
            if args.step == "fused_loss":
                o_[6].backward(dlp_v)
            elif args.step == "unfused_loss":
                ops.SoftmaxFocalFn.apply(o_[3], labs_v, class_weights(dev), 2.0).backward(dlp_v)
            elif args.step == "both_heads":
                torch.autograd.backward([o_[3], o_[4]], [dseg, dsl_v])
            elif args.step == "silhouette_only":
                o_[4].backward(dsl_v)
            else:
                o_[3].backward(dseg)
            return xg.grad

    mode = args.mode
    run = step
    graph = None
    # `--graph-steps G` whole steps are captured per graph launch (each a full forward + backward of all B meshes, one
    # after the other on one stream): the ~3 us between two launches of a graph is paid once per G steps.  G is cut
    # to a divisor of --steps so that EXACTLY --steps steps run in the timed window.
    gsteps = 1
    if mode == "graph":
        gsteps = max(1, min(int(args.graph_steps), args.steps))
        while args.steps % gsteps:
            gsteps -= 1
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(gsteps):
                    static_grad = step()
            run = graph.replay
        except Exception as e:  # capture unsupported -> eager, reported in config
            sys.stderr.write("bench: HIP graph capture failed (%s); running eager\n" % e)
            mode = "eager"
            graph = None
            gsteps = 1
            run = step
            torch.cuda.synchronize()
    nrun, nwarm = args.steps // gsteps, -(-max(args.warmup, args.min_warmup) // gsteps)   # launches of `run` = steps / gsteps

    # untimed: the W warm-up steps asked for, and at least --min-warmup (default 200: 30 ms of replays) so that the GPU's clock has settled
    # before the timed window (with 10 the first window of 50 steps read 5 % over the nine that followed it)
    for _ in range(nwarm):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nrun):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_ms = elapsed / args.steps * 1e3                 # this rank's own window, before the MAX over ranks
    if dist:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    seen = _ranks_seen(dist, rank, world, dev, backend, own_ms)
    # the spread: further windows of the same K steps, same bracketing (the first window above is `value`)
    windows = [elapsed / args.steps * 1e3]
    for _ in range(max(0, args.windows - 1)):
        windows.append(_timed_steps(run, 0, nrun, dist, dev, backend, torch.cuda.synchronize) / args.steps * 1e3)
    # the same K steps with ONE step per graph launch: what a caller gets who replays the decoder's graph once per
    # training step, between two encoder steps (the headline replays `gsteps` decoder steps per launch)
    ms_graph1 = None
    if mode == "graph" and gsteps > 1:
        try:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                step()
            # (K host launches per window: the smallest of three windows - a busy host core shows up here first)
            g1w = [_timed_steps(g1.replay, 20 if i == 0 else 0, args.steps, dist, dev, backend, torch.cuda.synchronize)
                   / args.steps * 1e3 for i in range(3)]
            ms_graph1 = min(g1w)
            del g1
        except Exception as e:
            sys.stderr.write("bench: one-step graph failed (%s)\n" % e)
    elif mode == "graph":
        ms_graph1 = windows[0]
    # the design's one collective: the data-parallel train step (every rank takes part)
    tleg = None
    if not args.no_train_leg:
        try:
            tleg = train_leg(args, rank, world, dev, dist, backend, model, W)
        except Exception as e:              # the decoder line must still come out: the leg's failure is reported in it
            tleg = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("bench: train leg failed on rank %d: %s\n" % (rank, tleg["error"]))

    # BASELINE configs[3] as written: the same train step at B = 256 on one GPU (VERDICT r04 missing #4)
    tleg256 = None
    if not args.no_train_leg and world == 1 and int(args.train_batch) != 256:
        try:
            tleg256 = train_leg(args, rank, world, dev, dist, backend, model, W, per=256)
        except Exception as e:
            tleg256 = {"error": "%s: %s" % (type(e).__name__, e)}

    line = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        line = {
            "metric": "meshes/sec fwd+bwd (SMPL->48x48 31-part seg)",
            "value": round(value, 1), "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_windows": {"n": len(windows), "steps_each": args.steps, "min": round(min(windows), 4),
                                    "median": round(float(np.median(windows)), 4), "max": round(max(windows), 4),
                                    "note": "the first window is `ms_per_step` / `value`; boxes differ by +-4 %"},
            "build_id": _lib.build_id(),
            "warmup_actual": nwarm * gsteps,
            "ms_per_step_graph1": None if ms_graph1 is None else round(ms_graph1, 4),
            "rccl_ranks_seen": seen, "rank_ms_per_step": [d["ms_per_step"] for d in seen],
            "train_step": tleg, "train_step_B256": tleg256,
            "config": {"workload": ("full decoder fwd+bwd (batch_smpl + projection + compute_mask + "
                                    "projects_to_seg), BASELINE configs[2]" if variant_note is None
                                    else "VARIANT --step %s (not the headline): %s" % (args.step, variant_note)),
                       "meshes_per_gpu": B, "global_batch": B * world, "img_wh": W, "vertex_sampling": vs_main, "verts": 6890,
                       "params_per_mesh": 86, "launch": mode, "steps_per_graph_launch": gsteps,
                       "concurrent_chunks": args.streams,
                       "blend_gemm": ("bf16x3: fp32 operands as 3 bf16 terms (24 significant bits), 6 partial "
                                      "products, fp32 accumulation" if ops.blend_gemm_mode() == "bf16x3"
                                      else "fp32 MFMA"),
                       "sharding": "by mesh, no collective"},
        }
        if not args.no_breakdown:
            stages = stage_breakdown(x, consts, pt, W)
            line["stages_us"] = stages
            line["stage_rooflines"] = stage_rooflines(stages, B, W, consts.V, ops.blend_gemm_mode())
            # the whole step as a stream: HBM-side bytes of its kernels from the committed PMC passes (B = 128, W = 48)
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile) and B == 128 and W == 48:
                try:
                    ks = json.load(open(tfile))["kernels"]
                    step_ks = [k for k in ks if "pack" not in k and "copy" not in k and not k.startswith("at::")]
                    tot = sum(ks[k].get("hbm_bytes_per_launch", 0) for k in step_ks)
                    gbs = tot / (ms * 1e-3) / 1e9
                    line["step_hbm"] = {"bound": "hbm", "traffic": int(tot), "achieved": round(gbs, 1),
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                        "note": "PMC FETCH_SIZE/WRITE_SIZE bytes of the step's %d kernels (profiles/"
                                                "pmc_traffic.json) / graph step time: a chain of launch- and "
                                                "latency-bound kernels at B = 128, not a bandwidth-bound stream" % len(step_ks)}
                except Exception:
                    pass
            line["roofline"] = raster_roofline(x, consts, pt, W, stages)
            # the same step with the blend GEMMs on the fp32 matrix cores (SMPLR_BLEND_GEMM=f32) beside the bf16x3 default
            try:
                c32 = consts.fp32_gemm()

                def step32():
                    xg = x.detach().requires_grad_(True)
                    _v, _p, _m, sg_, _s, _j, _l = ops.DecoderFn.apply(xg, c32, 4, W, 1, pt, 64, True, False, 1)
                    sg_.backward(dseg)
                step32()
                t32 = graph_time_ms(step32, 10, torch.cuda.current_stream())
                # the strict-fp32 headline beside `value`: same step, same batch, ten steps per graph replay
                line["value_f32"] = round(world * B / (t32 * 1e-3), 1)
                line["ms_per_step_f32"] = round(t32, 4)
                line["step_blend_gemm_f32"] = {"ms_per_step": round(t32, 4), "meshes_per_s": round(B / (t32 * 1e-3), 1),
                                               "note": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) blend GEMMs; the headline "
                                                       "uses bf16x3 (3 x 8 = 24 significant bits per operand)"}
            except Exception as e:
                line["step_blend_gemm_f32"] = {"error": str(e)}
        if not args.no_breakdown:
            # BASELINE configs[1]: batch_smpl fwd+bwd only, B=256, eager, HIP events
            x2 = torch.tensor(make_x(256, W, 7), device=dev)
            gv = torch.randn(256, consts.V, 3, device=dev)

            def smpl_only():
                xg = x2.detach().requires_grad_(True)
                v, _jt = ops.BatchSMPLFn.apply(xg, consts, 4)
                v.backward(gv)
            for _ in range(5):
                smpl_only()
            t_smpl = event_time_ms(smpl_only, 30, torch.cuda.current_stream())
            line["aux"] = {"batch_smpl_fwd_bwd_B256": {"meshes_per_s": round(256 / (t_smpl * 1e-3), 1),
                                                       "ms_per_step": round(t_smpl, 4), "launch": "eager"}}
            try:                                   # the same step replayed from a HIP graph (no host launch time)
                sg = torch.cuda.Stream()
                sg.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sg):
                    smpl_only()
                torch.cuda.current_stream().wait_stream(sg)
                torch.cuda.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    smpl_only()
                for _ in range(5):
                    g2.replay()
                torch.cuda.synchronize()
                t0g = time.perf_counter()
                for _ in range(100):
                    g2.replay()
                torch.cuda.synchronize()
                t_g = (time.perf_counter() - t0g) / 100 * 1e3
                line["aux"]["batch_smpl_fwd_bwd_B256"].update(
                    {"graph_meshes_per_s": round(256 / (t_g * 1e-3), 1), "graph_ms_per_step": round(t_g, 4)})
            except Exception as e:
                line["aux"]["batch_smpl_fwd_bwd_B256"]["graph_error"] = str(e)
            # BASELINE configs[0] shape on the GPU: predict.py's forward for ONE 256x256 image (ENet + IEF on stock
            # torch ops, random weights; decoder on the HIP path), eager launches, HIP events
            try:
                from ilps_amd.decoder import SMPLDecoder
                from ilps_amd.inference import predict_batch
                from ilps_amd.model import SMPLRegressor
                torch.manual_seed(0)
                reg = SMPLRegressor(W, "enet", True).to(dev).eval()
                dec1 = SMPLDecoder(model, img_wh=W)
                img = torch.rand(1, 3, 256, 256, device=dev)
                for _ in range(3):
                    predict_batch(reg, dec1, img)
                t_p = event_time_ms(lambda: predict_batch(reg, dec1, img), 10, torch.cuda.current_stream())
                dec_only = lambda: ops.DecoderFn.apply(x[:1], consts, 4, W, 1, pt, 64, True, False, 1)
                dec_only()
                t_d = graph_time_ms(dec_only, 10, torch.cuda.current_stream())
                line["aux"]["predict_B1"] = {"ms_per_image": round(t_p, 3), "decoder_forward_us": round(t_d * 1e3, 1),
                                             "note": "encoder + regressor + decoder forward for one image, eager launches "
                                                     "(host-bound); decoder_forward_us = its 5 kernels replayed from a graph"}
                # the decoder's side of that eager forward by binding: one at::Tensor op (torch.ops.smplraster.decoder_fwd,
                # what SMPLDecoder runs without gradients) against the autograd node's ctypes calls - host wall time per call
                try:
                    x1 = x[:1].contiguous()

                    def host_us(n=200):
                        with torch.no_grad():
                            for _ in range(20):
                                dec1(x1)
                            torch.cuda.synchronize()
                            t0h = time.perf_counter()
                            for _ in range(n):
                                dec1(x1)
                            torch.cuda.synchronize()
                        return (time.perf_counter() - t0h) / n * 1e6
                    prev = os.environ.get("SMPLR_TORCH_OPS")
                    os.environ["SMPLR_TORCH_OPS"] = "0"
                    us_ct = host_us()
                    t_p0 = event_time_ms(lambda: predict_batch(reg, dec1, img), 10, torch.cuda.current_stream())
                    os.environ["SMPLR_TORCH_OPS"] = "1"
                    us_op = host_us()
                    if prev is None:
                        del os.environ["SMPLR_TORCH_OPS"]
                    else:
                        os.environ["SMPLR_TORCH_OPS"] = prev
                    line["aux"]["predict_B1"].update({"decoder_eager_us_ctypes": round(us_ct, 1),
                                                      "decoder_eager_us_torch_op": round(us_op, 1),
                                                      "ms_per_image_ctypes_decoder": round(t_p0, 3)})
                except Exception as e:
                    line["aux"]["predict_B1"]["binding_error"] = "%s: %s" % (type(e).__name__, e)
                try:
                    from ilps_amd.inference import GraphedPredictor
                    gp = GraphedPredictor(reg, dec1, img)
                    t_g = event_time_ms(lambda: gp(img), 20, torch.cuda.current_stream())
                    line["aux"]["predict_B1"]["graph_ms_per_image"] = round(t_g, 3)
                    del gp
                except Exception as e:
                    line["aux"]["predict_B1"]["graph_error"] = str(e)
                del reg, dec1, img
            except Exception as e:
                line["aux"]["predict_B1"] = {"error": str(e)}
            # the reference's own renderer-profiling workload (profiling_renderer.py:19-39): ONE sample of 6890 uniform
            # random vertices x 80, compute_mask over all of them, projects_to_seg at 48x48 with vertex_sampling = 5
            # (which gathers positions ids // 5 of the 6890-long list, i.e. from its first 1378 rows), forward only
            try:
                g5 = torch.Generator(device="cpu").manual_seed(5)
                pr = (torch.rand(1, 6890, 3, generator=g5) * 80.0).to(dev)
                pt5 = ops.get_part_table(5, dev, consts.V)

                def ref_recipe():
                    mk = ops.visibility(pr)
                    return ops._seg_fwd(pr[:, :pt5.VP].contiguous(), mk[:, :pt5.VP].contiguous(), 48, pt5)
                ref_recipe()
                t_r = graph_time_ms(ref_recipe, 10, torch.cuda.current_stream())
                entry = {"gpu_us": round(t_r * 1e3, 1), "batch": 1, "img_wh": 48, "vertex_sampling": 5}
                if world == 1 and not args.no_cpu_baseline:
                    from oracle import np_oracle as no_
                    ids5, off5 = load_part_tables(5)
                    pj5 = pr.cpu().numpy().astype(np.float64)
                    t0c = time.perf_counter()
                    mk5 = no_.compute_mask(pj5)
                    no_.projects_to_seg(pj5[:, :pt5.VP], mk5[:, :pt5.VP], 48, ids5, off5, 5)
                    entry["cpu_port_ms"] = round((time.perf_counter() - t0c) * 1e3, 1)
                line["aux"]["reference_renderer_profile_workload"] = entry
            except Exception as e:
                line["aux"]["reference_renderer_profile_workload"] = {"error": str(e)}
            # ... and the same recipe at the headline's batch, vertex_sampling None and 5, forward + backward
            line["aux"]["reference_raster_recipe_B%d" % B] = {}
            for nm, vs_ in (("vs_none", 1), ("vs5", 5)):
                try:
                    line["aux"]["reference_raster_recipe_B%d" % B][nm] = raster_recipe_leg(B, 48, vs_, dev, consts)
                except Exception as e:
                    line["aux"]["reference_raster_recipe_B%d" % B][nm] = {"error": "%s: %s" % (type(e).__name__, e)}
            # the reference's other shipped sizes (VERDICT r03 missing #2, #3): W = 64 and vertex_sampling 2 / 5
            line["aux"]["reference_sizes"] = {}
            for nm, (W_, vs_) in (("w64", (64, 1)), ("w48_vs2", (48, 2)), ("w48_vs5", (48, 5)), ("w64_vs5", (64, 5))):
                try:
                    xs_ = x if W_ == W else torch.tensor(make_x(B, W_, 1000 + rank), device=dev)
                    line["aux"]["reference_sizes"][nm] = size_leg(xs_, consts, W_, vs_, dev)
                except Exception as e:
                    line["aux"]["reference_sizes"][nm] = {"error": "%s: %s" % (type(e).__name__, e)}
            # silhouette rasteriser (part of configs[4]'s second loss), same meshes
            c0s = ops._pose_fwd(x, 4, consts)
            pjs = ops._skin_fwd(ops._blend_fwd(c0s[0], consts, x.shape[0]), c0s[3], consts, cam=x)[1]
            sil, sarg = ops._silh_fwd(pjs, W)
            dsil = torch.randn_like(sil)
            t_sf = event_time_ms(lambda: ops._silh_fwd(pjs, W), 20, torch.cuda.current_stream())
            t_sb = event_time_ms(lambda: ops._silh_bwd(dsil, sil, sarg, pjs, W), 20, torch.cuda.current_stream())
            line["aux"]["silhouette"] = {"fwd_us": round(t_sf * 1e3, 2), "bwd_us": round(t_sb * 1e3, 2), "meshes": B}
            # the decoder of configs[4]: part segmentation AND silhouette from one pass, fwd+bwd of both heads
            try:
                dsl = torch.randn(B, W, W, 2, device=dev)

                def step_silh():
                    xg = x.detach().requires_grad_(True)
                    _v, _p, _m, sg_, sl_, _j, _l = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, True, 1)
                    torch.autograd.backward([sg_, sl_], [dseg, dsl])
                step_silh()
                t_ds = graph_time_ms(step_silh, 5, torch.cuda.current_stream())
                line["aux"]["decoder_with_silhouette"] = {"ms_per_step": round(t_ds, 4),
                                                          "meshes_per_s": round(B / (t_ds * 1e-3), 1)}
            except Exception as e:
                line["aux"]["decoder_with_silhouette"] = {"error": str(e)}
            # decoder + loss head fwd+bwd (the TRAIN path's decoder): unfused = scores written,
            # smplr_focal_fwd / bwd, dseg read back; fused = the loss head inside the rasteriser (DecoderOpts.loss), the
            # (B,W,W,32) scores and their gradient never in memory; seg_only = the default step without verts / proj / mask
            try:
                from ilps_amd.focal_loss import class_weights
                labs = torch.randint(0, 32, (B, W, W), device=dev, dtype=torch.int32)
                cwt = class_weights(dev)
                dlp = torch.full((B, W * W), 1.0 / (B * W * W), device=dev)

                def step_unfused():
                    xg = x.detach().requires_grad_(True)
                    o_ = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1)
                    ops.SoftmaxFocalFn.apply(o_[3], labs, cwt, 2.0).backward(dlp)

                def mk_fused(**kw):
                    op_ = ops.DecoderOpts(loss=(labs, cwt, 2.0), want_seg=False, **kw)

                    def f():
                        xg = x.detach().requires_grad_(True)
                        ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1, False, op_)[6].backward(dlp)
                    return f
                step_fused_all = mk_fused()
                step_fused = mk_fused(want_verts=False, want_proj=False, want_mask=False)
                op_so = ops.DecoderOpts(want_verts=False, want_proj=False, want_mask=False)

                def step_seg_only():
                    xg = x.detach().requires_grad_(True)
                    ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1, False, op_so)[3].backward(dseg)
                ent = {}
                for nm, fn in (("unfused", step_unfused), ("fused", step_fused), ("fused_all_outputs", step_fused_all),
                               ("seg_only_no_loss", step_seg_only)):
                    fn()
                    tms = graph_time_ms(fn, 5, torch.cuda.current_stream())
                    ent[nm] = {"ms_per_step": round(tms, 4), "meshes_per_s": round(B / (tms * 1e-3), 1)}
                ent["note"] = ("decoder + softmax-focal loss fwd+bwd; fused = loss head as the rasteriser's epilogue / "
                               "seg_bwd's prologue, no verts / proj / mask / seg written; seg_only_no_loss = the headline "
                               "step without verts / proj / mask")
                line["aux"]["decoder_plus_loss"] = ent
            except Exception as e:
                line["aux"]["decoder_plus_loss"] = {"error": str(e)}
            # the silhouette-only pass of the alternating stage-2 schedule (train_stage2_silhouette.py:262-270)
            try:
                op_s = ops.DecoderOpts(want_verts=False, want_mask=False, seg=False)
                dsl2 = torch.randn(B, W, W, 2, device=dev)

                def step_silh_only():
                    xg = x.detach().requires_grad_(True)
                    ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, True, 1, False, op_s)[4].backward(dsl2)
                step_silh_only()
                t_so = graph_time_ms(step_silh_only, 5, torch.cuda.current_stream())
                line["aux"]["silhouette_only_step"] = {"ms_per_step": round(t_so, 4), "meshes_per_s": round(B / (t_so * 1e-3), 1)}
            except Exception as e:
                line["aux"]["silhouette_only_step"] = {"error": str(e)}
            # the same step with the bit-reproducible backward (SMPLDecoder(deterministic=True): 64-bit fixed-point
            # accumulation in seg_bwd instead of fp32 LDS atomics) - its price
            try:
                def step_det():
                    xg = x.detach().requires_grad_(True)
                    _v, _p, _m, sg_, _s, _j, _l = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1, True)
                    sg_.backward(dseg)
                step_det()
                t_dd = graph_time_ms(step_det, 5, torch.cuda.current_stream())
                line["aux"]["deterministic_step"] = {"ms_per_step": round(t_dd, 4), "meshes_per_s": round(B / (t_dd * 1e-3), 1),
                                                     "note": "gradients bit-identical run to run (tests/test_gpu_baseline_sizes.py)"}
            except Exception as e:
                line["aux"]["deterministic_step"] = {"error": str(e)}
            # loss head: softmax + focal loss on the (B,W,W,32) scores, HBM-bound.
            # algorithmic bytes: fwd = scores 128 + label 4 + loss 4 B/pixel; bwd = 128 + 4 + 4 + 128 B/pixel
            seg_s = torch.rand(B, W, W, 32, device=dev)
            lab = torch.randint(0, 32, (B, W * W), device=dev, dtype=torch.int32)
            cw = torch.ones(32, device=dev)
            dl = torch.full((B, W * W), 1.0 / (B * W * W), device=dev)
            lossb, dsc = torch.empty(B, W * W, device=dev), torch.empty_like(seg_s)
            lib, P_, st_ = _lib.load(), _lib.ptr, _lib.stream
            f_fwd = lambda: lib.smplr_focal_fwd(P_(seg_s), P_(lab), None, P_(cw), 2.0, B * W * W, 32, P_(lossb),
                                                None, st_())
            f_bwd = lambda: lib.smplr_focal_bwd(P_(seg_s), P_(lab), None, P_(cw), 2.0, P_(dl), B * W * W, 32,
                                                P_(dsc), st_())
            for _ in range(3):
                f_fwd(); f_bwd()
            tf_ms = event_time_ms(f_fwd, 50, torch.cuda.current_stream())
            tb_ms = event_time_ms(f_bwd, 50, torch.cuda.current_stream())
            npx = B * W * W
            line["aux"]["loss_head_softmax_focal"] = {
                "fwd_us": round(tf_ms * 1e3, 2), "bwd_us": round(tb_ms * 1e3, 2),
                "fwd_GBps": round(npx * 136 / (tf_ms * 1e-3) / 1e9, 1),
                "bwd_GBps": round(npx * 264 / (tb_ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000,
                "note": "latency-bound at B=128 (37.7 MB tensor); bytes = algorithmic, per SURVEY 8(f) next-2"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(model, W)
            except Exception as e:
                line["cpu_baseline"] = {"error": str(e)}
            try:
                line["parity"] = parity_sample(model, consts, pt, W, dev)
            except Exception as e:
                line["parity"] = {"error": str(e)}
        print(json.dumps(line), flush=True)
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:              # (only reachable after a failed leg: the line is already out)
            sys.stderr.write("bench: process group shutdown: %s\n" % e)


if __name__ == "__main__":
    main()
