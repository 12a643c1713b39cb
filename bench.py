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
  roofline      - the dominant kernel (segmentation raster forward), timed live with HIP events on
                  the launch stream; achieved = algorithmic FLOPs per launch / mean launch time
  cpu_baseline  - the oracle's fp32 reference-shaped torch-CPU restatement timed on this box's host
                  cores on a bounded sample (rank 0, N=1 only); a reported baseline, not the target
"""
import argparse
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
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            g.replay()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / (3 * reps)
    except Exception:
        torch.cuda.synchronize()
        return event_time_ms(fn, reps, stream)


def stage_breakdown(x, consts, pt, W, reps=20):
    """Per-C-ABI-call mean times (us) on the current stream; same kernels, same call sequence as the timed step."""
    st = torch.cuda.current_stream()
    B = x.shape[0]
    VP = consts.V
    res = {}
    coef, Rs, J, A, Jt = ops._pose_fwd(x, 4, consts)
    res["pose_fwd"] = graph_time_ms(lambda: ops._pose_fwd(x, 4, consts), reps, st)
    v_posed = ops._blend_fwd(coef, consts, B)
    res["blend_fwd"] = graph_time_ms(lambda: ops._blend_fwd(coef, consts, B), reps, st)
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


def cpu_baseline(model, W, budget_s=20.0):
    """fp32 reference-shaped CPU restatement (oracle/torch_oracle.py), fwd+bwd, bounded sample."""
    from oracle import torch_oracle as to
    from oracle import np_oracle as no
    # the GPU box's CPU share for one GPU is 16 cores whatever os.cpu_count() says
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(avail, 16))
    torch.set_num_threads(ncores)
    ids, off = load_part_tables(1)
    smpl = to.TorchSMPL(model, dtype=torch.float32)
    Bc = 1
    x = torch.tensor(make_x(Bc, W, 123), requires_grad=True)
    g = torch.randn(Bc, W, W, 32)

    def one():
        if x.grad is not None:
            x.grad = None
        verts, proj, mask, seg = to.decoder_forward(
            smpl, x, lambda p: torch.tensor(no.compute_mask(p.numpy().astype(np.float64)), dtype=torch.float32),
            W, ids, off)
        (seg * g).sum().backward()

    one()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    cpu_model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(Bc * n / el, 3), "unit": "meshes/s", "cores": int(torch.get_num_threads()),
            "cpu": cpu_model, "os_cpu_count": os.cpu_count(),
            "kind": "port",
            "sample": "%d fwd+bwd passes of the full decoder at B=%d, W=%d (%.1f s); fp32 torch-CPU "
                      "restatement of the reference's dense formulation (TensorFlow itself is not "
                      "installable here)" % (n, Bc, W, el)}


def parity_sample(model, consts, pt, W, dev, Bp=2):
    """The other half of BASELINE's metric ("vert Linf vs ref"): the HIP path against the float64 oracle on Bp
    seeded meshes - vertex L-inf, and the largest |seg - ref| / (1e-3 |ref| + 1e-6) (<= 1 passes the bar)."""
    from oracle import np_oracle as no
    xs = make_x(Bp, W, 4242)
    x = torch.tensor(xs, device=dev)
    verts, proj, mask, seg, _silh, _jt = ops.DecoderFn.apply(x, consts, 4, W, 1, pt, 64, True, False, 1)
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
    if rank == 0:
        print(json.dumps({
            "metric": "meshes/sec fwd+bwd (SMPL->48x48 31-part seg)", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "meshes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "dry_run": True,
            "config": {"workload": "DRY RUN (no GPU): launcher / sharding / timing protocol only",
                       "meshes_per_gpu": B, "global_batch": B * world, "img_wh": W},
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
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent mesh chunks per step (HIP streams / parallel graph branches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B, W = args.batch, args.wh
    model = synthetic_smpl_model(1234)
    consts = ops.SMPLConstants.from_model(model, dev)
    pt = ops.get_part_table(1, dev, consts.V)
    x = torch.tensor(make_x(B, W, 1000 + rank), device=dev)      # resident in HBM before timing
    gen = torch.Generator(device="cpu").manual_seed(rank)
    dseg = torch.randn(B, W, W, 32, generator=gen).to(dev)

    def step():
        xg = x.detach().requires_grad_(True)
        verts, proj, mask, seg, silh, jt = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False,
                                                               args.streams)
        seg.backward(dseg)
        return xg.grad

    mode = args.mode
    run = step
    graph = None
    if mode == "graph":
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
                static_grad = step()
            run = graph.replay
        except Exception as e:  # capture unsupported -> eager, reported in config
            sys.stderr.write("bench: HIP graph capture failed (%s); running eager\n" % e)
            mode = "eager"
            graph = None
            run = step
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    line = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        line = {
            "metric": "meshes/sec fwd+bwd (SMPL->48x48 31-part seg)",
            "value": round(value, 1), "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "full decoder fwd+bwd (batch_smpl + projection + compute_mask + "
                                   "projects_to_seg), BASELINE configs[2]",
                       "meshes_per_gpu": B, "global_batch": B * world, "img_wh": W, "verts": 6890,
                       "params_per_mesh": 86, "launch": mode, "concurrent_chunks": args.streams,
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
                    tot = sum(v.get("hbm_bytes_per_launch", 0) for k, v in ks.items()
                              if "pack" not in k and "copy" not in k)
                    gbs = tot / (ms * 1e-3) / 1e9
                    line["step_hbm"] = {"bound": "hbm", "traffic": int(tot), "achieved": round(gbs, 1),
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                        "note": "PMC FETCH_SIZE/WRITE_SIZE bytes of the step's 9 kernels (profiles/"
                                                "pmc_traffic.json) / graph step time: a chain of launch- and "
                                                "latency-bound kernels at B = 128, not a bandwidth-bound stream"}
                except Exception:
                    pass
            t_seg = stages["vis_seg_fwd"] * 1e-6
            flop = SEG_FWD_FLOP_PER_MESH * B if W == 48 else (W * W * 6879 * 7 + W * W * 62) * B
            ach = flop / t_seg / 1e12
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get("seg_fwd_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # executed work: pairs actually evaluated (visible records only) x ~10 FLOP (9 VALU ops, one an FMA)
            c0 = ops._pose_fwd(x, 4, consts)
            pj = ops._skin_fwd(ops._blend_fwd(c0[0], consts, x.shape[0]), c0[3], consts, cam=x)[1]
            nvis = float((ops.visibility(pj) == 1.0).sum().item()) / B
            executed = (W * W * nvis * 10.0 * B) / t_seg / 1e12
            line["roofline"] = {
                "kernel": "seg_bin_kernel + raster_fwd_kernel (smplr_vis_seg_fwd)", "bound": "mfma",
                "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                "executed_tflops": round(executed, 3), "executed_frac": round(executed / FP32_PEAK_TFLOPS, 4),
                "visible_vertices_per_mesh": round(nvis, 1),
                "note": "compute roof (the schema offers hbm|mfma): the pair loop runs on the fp32 VALU, whose peak "
                        "equals the fp32 MFMA peak (157.3 TF). achieved = ALGORITHMIC FLOPs (SURVEY 8(d): 111 "
                        "MFLOP/mesh x B) / launch time; it can exceed the pipe's real utilisation because pairs "
                        "whose fp32 score is provably 0 are never evaluated: executed_* counts only evaluated pairs",
                "launch_us": stages["vis_seg_fwd"],
            }
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
                    _v, _p, _m, sg_, sl_, _j = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, True, 1)
                    torch.autograd.backward([sg_, sl_], [dseg, dsl])
                step_silh()
                t_ds = graph_time_ms(step_silh, 5, torch.cuda.current_stream())
                line["aux"]["decoder_with_silhouette"] = {"ms_per_step": round(t_ds, 4),
                                                          "meshes_per_s": round(B / (t_ds * 1e-3), 1)}
            except Exception as e:
                line["aux"]["decoder_with_silhouette"] = {"error": str(e)}
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
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
