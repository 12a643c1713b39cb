#!/usr/bin/env python3
"""Auxiliary benchmark (BASELINE configs[3]/[4]): end-to-end train step, ENet(256x256x3) + IEF + HIP decoder +
focal loss + Adam, B meshes per GPU; under torchrun it is data-parallel over RCCL.  Prints one JSON line."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ilps_amd  # noqa
from ilps_amd.training import SegTrainer, init_distributed
from ilps_amd.smpl_model import synthetic_smpl_model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--encoder", default="enet")
ap.add_argument("--silhouette", action="store_true")
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
rank, world = init_distributed()
dev = torch.device("cuda", torch.cuda.current_device())
tr = SegTrainer(synthetic_smpl_model(1234), output_wh=48, encoder_architecture=a.encoder, use_IEF=True, device=dev,
                ddp=world > 1, with_silhouette=a.silhouette)
tr.smpl_model.train()
g = torch.Generator().manual_seed(rank)
images = torch.rand(a.batch, 3, 256, 256, generator=g).to(dev)
labels = torch.randint(0, 32, (a.batch, 48, 48), generator=g).to(dev)
sl = torch.nn.functional.one_hot(torch.randint(0, 2, (a.batch, 48 * 48), generator=g), 2).float().to(dev) if a.silhouette else None
for _ in range(a.warmup):
    tr.step(images, labels, sl)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step(images, labels, sl)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
el = time.perf_counter() - t0
if a.profile and rank == 0:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr.step(images, labels, sl); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))
if rank == 0:
    print(json.dumps({"workload": "train step: %s(256x256x3)+IEF+decoder(W=48)+focal+Adam%s" % (a.encoder, "+silhouette CE" if a.silhouette else ""),
                      "images_per_s": round(world * a.batch * a.steps / el, 1), "ms_per_step": round(el / a.steps * 1e3, 2),
                      "n_gpus": world, "batch_per_gpu": a.batch}))
if world > 1:
    torch.distributed.destroy_process_group()
