"""world_size-2 gloo test of the N>1 path: mesh sharding with no data-path collective, and the
bench's max-over-ranks timing reduction.  CPU only (the decoder kernels are covered by -m gpu);
the per-rank work here is the CPU-runnable part of the path (parameter conditioning)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ilps_amd  # noqa: F401
    from ilps_amd.sharding import shard_range
    from ilps_amd.keras_smpl.set_cam_params import load_mean_set_cam_params
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 10
    full = torch.arange(G * 86, dtype=torch.float32).reshape(G, 86) * 1e-3
    lo, hi = shard_range(G, rank, world)
    mine = load_mean_set_cam_params(full[lo:hi], 48)          # rank-local rows, no exchange
    sizes = [shard_range(G, r, world) for r in range(world)]
    bufs = [torch.zeros(b - a, 86) for a, b in sizes]
    dist.all_gather(bufs, mine)                                # test-only gather to compare
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # bench.py's timing reduction
    if rank == 0:
        want = load_mean_set_cam_params(full, 48)
        out.put((torch.allclose(torch.cat(bufs), want), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and abs(tmax - 0.2) < 1e-12
