"""model.py counterpart on CPU: shapes, parameter counts, IEF semantics, focal loss vs the reference
formula, DDP gradient equivalence over gloo (the decoder itself needs a GPU: tests/test_gpu_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_enet_shapes_and_params():
    from ilps_amd.model import ENetEncoder, SMPLRegressor
    enet = ENetEncoder().eval()
    with torch.no_grad():
        y = enet(torch.zeros(1, 3, 256, 256))
    assert y.shape == (1, 128, 32, 32)
    n_enet = sum(p.numel() for p in enet.parameters())
    assert 300_000 < n_enet < 400_000                      # 338 k
    reg = SMPLRegressor(48, "enet", use_IEF=True).eval()
    n = sum(p.numel() for p in reg.parameters())
    assert abs(n - 14.58e6) < 0.3e6                         # ENet + IEF = 14.58 M
    with torch.no_grad():
        out = reg(torch.zeros(2, 256, 256, 3))              # NHWC accepted
    assert out.shape == (2, 86)
    assert torch.allclose(out[:, :4], torch.tensor([24.0, 24.0, 24.0, 30.0]), atol=0.5)


def test_plain_mlp_head_adds_mean_and_scales():
    from ilps_amd.model import SMPLRegressor
    from ilps_amd.smpl_model import mean86
    reg = SMPLRegressor(64, "enet", use_IEF=False).eval()
    for p in reg.mlp.parameters():
        torch.nn.init.zeros_(p)
    with torch.no_grad():
        out = reg(torch.zeros(1, 3, 256, 256))
    assert np.allclose(out[0].numpy(), mean86(64), atol=1e-6)   # zero regressor -> mean + camera


def test_enet_needs_256_inputs():
    from ilps_amd.model import SMPLRegressor
    reg = SMPLRegressor(48, "enet").eval()
    with pytest.raises(RuntimeError, match="256x256"):
        with torch.no_grad():
            reg(torch.zeros(1, 3, 128, 128))


def test_focal_loss_matches_reference_formula():
    from ilps_amd.focal_loss import categorical_focal_loss, classlab, CLASS_WEIGHTS
    rng = np.random.default_rng(0)
    logits = torch.tensor(rng.normal(0, 1, (2, 9, 32)), dtype=torch.float64)
    probs = torch.softmax(logits, -1)
    lab = torch.tensor(rng.integers(0, 32, (2, 3, 3)))
    onehot = classlab(lab).reshape(2, 9, 32).double()
    assert onehot.sum(-1).eq(1).all()
    for wc in (False, True):
        f = categorical_focal_loss(2.0, wc)
        a, b = f(onehot, probs), f(lab, probs)
        assert a.shape == (2, 9) and torch.allclose(a, b)
        p = probs.numpy().clip(1e-7, 1 - 1e-7)
        w = np.asarray(CLASS_WEIGHTS) if wc else np.ones(32)
        want = ((1 - p) ** 2 * (-onehot.numpy() * np.log(p)) * w).sum(-1)      # focal_loss.py:17-44
        assert np.allclose(a.numpy(), want)
    assert CLASS_WEIGHTS[0] == 0.3 and CLASS_WEIGHTS[25] == 2.0 and CLASS_WEIGHTS[5] == 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ilps_amd  # noqa: F401
    from ilps_amd.sharding import shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    head = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ReLU(), torch.nn.Linear(32, 86))
    ddp = torch.nn.parallel.DistributedDataParallel(head, bucket_cap_mb=1)
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn(8, 64, generator=g), torch.randn(8, 86, generator=g)
    lo, hi = shard_range(8, rank, world)
    ((ddp(x[lo:hi]) - tgt[lo:hi]) ** 2).mean().backward()         # DDP averages the rank gradients
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ReLU(), torch.nn.Linear(32, 86))
        ref.load_state_dict(head.state_dict())
        ((ref(x) - tgt) ** 2).mean().backward()
        ok = all(torch.allclose(a.grad, b.grad, atol=1e-6) for a, b in zip(head.parameters(), ref.parameters()))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradient_equals_full_batch_gloo():
    """The data-parallel step (one rank per GPU, all-reduce of encoder/regressor gradients) gives the
    gradient of the full batch: world_size 2 over gloo on CPU stands in for RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
