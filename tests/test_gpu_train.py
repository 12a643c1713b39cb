"""End-to-end train step on the GPU: ENet + IEF regressor on stock torch
ops, the HIP decoder, focal loss, Adam.  Checks that gradients reach the encoder through the
hand-written backward and that a few steps reduce the loss on a fixed batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_overfits_fixed_batch(smpl_model):
    from ilps_amd.training import SegTrainer
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    tr = SegTrainer(smpl_model, output_wh=48, encoder_architecture="enet", use_IEF=True, lr=1e-4, device=dev)
    tr.smpl_model.train()
    B = 4
    images = torch.rand(B, 3, 256, 256, device=dev)
    # labels = the segmentation of a perturbed pose, so that there is something to learn
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.smpl_model import mean86
    x = torch.tensor(np.tile(mean86(48), (B, 1)), dtype=torch.float32, device=dev)
    x[:, 4:76] += 0.1 * torch.randn(B, 72, device=dev)
    with torch.no_grad():
        labels = SMPLDecoder(smpl_model, img_wh=48)(x)["seg"].argmax(-1)          # (B,48,48)
    losses = [float(tr.step(images, labels)) for _ in range(6)]
    assert all(np.isfinite(losses))
    g = [p.grad for p in tr.smpl_model.parameters() if p.grad is not None]
    assert len(g) > 100 and any(float(t.abs().sum()) > 0 for t in g)
    first_conv = tr.smpl_model.backbone.enet.init_conv.weight.grad
    assert first_conv is not None and float(first_conv.abs().sum()) > 0          # gradient reached the stem
    assert losses[-1] < losses[0]


def test_build_model_handles(smpl_model):
    from ilps_amd.model import build_model, build_full_model_for_predict
    dev = torch.device("cuda:0")
    segs, smpl, verts, projects = build_model(2, (256, 256, 3), smpl_model, 48, 32, encoder_architecture="enet",
                                              use_IEF=True)
    for m in (segs, smpl, verts, projects):
        m.to(dev).eval()
    img = torch.rand(2, 256, 256, 3, device=dev)
    with torch.no_grad():
        s, p, v, pr = segs(img), smpl(img), verts(img), projects(img)
    assert s.shape == (2, 48 * 48, 32) and torch.allclose(s.sum(-1), torch.ones_like(s.sum(-1)), atol=1e-5)
    assert p.shape == (2, 86) and v.shape == (2, 6890, 3) and pr.shape == (2, 6890, 3)
    vm, pm, sm = build_full_model_for_predict(smpl, 48, smpl_model)
    with torch.no_grad():
        raw = sm.to(dev).eval()(img)
    assert raw.shape == (2, 48, 48, 32)
    assert torch.allclose(torch.softmax(raw.reshape(2, -1, 32), -1), s, atol=1e-5)


def test_predict_batch_of_one(smpl_model):
    """BASELINE configs[0] shape: one 48x48 prediction (encoder on stock torch, decoder on the HIP path);
    the decoder part is checked against the oracle on the regressed parameters."""
    import numpy as np
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.inference import predict_batch
    from ilps_amd.model import SMPLRegressor
    from oracle import np_oracle as o
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    reg = SMPLRegressor(48, "enet", True).to(dev)
    dec = SMPLDecoder(smpl_model, img_wh=48)
    img = torch.rand(1, 3, 256, 256, device=dev)
    out = predict_batch(reg, dec, img)
    assert out["verts"].shape == (1, 6890, 3) and out["segs"].shape == (1, 48, 48, 32)
    assert out["seg_maps"].shape == (1, 48, 48) and out["seg_maps"].dtype == torch.int64
    x = out["smpl"].cpu().numpy().astype(np.float64)
    ref = o.smpl_layer_call(x, smpl_model)
    assert np.abs(out["verts"].cpu().numpy() - ref).max() <= 1e-4
    assert torch.equal(out["seg_maps"], out["segs"].argmax(-1))


@pytest.mark.parametrize("shape", [(3, 16, 128, 128), (2, 5, 7, 9), (4, 64, 1, 1), (1, 3, 70, 70)])
def test_prelu_kernels_match_torch(shape):
    """smplr_prelu_fwd/bwd against torch's own PReLU (fp64 on CPU): values, input gradient, slope gradient."""
    from ilps_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    w = torch.rand(shape[1], generator=g) * 0.5 - 0.1
    gy = torch.randn(*shape, generator=g)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.PReLUFn.apply(xd, wd)
    y.backward(gy.to(dev))
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = torch.nn.functional.prelu(x64, w64)
    y64.backward(gy.double())
    assert torch.allclose(y.detach().cpu().double(), y64.detach(), rtol=1e-6, atol=1e-7)
    assert torch.allclose(xd.grad.cpu().double(), x64.grad, rtol=1e-6, atol=1e-7)
    assert torch.allclose(wd.grad.cpu().double(), w64.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,with_act", [((6, 16, 128, 128), True), ((5, 3, 33, 31), True), ((4, 64, 32, 32), False),
                                            ((2, 5, 16, 16), True), ((3, 7, 70, 70), False)])
def test_batch_norm_act_kernels_match_torch(shape, with_act):
    """smplr_bn_fwd/bwd (training-mode BatchNorm2d + optional PReLU in one op) against torch's own modules in
    float64 on the CPU: output, input gradient, gamma / beta / slope gradients, running statistics and the batch
    counter; twice in a row on the same input must be bit-identical (fixed summation order)."""
    from ilps_amd import ops
    from ilps_amd.model import PReLU
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    x = torch.randn(*shape, generator=g) * 2.0 + 0.5
    gy = torch.randn(*shape, generator=g)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    act = PReLU(C) if with_act else None
    if act is not None:
        with torch.no_grad():
            act.weight.copy_(torch.rand(C, generator=g) * 0.5 - 0.1)
    import copy
    bn64, act64 = copy.deepcopy(bn).double().train(), (copy.deepcopy(act).double() if act is not None else None)
    bnd, actd = copy.deepcopy(bn).to(dev).train(), (copy.deepcopy(act).to(dev) if act is not None else None)
    xd = x.to(dev).requires_grad_(True)
    z = ops.batch_norm_act(xd, bnd, actd)
    assert z.grad_fn is not None and "BatchNormActFn" in type(z.grad_fn).__name__      # the HIP op ran
    z.backward(gy.to(dev))
    x64 = x.double().requires_grad_(True)
    y64 = bn64(x64)
    z64 = torch.nn.functional.prelu(y64, act64.weight) if act64 is not None else y64
    z64.backward(gy.double())
    close = lambda a, b, tol: torch.allclose(a.detach().cpu().double(), b.detach(), rtol=tol, atol=tol)
    assert close(z, z64, 2e-5)
    assert close(xd.grad, x64.grad, 2e-4)
    assert close(bnd.weight.grad, bn64.weight.grad, 2e-4) and close(bnd.bias.grad, bn64.bias.grad, 2e-4)
    if act is not None:
        assert close(actd.weight.grad, act64.weight.grad, 2e-4)
    assert close(bnd.running_mean, bn64.running_mean, 1e-5) and close(bnd.running_var, bn64.running_var, 1e-5)
    assert int(bnd.num_batches_tracked) == int(bn64.num_batches_tracked) == 1
    # repeatable bit for bit
    bn2 = copy.deepcopy(bn).to(dev).train()
    x2 = x.to(dev).requires_grad_(True)
    z2 = ops.batch_norm_act(x2, bn2, copy.deepcopy(act).to(dev) if act is not None else None)
    z2.backward(gy.to(dev))
    assert torch.equal(z2, z) and torch.equal(x2.grad, xd.grad)
    # eval mode and tiny planes take the stock modules
    bnd.eval()
    ze = ops.batch_norm_act(x.to(dev), bnd, actd)
    assert "BatchNormActFn" not in type(ze.grad_fn).__name__ if ze.grad_fn is not None else True


@pytest.mark.parametrize("shape,with_scale", [((4, 64, 64, 64), True), ((3, 5, 33, 31), True), ((2, 16, 16, 16), False)])
def test_batch_norm_residual_act_kernels_match_torch(shape, with_scale):
    """smplr_bn_res_fwd/bwd - prelu(plane_scale * bn(x) + other) in one op - against the stock composition in
    float64 on the CPU with the same dropout factors: output, both input gradients, gamma / beta / slope gradients
    and running statistics."""
    import copy
    from ilps_amd import ops
    from ilps_amd.model import PReLU
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape) + 1)
    N, C = shape[0], shape[1]
    x, other, gy = (torch.randn(*shape, generator=g) for _ in range(3))
    scale = None
    if with_scale:
        scale = (torch.rand(N, C, generator=g) > 0.3).float() / 0.7
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.1)
    act = PReLU(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        act.weight.copy_(torch.rand(C, generator=g) * 0.6 - 0.2)          # some slopes negative
    bn64, act64 = copy.deepcopy(bn).double().train(), copy.deepcopy(act).double()
    bnd, actd = copy.deepcopy(bn).to(dev).train(), copy.deepcopy(act).to(dev)
    xd, od = x.to(dev).requires_grad_(True), other.to(dev).requires_grad_(True)
    drop = torch.nn.Dropout2d(0.3).train()
    out = ops.batch_norm_residual_act(xd, bnd, drop if with_scale else None, od, actd,
                                      plane_scale=scale.to(dev) if with_scale else None)
    assert "BatchNormResActFn" in type(out.grad_fn).__name__
    out.backward(gy.to(dev))
    x64, o64 = x.double().requires_grad_(True), other.double().requires_grad_(True)
    y64 = bn64(x64)
    if with_scale:
        y64 = y64 * scale.double()[:, :, None, None]
    z64 = torch.nn.functional.prelu(y64 + o64, act64.weight)
    z64.backward(gy.double())
    close = lambda a, b, tol: torch.allclose(a.detach().cpu().double(), b.detach(), rtol=tol, atol=tol)
    assert close(out, z64, 2e-5)
    assert close(xd.grad, x64.grad, 2e-4) and close(od.grad, o64.grad, 2e-5)
    assert close(bnd.weight.grad, bn64.weight.grad, 2e-4) and close(bnd.bias.grad, bn64.bias.grad, 2e-4)
    assert close(actd.weight.grad, act64.weight.grad, 2e-4)
    assert close(bnd.running_mean, bn64.running_mean, 1e-5) and close(bnd.running_var, bn64.running_var, 1e-5)


def test_graphed_predictor_equals_eager(smpl_model):
    """inference.GraphedPredictor (the predict forward replayed from one HIP graph) returns what predict_batch does."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.inference import GraphedPredictor, predict_batch
    from ilps_amd.model import SMPLRegressor
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    reg = SMPLRegressor(48, "enet", True).to(dev)
    dec = SMPLDecoder(smpl_model, img_wh=48)
    a, b = torch.rand(2, 3, 256, 256, device=dev), torch.rand(2, 3, 256, 256, device=dev)
    gp = GraphedPredictor(reg, dec, a)
    for img in (b, a):
        got = {k: v.clone() for k, v in gp(img).items()}
        want = predict_batch(reg, dec, img)
        assert torch.allclose(got["smpl"], want["smpl"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(got["verts"], want["verts"], rtol=1e-4, atol=1e-5)
        assert got["seg_maps"].shape == want["seg_maps"].shape == (2, 48, 48)
    with pytest.raises(RuntimeError):
        gp(torch.rand(1, 3, 256, 256, device=dev))


def test_ddp_train_step_single_rank_rccl(smpl_model):
    """The data-parallel wrapper (training.SegTrainer(ddp=True): DistributedDataParallel over RCCL, train.py:205-210's
    multi_gpu_model) with a one-rank process group on this GPU: the hand-written autograd nodes (decoder, batch
    norm, PReLU, loss) run under DDP's hooks, the bucketed all-reduce executes, and two steps reduce nothing to NaN."""
    import os
    import torch.distributed as dist
    from ilps_amd.training import SegTrainer
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        tr = SegTrainer(smpl_model, output_wh=48, encoder_architecture="enet", use_IEF=True, device=dev, ddp=True,
                        with_silhouette=True)
        tr.smpl_model.train()
        B = 2
        images = torch.rand(B, 3, 256, 256, device=dev)
        labels = torch.randint(0, 32, (B, 48, 48), device=dev)
        sl = torch.nn.functional.one_hot(torch.randint(0, 2, (B, 48 * 48), device=dev), 2).float()
        losses = [float(tr.step(images, labels, sl)) for _ in range(2)]
        assert all(np.isfinite(losses))
        g = [p.grad for p in tr.smpl_model.parameters() if p.grad is not None]
        assert len(g) > 100 and all(bool(torch.isfinite(t).all()) for t in g)
    finally:
        dist.destroy_process_group()


def test_fit_loop_saves_and_resumes(smpl_model, tmp_path):
    """training.fit (train.py:221-315's trial loop) writes checkpoints under the reference's naming scheme, and
    SegTrainer.resume restores weights, Adam state and the trial counter exactly."""
    import itertools
    import os
    from ilps_amd.training import SegTrainer, fit, save_name
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = SegTrainer(smpl_model, output_wh=48, encoder_architecture="enet", use_IEF=True, device=dev)
    tr.smpl_model.train()
    B = 2
    images = torch.rand(B, 3, 256, 256, device=dev)
    labels = torch.randint(0, 32, (B, 48, 48), device=dev)
    name = lambda t_: save_name("up-s31", 48, True, 0.005, None, True, t_, encoder="enet")
    assert name(10) == "up-s31_48x48_enet_ief_scaledown0005_arms_weighted_2_bg_weighted_0point3_gamma2_multigpu_10.pt"
    seen, shapes = [], []

    def monitor_hook(t_, trainer):                        # train.py:245-262's monitor step through the trainer
        seen.append(t_)
        out = trainer.monitor(images)
        shapes.append({k: tuple(v.shape) for k, v in out.items()})
        assert trainer.smpl_model.training                # the hook leaves the regressor in train mode

    hist = fit(tr, itertools.repeat((images, labels)), trials=3, steps_per_trial=2, save_dir=str(tmp_path), save_every=2,
               name_fn=name, on_trial_end=monitor_hook)
    assert len(hist) == 3 and all(np.isfinite(hist)) and seen == [0, 2]
    assert shapes[0]["verts"] == (B, 6890, 3) and shapes[0]["projects"] == (B, 6890, 3) and shapes[0]["mask"] == (B, 6890)
    assert shapes[0]["seg"] == (B, 48, 48, 32) and shapes[0]["smpl"] == (B, 86)
    # (the train decoder writes losses only; its input here comes from the eval-mode monitor pass, which leaves the
    # batch-norm buffers - part of the checkpoint compared below - alone)
    assert "verts" not in tr.decoder(tr.monitor(images)["smpl"], labels)
    assert sorted(os.listdir(tmp_path)) == sorted([name(0), name(2)])
    want = {k: v.clone() for k, v in tr.smpl_model.state_dict().items()}
    step_count = [int(st["step"]) for st in tr.opt.state_dict()["state"].values()][:1]
    tr2 = SegTrainer(smpl_model, output_wh=48, encoder_architecture="enet", use_IEF=True, device=dev)
    nxt = tr2.resume(os.path.join(tmp_path, name(2)))
    assert nxt == 3
    got = tr2.smpl_model.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert [int(st["step"]) for st in tr2.opt.state_dict()["state"].values()][:1] == step_count
    with pytest.raises(RuntimeError):
        SegTrainer(smpl_model, output_wh=64, encoder_architecture="enet", use_IEF=True, device=dev).resume(
            os.path.join(tmp_path, name(2)))


def test_decoder_loss_debugging_fit(smpl_model):
    """decoder_loss_debugging.py's experiment: no encoder, a table of learnable SMPL parameters (`build_debug_model`)
    fitted by Adam through the decoder to target part maps with `categorical_focal_loss(gamma=5)`.  Starting from
    the mean pose the loss must fall and the rendered part maps must move towards the targets - the hand-written
    backward of every kernel on the path is what makes that happen."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    from ilps_amd.model import build_debug_model
    from ilps_amd.smpl_model import mean86
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, W = 4, 48
    segs_model, params_model, verts_model, projects_model = build_debug_model(n, smpl_model, W, 32)
    for m in (segs_model, params_model):
        m.to(dev)
    # targets: part maps of perturbed poses / shapes / cameras
    g = torch.Generator().manual_seed(3)
    xt = torch.tensor(np.tile(mean86(W), (n, 1)), dtype=torch.float32)
    xt[:, 4:76] += 0.25 * torch.randn(n, 72, generator=g)
    xt[:, 76:] += 0.8 * torch.randn(n, 10, generator=g)
    xt[:, 2:4] += 2.0 * torch.randn(n, 2, generator=g)
    with torch.no_grad():
        target = SMPLDecoder(smpl_model, img_wh=W)(xt.to(dev))["seg"].argmax(-1)            # (n, W, W)
    idx = torch.arange(n, device=dev).reshape(n, 1)
    loss_fn = softmax_focal_loss(5.0, False)                       # gamma = 5, as the script compiles it
    opt = torch.optim.Adam(params_model.parameters(), lr=0.02)
    dec = segs_model.decoder

    def render():
        return dec(params_model(idx))["seg"]

    def agreement():
        with torch.no_grad():
            pm = render().argmax(-1)
            fg = target > 0
            return float(((pm == target) & fg).sum()) / float(fg.sum())
    a0 = agreement()
    losses = []
    for _ in range(150):
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(target, render()).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    a1 = agreement()
    print("loss %.4f -> %.4f, foreground part agreement %.3f -> %.3f" % (losses[0], losses[-1], a0, a1))
    # (the softmax runs over scores in [0, 1], so the loss has a high floor: the reference's design, model.py:119-120)
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, (losses[0], losses[-1])
    assert a1 > a0 + 0.1, "foreground part agreement %.3f -> %.3f" % (a0, a1)
    assert verts_model(idx).shape == (n, 6890, 3) and projects_model(idx).shape == (n, 6890, 3)
    assert segs_model(idx).shape == (n, W * W, 32)


def test_stage2_models_silhouette_at_its_own_resolution(smpl_model):
    """train_stage2_silhouette.py:72-104: segs at 48x48 and silhouettes at 64x64 from one decoder pass; the
    silhouette equals projects_to_silhouette on the same projection, and both heads send gradient to the encoder."""
    from ilps_amd.keras_smpl.projects_to_silhouette import projects_to_silhouette
    from ilps_amd.model import SMPLRegressor, build_full_model_from_saved_model_stage2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    reg = SMPLRegressor(48, "enet", True).to(dev).eval()
    verts_m, proj_m, silh_m, segs_m = build_full_model_from_saved_model_stage2(reg, 48, 64, smpl_model, 2)
    img = torch.rand(2, 3, 256, 256, device=dev)
    sil, seg, pr = silh_m(img), segs_m(img), proj_m(img)
    assert sil.shape == (2, 64 * 64, 2) and seg.shape == (2, 48 * 48, 32)
    want = torch.softmax(projects_to_silhouette(pr.detach(), 64).reshape(2, -1, 2), -1)
    assert torch.allclose(sil, want, atol=1e-6)
    (sil[..., 1].sum() + seg[..., 5].sum()).backward()
    g = reg.IEF_layer_3.weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0


# ------------------------------------------------------------------ two-rank data parallelism on one GPU
def _ddp2_worker(rank, world, port, q):
    """One rank of a 2-process SegTrainer(ddp=True) step; both ranks share cuda:0 (a one-GPU box), the
    process group runs over gloo (RCCL needs one device per rank)."""
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np
    import torch
    import torch.distributed as dist
    import ilps_amd  # noqa: F401
    from ilps_amd.smpl_model import synthetic_smpl_model
    from ilps_amd.training import SegTrainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = synthetic_smpl_model(1234)
        G, per = 4, 2
        g = torch.Generator().manual_seed(7)
        images = torch.rand(G, 3, 256, 256, generator=g).to(dev)
        labels = torch.randint(0, 32, (G, 48, 48), generator=g).to(dev)

        def grads(ddp, lo, hi):
            torch.manual_seed(0)                                   # identical initial weights everywhere
            tr = SegTrainer(model, output_wh=48, encoder_architecture="enet", use_IEF=True, device=dev, ddp=ddp)
            # eval mode: batch norm uses its running statistics and dropout is off, so the loss is a mean of
            # per-image terms and the mean of the two ranks' gradients IS the full-batch gradient (in training mode
            # each tower normalises with its own batch statistics, as Keras' multi_gpu_model towers do)
            tr.smpl_model.eval()
            param = tr.net(images[lo:hi])
            out = tr.decoder(param)
            tr.loss_fn(labels[lo:hi], out["seg"]).mean().backward()
            return [p.grad.detach().clone() for p in tr.smpl_model.parameters() if p.grad is not None]

        lo = rank * per
        mine = grads(True, lo, lo + per)                           # DDP: gradients all-reduced (mean) in backward
        ok, worst = True, 0.0
        if rank == 0:
            full = grads(False, 0, G)                              # single process, the whole batch of 4
            assert len(full) == len(mine) and len(full) > 100
            for a, b in zip(mine, full):
                scale = float(b.abs().max()) + 1e-12
                err = float((a - b).abs().max()) / scale
                worst = max(worst, err)
            ok = worst <= 2e-3
        # every rank holds the same averaged gradient after the all-reduce
        chk = torch.tensor([float(sum(float(t.double().sum()) for t in mine))], dtype=torch.float64)
        both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, chk)
        same = abs(float(both[0]) - float(both[1])) <= 1e-9 * max(1.0, abs(float(both[0])))
        if rank == 0:
            q.put((bool(ok), float(worst), bool(same)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_equal_full_batch_gradient(smpl_model):
    """`multi_gpu_model` (train.py:205-210) replaced by DistributedDataParallel, with evidence on hardware: two
    spawned processes, B = 2 each, both on cuda:0 - the full train-step graph (ENet + IEF on stock torch ops, the
    HIP decoder and loss head) under DDP's bucketed all-reduce - give the encoder/regressor gradients of the
    single-process B = 4 batch."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        ok, worst, same = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert same, "the two ranks hold different gradients after the all-reduce"
    assert ok, "DDP gradient differs from the full-batch gradient: worst max-norm error %.3e" % worst
