"""GPU parity at the sizes BASELINE.json quotes (one-GPU legs):

  configs[1]  batch_smpl fwd+bwd only, B = 256
  configs[3]  end-to-end train step (ENet + IEF + decoder + focal loss + Adam), 48x48, B = 256
  configs[4]  decoder with BOTH heads (31-part segmentation + silhouette), 128 meshes per GPU

The float64 oracle is run on sampled rows (every op of the path is independent per mesh); the other rows
are pinned by row independence: the big batch must reproduce the rows computed 32 at a time, bit for bit
in the forward.  Plus the intermediates (Rs, J, A) and the reference class's exported helpers
(batch_smpl.py:168, :230, :255) against the oracle's."""
import numpy as np
import pytest
import torch

from _inputs import make_x
from test_gpu_parity import VERT_ATOL, SEG_RTOL, SEG_ATOL, dev, grad_close, t

pytestmark = pytest.mark.gpu

ROWS_256 = [0, 31, 32, 127, 128, 255]


def test_config1_batch_smpl_B256(smpl_model):
    """BASELINE configs[1]: BatchSMPLFn forward + backward at B = 256 (8 mesh tiles; the split-K backward's slice
    geometry at its largest tested size)."""
    from ilps_amd.keras_smpl.batch_smpl import SMPLLayer
    from oracle import np_oracle as o
    from oracle.torch_oracle import TorchSMPL
    B = 256
    x = make_x(B, 48, seed=256)
    rng = np.random.default_rng(256)
    gv = rng.normal(0, 1, (B, 6890, 3)).astype(np.float32)
    gj = rng.normal(0, 1, (B, 24, 3)).astype(np.float32)
    layer = SMPLLayer(smpl_model)
    xg = t(x).requires_grad_(True)
    verts = layer(xg)
    jt = layer.J_transformed
    ((verts * t(gv)).sum() + (jt * t(gj)).sum()).backward()
    torch.cuda.synchronize()
    assert verts.shape == (B, 6890, 3) and jt.shape == (B, 24, 3)
    # sampled rows against the float64 oracle: forward ...
    ref = o.smpl_layer_call(x[ROWS_256].astype(np.float64), smpl_model, return_all=True)
    assert np.abs(verts[ROWS_256].detach().cpu().numpy() - ref["verts"]).max() <= VERT_ATOL
    assert np.abs(jt[ROWS_256].detach().cpu().numpy() - ref["J_transformed"]).max() <= VERT_ATOL
    # ... and gradient (float64 autograd of the same rows with the same cotangents)
    xo = torch.tensor(x[ROWS_256], dtype=torch.float64, requires_grad=True)
    vo, jo, _ = TorchSMPL(smpl_model)(xo, return_all=True)
    ((vo * torch.tensor(gv[ROWS_256], dtype=torch.float64)).sum()
     + (jo * torch.tensor(gj[ROWS_256], dtype=torch.float64)).sum()).backward()
    got, want = xg.grad[ROWS_256].cpu().numpy(), xo.grad.numpy()
    assert np.all(got[:, :4] == 0)
    grad_close(got[:, 4:76], want[:, 4:76], name="dtheta B=256")
    grad_close(got[:, 76:], want[:, 76:], name="dbeta B=256")
    # every row: independent of the batch it was computed in (8 x 32)
    for lo in range(0, B, 32):
        xs = t(x[lo:lo + 32]).requires_grad_(True)
        v2 = layer(xs)
        j2 = layer.J_transformed
        ((v2 * t(gv[lo:lo + 32])).sum() + (j2 * t(gj[lo:lo + 32])).sum()).backward()
        assert torch.equal(v2, verts[lo:lo + 32]), "verts rows %d.." % lo
        assert torch.equal(j2, jt[lo:lo + 32]), "J_transformed rows %d.." % lo
        grad_close(xg.grad[lo:lo + 32].cpu().numpy(), xs.grad.cpu().numpy(), 1e-4, "dx rows %d.." % lo)
    assert torch.isfinite(xg.grad).all()


def test_smpl_intermediates_and_exported_helpers(smpl_model):
    """Rs, J, A of the pose kernel against np_oracle.smpl_layer_call(return_all=True), and the reference class's
    public helpers batch_rodrigues / batch_skew / batch_global_rigid_transformation (batch_smpl.py:255, :230, :168)
    against the oracle's restatements - batch_rodrigues with n = 48 and n = 50 rows (more than one 24-joint row of
    the pose kernel it runs on, and a ragged last row)."""
    from ilps_amd import ops
    from ilps_amd.keras_smpl.batch_smpl import SMPLLayer
    from oracle import np_oracle as o
    B = 7
    x = make_x(B, 48, seed=77)
    layer = SMPLLayer(smpl_model)
    c = layer.constants(t(x).device)
    _, Rs, J, A, Jt = ops._pose_fwd(t(x), 4, c)
    ref = o.smpl_layer_call(x.astype(np.float64), smpl_model, return_all=True)
    assert np.abs(Rs.cpu().numpy().reshape(B, 24, 3, 3) - ref["Rs"]).max() <= 2e-6
    assert np.abs(J.cpu().numpy() - ref["J"]).max() <= 2e-6
    assert np.abs(A.cpu().numpy().reshape(B, 24, 3, 4) - ref["A"][:, :, :3, :]).max() <= 1e-5
    assert np.abs(Jt.cpu().numpy() - ref["J_transformed"]).max() <= 1e-5
    rng = np.random.default_rng(4)
    for n in (48, 50, 1):
        th = rng.normal(0, 0.7, (n, 3)).astype(np.float32)
        got = layer.batch_rodrigues(t(th))
        assert got.shape == (n, 3, 3)
        assert np.abs(got.cpu().numpy() - o.batch_rodrigues(th.astype(np.float64))).max() <= 2e-6
    v = rng.normal(0, 1, (9, 3)).astype(np.float32)
    assert np.array_equal(SMPLLayer.batch_skew(t(v)).cpu().numpy(), o.batch_skew(v).astype(np.float32))
    Rs64, J64 = ref["Rs"], ref["J"]
    nj, A4 = SMPLLayer.batch_global_rigid_transformation(t(Rs64), t(J64), smpl_model.parents)
    wj, wA = o.batch_global_rigid_transformation(Rs64, J64, np.asarray(smpl_model.parents))
    assert nj.shape == (B, 24, 3) and A4.shape == (B, 24, 4, 4)
    assert np.abs(nj.cpu().numpy() - wj).max() <= 1e-5 and np.abs(A4.cpu().numpy() - wA).max() <= 1e-5
    nj, A4 = SMPLLayer.batch_global_rigid_transformation(t(Rs64), t(J64), smpl_model.parents, rotate_base=True)
    wj, wA = o.batch_global_rigid_transformation(Rs64, J64, np.asarray(smpl_model.parents), rotate_base=True)
    assert np.abs(nj.cpu().numpy() - wj).max() <= 1e-5 and np.abs(A4.cpu().numpy() - wA).max() <= 1e-5


def test_config4_decoder_seg_and_silhouette_B128(smpl_model, part_tables):
    """BASELINE configs[4], one-GPU leg: 128 meshes through the decoder with both heads, forward + backward.
    Rows {0, 63, 127} against the float64 oracle (verts, seg, silhouette, dx with the HIP mask injected: visibility
    is discrete), all rows against the same rows run 32 at a time."""
    from ilps_amd.decoder import SMPLDecoder
    from oracle import np_oracle as o
    from oracle import torch_oracle as to
    W, B = 48, 128
    x = make_x(B, W, seed=4128)
    rng = np.random.default_rng(8)
    gs = rng.normal(0, 1, (B, W, W, 32)).astype(np.float32)
    gl = rng.normal(0, 1, (B, W, W, 2)).astype(np.float32)
    dec = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True)
    xg = t(x).requires_grad_(True)
    out = dec(xg)
    ((out["seg"] * t(gs)).sum() + (out["silhouette"] * t(gl)).sum()).backward()
    torch.cuda.synchronize()
    assert out["seg"].shape == (B, W, W, 32) and out["silhouette"].shape == (B, W, W, 2)
    rows = [0, 63, 127]
    ids, off = part_tables[1]
    proj = out["projects"][rows].detach().cpu().numpy().astype(np.float64)
    mask = out["mask"][rows].cpu().numpy()
    assert np.array_equal(mask, o.compute_mask(proj))
    ref_v = o.smpl_layer_call(x[rows].astype(np.float64), smpl_model)
    assert np.abs(out["verts"][rows].detach().cpu().numpy() - ref_v).max() <= VERT_ATOL
    want_seg = o.projects_to_seg(proj, mask, W, ids, off)
    got_seg = out["seg"][rows].detach().cpu().numpy()
    assert np.all(np.abs(got_seg - want_seg) <= SEG_RTOL * np.abs(want_seg) + SEG_ATOL)
    want_sil = o.projects_to_silhouette(proj, W)
    got_sil = out["silhouette"][rows].detach().cpu().numpy()
    assert np.all(np.abs(got_sil - want_sil) <= SEG_RTOL * np.abs(want_sil) + SEG_ATOL)
    xo = torch.tensor(x[rows], dtype=torch.float64, requires_grad=True)
    mo = torch.tensor(mask, dtype=torch.float64)
    _, po, _, so = to.decoder_forward(to.TorchSMPL(smpl_model), xo, lambda p: mo, W, ids, off)
    ((so * torch.tensor(gs[rows], dtype=torch.float64)).sum()
     + (to.projects_to_silhouette(po, W) * torch.tensor(gl[rows], dtype=torch.float64)).sum()).backward()
    got, want = xg.grad[rows].cpu().numpy(), xo.grad.numpy()
    for sl, name in ((slice(0, 4), "dcam"), (slice(4, 76), "dtheta"), (slice(76, 86), "dbeta")):
        grad_close(got[:, sl], want[:, sl], 5e-3, name + " (seg+silh, B=128)", elem_atol=2e-3)
    for lo in range(0, B, 32):
        xs = t(x[lo:lo + 32]).requires_grad_(True)
        o2 = dec(xs)
        ((o2["seg"] * t(gs[lo:lo + 32])).sum() + (o2["silhouette"] * t(gl[lo:lo + 32])).sum()).backward()
        for k in ("verts", "projects", "mask", "seg", "silhouette"):
            assert torch.equal(o2[k], out[k][lo:lo + 32]), "%s rows %d.." % (k, lo)
        grad_close(xg.grad[lo:lo + 32].cpu().numpy(), xs.grad.cpu().numpy(), 1e-4, "dx rows %d.." % lo)
    assert torch.isfinite(xg.grad).all()


def test_config3_train_step_B256(smpl_model):
    """BASELINE configs[3]: SegTrainer.step at B = 256 (ENet on 256x256 inputs + IEF + decoder at 48x48 + softmax-
    focal loss + Adam).  Finite loss and gradients, the decoder's outputs on the regressed parameters equal to the
    same rows run 32 at a time (and to the oracle's vertices on sampled rows), and the loss on a fixed batch drops
    over 3 steps."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.smpl_model import mean86
    from ilps_amd.training import SegTrainer
    from oracle import np_oracle as o
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    B, W = 256, 48
    tr = SegTrainer(smpl_model, output_wh=W, encoder_architecture="enet", use_IEF=True, lr=1e-4, device=dev)
    tr.smpl_model.train()
    images = torch.rand(B, 3, 256, 256, device=dev)
    xl = torch.tensor(np.tile(mean86(W), (B, 1)), dtype=torch.float32, device=dev)
    xl[:, 4:76] += 0.1 * torch.randn(B, 72, device=dev)
    dec = SMPLDecoder(smpl_model, img_wh=W)
    with torch.no_grad():
        labels = dec(xl)["seg"].argmax(-1)
    losses = [float(tr.step(images, labels)) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    grads = [p.grad for p in tr.smpl_model.parameters() if p.grad is not None]
    assert len(grads) > 100 and all(bool(torch.isfinite(g).all()) for g in grads)
    assert float(tr.smpl_model.backbone.enet.init_conv.weight.grad.abs().sum()) > 0
    assert losses[-1] < losses[0], losses
    tr.smpl_model.eval()
    with torch.no_grad():
        param = tr.smpl_model(images)
        full = dec(param)                 # (the trainer's own decoder writes no verts / mask out: outputs=())
        for lo in range(0, B, 32):
            part = dec(param[lo:lo + 32].contiguous())
            for k in ("verts", "mask", "seg"):
                assert torch.equal(part[k], full[k][lo:lo + 32]), "%s rows %d.." % (k, lo)
    ref = o.smpl_layer_call(param[ROWS_256].cpu().numpy().astype(np.float64), smpl_model)
    assert np.abs(full["verts"][ROWS_256].cpu().numpy() - ref).max() <= VERT_ATOL


def test_config4_train_step_with_silhouette_B128(smpl_model):
    """BASELINE configs[4]'s per-GPU train step: SegTrainer(with_silhouette=True).step at 128 images per GPU
    (ENet on 256x256 inputs + IEF + decoder with BOTH heads at 48x48 + softmax-focal loss + silhouette
    cross-entropy + Adam; train_stage2_silhouette.py:226-234).  Finite losses and gradients, and the loss on a
    fixed batch drops over 3 steps."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.smpl_model import mean86
    from ilps_amd.training import SegTrainer
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    B, W = 128, 48
    tr = SegTrainer(smpl_model, output_wh=W, encoder_architecture="enet", use_IEF=True, lr=1e-4, device=dev,
                    with_silhouette=True)
    tr.smpl_model.train()
    images = torch.rand(B, 3, 256, 256, device=dev)
    xl = torch.tensor(np.tile(mean86(W), (B, 1)), dtype=torch.float32, device=dev)
    xl[:, 4:76] += 0.1 * torch.randn(B, 72, device=dev)
    with torch.no_grad():
        tgt = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True)(xl)
        labels, silh_labels = tgt["seg"].argmax(-1), tgt["silhouette"].argmax(-1)
    assert silh_labels.shape == (B, W, W) and int(silh_labels.sum()) > 0
    losses = [float(tr.step(images, labels, silh_labels)) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    grads = [p.grad for p in tr.smpl_model.parameters() if p.grad is not None]
    assert len(grads) > 100 and all(bool(torch.isfinite(g).all()) for g in grads)
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("W", [112, 128, 160])
def test_decoder_at_large_raster_sizes(smpl_model, part_tables, W):
    """The skinning form of the binning kernel keeps the mesh's (u, v) in LDS beside the pixel counters: it fits up
    to W = 112 at V = 6890 (smplr_skin_vis_seg_fits); beyond, the decoder takes the two-call path (smplr_skin_fwd
    + smplr_vis_seg_fwd) by itself.  Either way: vertices, mask and every score against the float64 oracle."""
    from ilps_amd import _lib
    from ilps_amd.decoder import SMPLDecoder
    from oracle import np_oracle as o
    assert bool(_lib.load().smplr_skin_vis_seg_fits(6890, W, 64)) == (W <= 112)
    B = 2
    x = make_x(B, W, seed=W)
    dev = torch.device("cuda:0")
    xg = torch.tensor(x, device=dev, requires_grad=True)
    out = SMPLDecoder(smpl_model, img_wh=W)(xg)
    out["seg"].square().sum().backward()
    assert bool(torch.isfinite(xg.grad).all()) and float(xg.grad.abs().sum()) > 0
    ref_v = o.smpl_layer_call(x.astype(np.float64), smpl_model)
    assert np.abs(out["verts"].detach().cpu().numpy() - ref_v).max() <= VERT_ATOL
    pj = out["projects"].detach().cpu().numpy().astype(np.float64)
    ref_m = o.compute_mask(pj)
    assert np.array_equal(ref_m, out["mask"].cpu().numpy())
    ids, off = part_tables[1]
    want = o.projects_to_seg(pj, ref_m, W, ids, off)
    got = out["seg"].detach().cpu().numpy()
    assert np.all(np.abs(got - want) <= 1e-3 * np.abs(want) + 1e-6)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second HIP device")
@pytest.mark.parametrize("blend_gemm", ["bf16x3", "f32"])
def test_ops_follow_their_operands_device(smpl_model, blend_gemm, monkeypatch):
    """Operands on cuda:1 while cuda:0 is the current device: the launches must go to the operands' device
    (every C-ABI call site enters a device guard), forward and backward - under both blend GEMM forms: each keeps a
    per-DEVICE record of the LDS attribute it set on its kernels (blend3.hip, and since round 4 blend.hip too)."""
    from ilps_amd.decoder import SMPLDecoder
    monkeypatch.setenv("SMPLR_BLEND_GEMM", blend_gemm)          # read when the constants are uploaded
    W, B = 48, 3
    x = make_x(B, W, seed=5)
    g = np.random.default_rng(5).normal(0, 1, (B, W, W, 32)).astype(np.float32)
    res = []
    for d in ("cuda:0", "cuda:1"):
        torch.cuda.set_device(0)
        dec = SMPLDecoder(smpl_model, img_wh=W)
        xg = torch.tensor(x, device=d, requires_grad=True)
        out = dec(xg)
        (out["seg"] * torch.tensor(g, device=d)).sum().backward()
        torch.cuda.synchronize(d)
        assert out["seg"].device == torch.device(d) and xg.grad.device == torch.device(d)
        res.append((out["verts"].detach().cpu(), out["seg"].detach().cpu(), xg.grad.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    grad_close(res[1][2].numpy(), res[0][2].numpy(), 1e-5, "dx cuda:1 vs cuda:0")


@pytest.mark.parametrize("blend_gemm", ["bf16x3", "f32"])
def test_lds_attribute_table_follows_the_device_ordinal(smpl_model, blend_gemm, monkeypatch):
    """The per-(kernel, device) table of raised LDS limits on a ONE-GPU box (the two-device test above
    is skipped here): a step sets each large-LDS kernel's attribute once, a second step sets nothing, and after
    smplr_debug_device_ordinal(1) - the library now files the current device under ordinal 1, as a second GPU's
    first launches would be - the same step sets every one of them again and computes the same bits."""
    from ilps_amd import _lib
    from ilps_amd.decoder import SMPLDecoder
    monkeypatch.setenv("SMPLR_BLEND_GEMM", blend_gemm)
    lib = _lib.load()
    W, B = 48, 3
    x = t(make_x(B, W, seed=5))
    g = t(np.random.default_rng(5).normal(0, 1, (B, W, W, 32)))
    gs = t(np.random.default_rng(6).normal(0, 1, (B, W, W, 2)))

    def step():
        dec = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True, deterministic=True)
        xg = x.clone().requires_grad_(True)
        out = dec(xg)
        torch.autograd.backward([out["seg"], out["silhouette"]], [g, gs])
        torch.cuda.synchronize()
        return out["seg"].detach().clone(), out["silhouette"].detach().clone(), xg.grad.clone()
    o1, o2 = (7, 8) if blend_gemm == "bf16x3" else (9, 10)     # ordinals nothing has run under yet
    prev = lib.smplr_debug_device_ordinal(o1)
    try:
        n0 = lib.smplr_debug_lds_attr_sets()
        a = step()
        n1 = lib.smplr_debug_lds_attr_sets()
        step()
        assert lib.smplr_debug_lds_attr_sets() == n1, "a second step on the same device set attributes again"
        assert n1 - n0 >= 3, "the step has at least three kernels above 48 KB of LDS (GEMM, binning, silhouette): %d" % (n1 - n0)
        lib.smplr_debug_device_ordinal(o2)
        b = step()
        assert lib.smplr_debug_lds_attr_sets() - n1 == n1 - n0, "a new device ordinal must set every attribute again"
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    finally:
        lib.smplr_debug_device_ordinal(prev)


@pytest.mark.parametrize("with_silh", [False, True])
def test_deterministic_backward_is_bit_reproducible_B128(smpl_model, with_silh):
    """SMPLDecoder(deterministic=True) at B = 128 (race row: the reference's ops are pure functions):
    five runs of the same step give torch.equal gradients - the rasterisers' backward accumulates in 64-bit fixed
    point instead of fp32 LDS atomics - and that gradient agrees with the default mode's to fp32 rounding."""
    from ilps_amd.decoder import SMPLDecoder
    W, B = 48, 128
    x = t(make_x(B, W, seed=2024))
    rng = np.random.default_rng(6)
    gs = t(rng.normal(0, 1, (B, W, W, 32)))
    gl = t(rng.normal(0, 1, (B, W, W, 2)))

    def run(dec):
        xg = x.clone().requires_grad_(True)
        out = dec(xg)
        loss = (out["seg"] * gs).sum()
        if with_silh:
            loss = loss + (out["silhouette"] * gl).sum()
        loss.backward()
        return xg.grad.clone()

    det = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=with_silh, deterministic=True)
    first = run(det)
    for _ in range(4):
        assert torch.equal(run(det), first)
    ref = run(SMPLDecoder(smpl_model, img_wh=W, with_silhouette=with_silh))
    grad_close(first.cpu().numpy(), ref.cpu().numpy(), 1e-5, "deterministic vs default dx")
    assert torch.isfinite(first).all()


def test_deterministic_rasteriser_ops_and_scaling(layer_inputs=None):
    """The stand-alone ops with deterministic=True: bit-identical dproj across runs, equal to the default's to
    rounding, and invariant to the cotangent's magnitude (the fixed-point scale follows max|dseg|): dproj(c g) =
    c dproj(g) to fp32 rounding for c = 2^-60 ... 2^40."""
    from ilps_amd.keras_smpl.batch_smpl import SMPLLayer
    from ilps_amd.keras_smpl.compute_mask import compute_mask
    from ilps_amd.keras_smpl.projection import orthographic_project
    from ilps_amd.keras_smpl.projects_to_seg import projects_to_seg
    from ilps_amd.keras_smpl.projects_to_silhouette import projects_to_silhouette
    from ilps_amd.smpl_model import synthetic_smpl_model
    W, B = 48, 3
    layer = SMPLLayer(synthetic_smpl_model(1234))
    x = t(make_x(B, W, seed=33))
    proj = orthographic_project([layer(x), x], None).detach()
    mask = compute_mask(proj)
    rng = np.random.default_rng(2)
    g = t(rng.normal(0, 1, (B, W, W, 32)))
    gl = t(rng.normal(0, 1, (B, W, W, 2)))

    def dseg(c, det):
        p = proj.clone().requires_grad_(True)
        (projects_to_seg([p, mask], W, deterministic=det) * (g * c)).sum().backward()
        return p.grad

    def dsil(c, det):
        p = proj.clone().requires_grad_(True)
        (projects_to_silhouette(p, W, deterministic=det) * (gl * c)).sum().backward()
        return p.grad

    for fn, name in ((dseg, "seg"), (dsil, "silh")):
        a = fn(1.0, True)
        assert torch.equal(a, fn(1.0, True)) and torch.equal(a, fn(1.0, True))
        grad_close(a.cpu().numpy(), fn(1.0, False).cpu().numpy(), 1e-5, name + " det vs default", per_column=False)
        for c in (2.0 ** -60, 2.0 ** 40):
            b = fn(c, True)
            grad_close((b / c).cpu().numpy(), a.cpu().numpy(), 1e-6, "%s cotangent x %g" % (name, c), per_column=False)
        assert float(fn(0.0, True).abs().max()) == 0.0


def test_global_batch_1024_row_independence(smpl_model):
    """configs[4]'s GLOBAL batch (1 024 meshes) on one GPU: eight 128-mesh groups of the blend GEMM, the 8-row form
    of the segmentation backward (B >= 512), the pose kernel and the blend GEMM as two launches (ops.POSE_BLEND_SPLIT_B),
    one silhouette workgroup per mesh: every row must equal the same row computed in a batch of 32 (one launch for pose
    + blend; forward bit for bit, gradient to rounding), and the gradient must be finite."""
    from ilps_amd.decoder import SMPLDecoder
    W, B = 48, 1024
    x = make_x(B, W, seed=1024)
    rng = np.random.default_rng(10)
    dec = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True)
    gs = t(rng.normal(0, 1, (B, W, W, 32)).astype(np.float32))
    gl = t(rng.normal(0, 1, (B, W, W, 2)).astype(np.float32))
    xg = t(x).requires_grad_(True)
    out = dec(xg)
    ((out["seg"] * gs).sum() + (out["silhouette"] * gl).sum()).backward()
    assert torch.isfinite(xg.grad).all()
    for lo in (0, 480, 992):
        xs = t(x[lo:lo + 32]).requires_grad_(True)
        o2 = dec(xs)
        ((o2["seg"] * gs[lo:lo + 32]).sum() + (o2["silhouette"] * gl[lo:lo + 32]).sum()).backward()
        for k in ("verts", "mask", "seg", "silhouette"):
            assert torch.equal(o2[k], out[k][lo:lo + 32]), "%s rows %d.." % (k, lo)
        grad_close(xg.grad[lo:lo + 32].cpu().numpy(), xs.grad.cpu().numpy(), 1e-4, "dx rows %d.." % lo)


def test_pose_and_blend_in_one_launch_or_two(smpl_model, monkeypatch):
    """The decoder's forward runs pose + blend as one launch below ops.POSE_BLEND_SPLIT_B meshes and as two from there
    on, and skins inside the binning workgroups below ops.FUSE_SKIN_BELOW_B meshes (and again from ops.FUSE_SKIN_FROM_B on)
    and in a launch of its own in between: the same bits either way (thresholds moved under a small batch instead of running a large one)."""
    from ilps_amd import ops
    from ilps_amd.decoder import SMPLDecoder
    B, W = 9, 48
    x = t(make_x(B, W, seed=77))
    dec = SMPLDecoder(smpl_model, img_wh=W)
    with torch.no_grad():
        one = dec(x)
        monkeypatch.setattr(ops, "POSE_BLEND_SPLIT_B", 4)
        two = dec(x)
        monkeypatch.setattr(ops, "FUSE_SKIN_BELOW_B", 4)
        monkeypatch.setattr(ops, "FUSE_SKIN_FROM_B", 1 << 30)
        three = dec(x)
    for k in ("verts", "projects", "mask", "seg", "J_transformed"):
        assert torch.equal(one[k], two[k]) and torch.equal(one[k], three[k]), k


def test_decoder_step_replays_from_a_hip_graph(smpl_model):
    """The decoder's forward + backward captured once in a HIP graph (what bench.py times) and replayed on NEW
    parameter values written into the captured input buffer: outputs and gradient equal to the eager step's (the
    launchers never allocate, synchronise or read host state that a capture would freeze)."""
    from ilps_amd import ops
    W, B = 48, 16
    consts = ops.SMPLConstants.from_model(smpl_model, dev())
    pt = ops.get_part_table(1, dev(), consts.V)
    x_static = t(make_x(B, W, seed=1))
    g = t(np.random.default_rng(2).normal(0, 1, (B, W, W, 32)).astype(np.float32))

    def step(xin):
        xg = xin.detach().requires_grad_(True)
        verts, proj, mask, seg, silh, jt, _ls = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1)
        seg.backward(g)
        return verts, seg, xg.grad

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step(x_static)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gv, gseg, gdx = step(x_static)
    for seed in (5, 6):
        xn = t(make_x(B, W, seed=seed))
        x_static.copy_(xn)
        graph.replay()
        torch.cuda.synchronize()
        ev, eseg, edx = step(xn)
        assert torch.equal(gv, ev) and torch.equal(gseg, eseg)
        grad_close(gdx.cpu().numpy(), edx.cpu().numpy(), 1e-5, "graph replay dx (seed %d)" % seed)


def test_step_is_bit_reproducible_under_allocator_churn(smpl_model):
    """200 decoder steps (both heads, deterministic backward) at B = 128 with other batch sizes and freed NaN-filled
    buffers in between: every output and the gradient bit-identical to the first step's.  (A hand-over between two
    waves of the fused pose + blend kernel through an LDS counter passed every parity test and was wrong about once
    in 30 launches; this is the test that would have caught it.)"""
    from ilps_amd import ops
    W, B = 48, 128
    consts = ops.SMPLConstants.from_model(smpl_model, dev())
    pt = ops.get_part_table(1, dev(), consts.V)
    x = t(make_x(B, W, seed=77))
    rng = np.random.default_rng(3)
    g = t(rng.normal(0, 1, (B, W, W, 32)).astype(np.float32))
    gl = t(rng.normal(0, 1, (B, W, W, 2)).astype(np.float32))
    names = ["verts", "proj", "mask", "seg", "silh", "Jt", "loss", "dx"]

    def step(xin, gs, gsl):
        xg = xin.detach().requires_grad_(True)
        outs = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, True, 1, True)
        torch.autograd.backward([outs[3], outs[4]], [gs, gsl])
        return [o.detach().clone() for o in outs] + [xg.grad.clone()]

    ref = step(x, g, gl)
    for k in range(200):
        if k % 5 == 0:
            junk = [torch.full((1 << 22,), float("nan"), device=dev()) for _ in range(4)]
            del junk
        if k % 9 == 0:
            step(x[:32], g[:32], gl[:32])
        cur = step(x, g, gl)
        for n, a, b in zip(names, cur, ref):
            assert torch.equal(a, b), "run %d: %s differs at %d places" % (k, n, int((a != b).sum()))
