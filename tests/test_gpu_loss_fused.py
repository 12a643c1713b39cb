"""The loss head fused into the rasteriser (model.py:119-120 + focal_loss.py:10-46):
smplr_seg_raster_ex / smplr_skin_vis_seg_fwd_ex / smplr_seg_loss_bwd against the float64 oracle
(np_oracle.categorical_focal_loss on np_oracle.projects_to_seg; gradients: torch_oracle autograd) and against the
unfused HIP path (scores written out, smplr_focal_fwd/bwd, smplr_seg_bwd).

Tolerances: the per-pixel loss against the oracle evaluated on the HIP path's OWN scores 1e-4 relative (fp32
softmax + log against float64, as for the unfused head); end to end against the oracle's scores 2e-3 (scores carry
the 1e-3 relative bar of north_star); gradients as the rasteriser's other backward tests (grad_close 2e-3)."""
import numpy as np
import pytest
import torch

from _inputs import argmin_disagreements, make_x
from test_gpu_parity import dev, grad_close, t

pytestmark = pytest.mark.gpu

# (gamma, class weights): the unfused head's label-map parametrisations (tests/test_gpu_parity.py) + two more
PARAMS = [(2.0, True), (2.0, False), (1.5, True), (0.0, False), (1.0, True), (0.0, True)]


@pytest.fixture(scope="module")
def layer(smpl_model):
    from ilps_amd.keras_smpl.batch_smpl import SMPLLayer
    return SMPLLayer(smpl_model)


def _proj_mask(layer, B, W, seed):
    from ilps_amd.keras_smpl.compute_mask import compute_mask
    from ilps_amd.keras_smpl.projection import orthographic_project
    x = t(make_x(B, W, seed=seed))
    with torch.no_grad():
        proj = orthographic_project([layer(x), x], None).contiguous()
        mask = compute_mask(proj)
    return proj, mask


def _labels(rng, B, W, seg=None):
    """Random labels, half of them replaced by the arg-max class where scores are given (a realistic map)."""
    lab = rng.integers(0, 32, (B, W, W))
    if seg is not None:
        am = seg.argmax(-1)
        pick = rng.random((B, W, W)) < 0.5
        lab = np.where(pick, am, lab)
    return lab.astype(np.int64)


@pytest.mark.parametrize("gamma,weighted", PARAMS)
@pytest.mark.parametrize("W", [48, 50])
def test_fused_loss_stage_parity(layer, part_tables, W, gamma, weighted):
    """Binning + rasteriser with the loss epilogue + the fused backward, through the raw stage calls (the C ABI)."""
    from ilps_amd import ops
    from oracle import np_oracle as o
    from oracle import torch_oracle as to
    B = 3
    proj, mask = _proj_mask(layer, B, W, seed=int(10 * gamma) + W)
    pt = ops.get_part_table(1, proj.device, 6890)
    ws, rec = ops._seg_bin(proj, mask, W, pt)                                # mask is an input here (grid_wh = 0)
    seg_ref, arg_ref = ops._seg_raster(ws, rec, B, W, pt)
    rng = np.random.default_rng(W + int(gamma * 7))
    lab = _labels(rng, B, W, seg_ref.cpu().numpy())
    w64 = o.FOCAL_CLASS_WEIGHTS[:32] if weighted else None
    cw = t(w64) if weighted else None
    seg = torch.empty_like(seg_ref)
    loss, stats, arg = ops._seg_raster_loss(ws, rec, B, W, pt, t(lab, torch.int32), cw, gamma, seg=seg)
    torch.cuda.synchronize()
    # the epilogue changes nothing of what the rasteriser writes
    assert torch.equal(seg, seg_ref) and torch.equal(arg, arg_ref)
    # a second launch without the score tensor gives the same loss and stats, bit for bit
    loss2, stats2, arg2 = ops._seg_raster_loss(ws, rec, B, W, pt, t(lab, torch.int32), cw, gamma, seg=None)
    assert torch.equal(loss2, loss) and torch.equal(stats2, stats) and torch.equal(arg2, arg)
    y = np.eye(32)[lab.reshape(B, -1)]
    # (a) the loss formula on the HIP path's own scores, in float64
    want_a = o.categorical_focal_loss(y, o.softmax_last(seg.cpu().numpy().astype(np.float64)), gamma, weighted)
    got = loss.cpu().numpy()
    assert got.shape == (B, W * W)
    assert np.all(np.abs(got - want_a) <= 1e-4 * np.abs(want_a) + 1e-7)
    # (b) end to end against the oracle's scores
    ids, off = part_tables[1]
    p64, m64 = proj.cpu().numpy().astype(np.float64), mask.cpu().numpy().astype(np.float64)
    want_b = o.categorical_focal_loss(y, o.softmax_last(o.projects_to_seg(p64, m64, W, ids, off)), gamma, weighted)
    assert np.all(np.abs(got - want_b) <= 2e-3 * np.abs(want_b) + 1e-6)
    # stats = (k / sum exp | k x the background's share | k = q_t softmax_t | label bits): against float64 on the HIP scores
    st = stats.cpu().numpy().astype(np.float64)
    assert np.array_equal(stats.cpu().numpy()[..., 3].view(np.int32).reshape(B, W, W), lab.astype(np.int32))
    s64_ = seg.cpu().numpy().astype(np.float64).reshape(B, W * W, 32)
    den = np.exp(s64_).sum(-1)
    smt = np.take_along_axis(np.exp(s64_), lab.reshape(B, -1, 1), 2)[..., 0] / den
    wt = (w64[lab.reshape(B, -1)] if weighted else 1.0)
    dpow = gamma * (1 - smt) ** (gamma - 1) if gamma > 0 else 0.0        # d/dx x^gamma at x = 1 - p
    q = wt * (dpow * np.log(smt) - (1 - smt) ** gamma / smt)               # d loss / d softmax_t (no clip in reach)
    k = q * smt
    assert np.allclose(st[..., 2], k, rtol=2e-5, atol=1e-9) and np.allclose(st[..., 0], k / den, rtol=2e-5, atol=1e-10)
    gate = (arg.cpu().numpy()[..., 0] == 1).reshape(B, -1)
    share = np.where(gate, (lab.reshape(B, -1) == 0) - np.exp(s64_[..., 0]) / den, 0.0)
    assert np.allclose(st[..., 1], k * share, rtol=2e-5, atol=1e-9)
    # backward: dproj against float64 autograd through projects_to_seg + softmax + focal loss
    cot = rng.normal(0, 1, (B, W * W))
    dproj = ops._seg_loss_bwd(t(cot), stats, arg, rec, 6890, W, pt, merge=True)
    po = torch.tensor(p64, requires_grad=True)
    s64 = to.projects_to_seg(po, torch.tensor(m64), W, ids, off)
    l64 = to.softmax_focal_loss(s64, torch.tensor(y), gamma, torch.tensor(w64) if weighted else None)
    (l64 * torch.tensor(cot)).sum().backward()
    # (where fp32 and float64 pick different arg-min vertices - near-ties, each one verified to be one - the gradient
    # of that pixel goes to another vertex: those vertices are compared by the max-norm bar only; here one pixel can
    # carry a visible share of a vertex' sum, the labelled channel's gradient being ~30 x any other channel's)
    ws64, warg = o.projects_to_seg(p64, m64, W, ids, off, return_argmin=True)
    garg = ops.argmin_vertices(arg, rec).cpu().numpy()
    ndiff, bad = argmin_disagreements(garg, warg, ws64, p64, m64, W)
    assert bad == 0 and ndiff <= 64, (ndiff, bad)
    got_d, want_d = dproj.cpu().numpy(), po.grad.numpy()
    assert np.abs(got_d - want_d).max() <= 2e-3 * np.abs(want_d).max()
    tied = np.zeros((B, 6890), bool)
    for n, ro, c, p in np.argwhere((garg != warg) & (ws64[..., 1:] >= 1e-30)):
        tied[n, warg[n, ro, c, p]] = True
        if garg[n, ro, c, p] >= 0:
            tied[n, garg[n, ro, c, p]] = True
    keep = ~tied[..., None] & np.ones((1, 1, 3), bool)
    grad_close(np.where(keep, got_d, 0.0), np.where(keep, want_d, 0.0), 2e-3, "dproj(fused loss, W=%d)" % W,
               per_column=False)
    # ... and against the unfused HIP path (scores -> smplr_focal_bwd -> smplr_seg_bwd)
    sd = seg_ref.clone().requires_grad_(True)
    lu = ops.SoftmaxFocalFn.apply(sd, t(lab.reshape(B, -1), torch.int64), cw, gamma)
    (lu * t(cot)).sum().backward()
    dproj_u = ops._seg_bwd(sd.grad.contiguous(), arg_ref, rec, 6890, W, pt, merge=True)
    assert np.all(np.abs(got - lu.detach().cpu().numpy()) <= 1e-5 * np.abs(want_a) + 1e-7)
    grad_close(dproj.cpu().numpy(), dproj_u.cpu().numpy(), 2e-4, "dproj fused vs unfused", per_column=False)


def test_fused_loss_kats(layer):
    """Labels outside the 32 classes contribute no loss and no gradient (as smplr_focal_fwd's one-hot of such an id);
    an all-background label map and an all-one-part label map against the float64 formula; deterministic = 1 gives
    the same gradient bit for bit on every launch and agrees with the default form."""
    from ilps_amd import ops
    from oracle import np_oracle as o
    B, W = 2, 48
    proj, mask = _proj_mask(layer, B, W, seed=5)
    pt = ops.get_part_table(1, proj.device, 6890)
    ws, rec = ops._seg_bin(proj, mask, W, pt)
    seg = torch.empty(B, W, W, 32, device=dev())
    for fill in (0, 7):
        lab = np.full((B, W, W), fill, np.int64)
        loss, stats, arg = ops._seg_raster_loss(ws, rec, B, W, pt, t(lab, torch.int32), None, 2.0, seg=seg)
        want = o.categorical_focal_loss(np.eye(32)[lab.reshape(B, -1)], o.softmax_last(seg.cpu().numpy().astype(np.float64)), 2.0)
        assert np.all(np.abs(loss.cpu().numpy() - want) <= 1e-4 * np.abs(want) + 1e-7)
    lab = np.random.default_rng(0).integers(0, 32, (B, W, W))
    lab[0, :10] = 32
    lab[1, 5:9] = -1
    lab[1, 20] = 1000
    loss, stats, arg = ops._seg_raster_loss(ws, rec, B, W, pt, t(lab, torch.int32), None, 2.0, seg=seg)
    bad = (lab < 0) | (lab > 31)
    got = loss.cpu().numpy().reshape(B, W, W)
    assert np.all(got[bad] == 0.0) and np.all(got[~bad] > 0.0)
    assert np.all(stats.cpu().numpy()[..., 2].reshape(B, W, W)[bad] == 0.0)          # q_t softmax_t = 0: no gradient
    cot = t(np.random.default_rng(1).normal(0, 1, (B, W * W)))
    d_all = ops._seg_loss_bwd(cot, stats, arg, rec, 6890, W, pt, merge=True)
    cz = cot.clone().reshape(B, W, W)
    cz[t(bad, torch.bool)] = 0.0                                   # zeroing their cotangents changes nothing
    d_zero = ops._seg_loss_bwd(cz.reshape(B, -1).contiguous(), stats, arg, rec, 6890, W, pt, merge=True)
    grad_close(d_all.cpu().numpy(), d_zero.cpu().numpy(), 1e-6, "bad labels carry no gradient", per_column=False)
    d1 = ops._seg_loss_bwd(cot, stats, arg, rec, 6890, W, pt, merge=True, deterministic=True)
    d2 = ops._seg_loss_bwd(cot, stats, arg, rec, 6890, W, pt, merge=True, deterministic=True)
    assert torch.equal(d1, d2)
    grad_close(d1.cpu().numpy(), d_all.cpu().numpy(), 1e-5, "deterministic vs default", per_column=False)


def test_fused_loss_argument_errors(layer):
    from ilps_amd import _lib, ops
    lib = _lib.load()
    # P != 31 or a negative gamma are refused before any launch
    one = torch.zeros(4, device=dev())
    rc = lib.smplr_seg_raster_ex(1, 48, 17, 100, _lib.ptr(one), _lib.ptr(one), _lib.ptr(one), None, 2.0, None,
                                 _lib.ptr(one), _lib.ptr(one), _lib.ptr(one), None, None)
    assert rc == -1 and b"32-class" in lib.smplr_last_error()
    rc = lib.smplr_seg_raster_ex(1, 48, 31, 6879, _lib.ptr(one), _lib.ptr(one), _lib.ptr(one), None, -1.0, None,
                                 _lib.ptr(one), _lib.ptr(one), _lib.ptr(one), None, None)
    assert rc == -1 and b"gamma" in lib.smplr_last_error()
    rc = lib.smplr_seg_loss_bwd(None, None, None, None, 1, 6890, 48, 31, 6879, None, None, 0, None)
    assert rc == -1 and b"dloss" in lib.smplr_last_error()
    assert lib.smplr_seg_loss_bwd(None, None, None, None, 0, 6890, 48, 31, 6879, None, None, 0, None) == 0


@pytest.mark.parametrize("B,W,with_silh,vs,scale", [(3, 48, False, None, 1.0), (128, 48, False, None, 1.0), (5, 48, True, None, 1.0),
                                                    (2, 64, False, None, 1.0), (2, 128, False, None, 1.0), (3, 48, False, 5, 1.0),
                                                    # a body 2.5 x larger in the image: record lists of 1 500+ (two to three
                                                    # chunks through the rasteriser's table, round 5), both block shapes
                                                    (4, 48, False, None, 2.5), (120, 48, False, None, 2.5)])
def test_decoder_with_fused_loss_equals_unfused(smpl_model, B, W, with_silh, vs, scale):
    """SMPLDecoder(loss=softmax_focal_loss(...)): forward(x, labels) returns the per-pixel loss without ever writing
    the scores; loss and dx agree with the unfused decoder + loss head.  W = 128 and vertex_sampling = 5 take the
    two-call path (the skinning form of the binning kernel does not apply there), B = 128 is BASELINE configs[2]'s batch."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    x = make_x(B, W, seed=B + W)
    x[:, 0:2] *= scale
    lf = softmax_focal_loss(2.0, True)
    plain = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=with_silh, vertex_sampling=vs)
    fused = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=with_silh, vertex_sampling=vs, loss=lf, outputs=(),
                        keep_seg=True)
    rng = np.random.default_rng(B)
    xa = t(x).requires_grad_(True)
    oa = plain(xa)
    lab = t(_labels(rng, B, W, oa["seg"].detach().cpu().numpy()), torch.int64)
    cot = t(rng.normal(0, 1, (B, W * W)) / (B * W * W))
    gl = t(rng.normal(0, 1, (B, W, W, 2)) / (B * W * W))
    la = lf(lab, oa["seg"])
    tot = (la * cot).sum()
    if with_silh:
        tot = tot + (oa["silhouette"] * gl).sum()
    tot.backward()
    xb = t(x).requires_grad_(True)
    ob = fused(xb, lab)
    assert set(ob) == {"J_transformed", "seg_loss", "seg"} | ({"silhouette"} if with_silh else set())
    tot = (ob["seg_loss"] * cot).sum()
    if with_silh:
        tot = tot + (ob["silhouette"] * gl).sum()
    tot.backward()
    torch.cuda.synchronize()
    assert torch.equal(ob["seg"], oa["seg"].detach())
    a, b = la.detach().cpu().numpy(), ob["seg_loss"].detach().cpu().numpy()
    assert np.all(np.abs(a - b) <= 1e-5 * np.abs(a) + 1e-7)
    grad_close(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), 1e-3, "dx fused vs unfused loss (B=%d, W=%d)" % (B, W))
    # without labels the same decoder returns the scores (the unfused path)
    with torch.no_grad():
        assert torch.equal(fused(t(x))["seg"], oa["seg"].detach())


def test_silhouette_only_decoder(smpl_model):
    """heads=("silhouette",): no mask, no binning, no 31-part rasteriser; silhouette and dx equal to the two-head
    decoder's silhouette branch."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    B, W = 7, 48
    x = make_x(B, W, seed=3)
    both = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True)
    only = SMPLDecoder(smpl_model, img_wh=W, heads=("silhouette",), outputs=("verts",)).share_constants(both)
    g = t(np.random.default_rng(4).normal(0, 1, (B, W, W, 2)))
    xa = t(x).requires_grad_(True)
    oa = both(xa)
    (oa["silhouette"] * g).sum().backward()
    xb = t(x).requires_grad_(True)
    ob = only(xb)
    assert set(ob) == {"J_transformed", "verts", "silhouette"}
    (ob["silhouette"] * g).sum().backward()
    assert torch.equal(ob["silhouette"], oa["silhouette"]) and torch.equal(ob["verts"], oa["verts"])
    grad_close(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), 1e-5, "dx silhouette-only vs two heads")
    with pytest.raises(ValueError):
        SMPLDecoder(smpl_model, heads=())
    with pytest.raises(ValueError):
        SMPLDecoder(smpl_model, heads=("silhouette",), loss=softmax_focal_loss())


def test_trainer_fused_loss_equals_unfused(smpl_model):
    """SegTrainer runs the loss inside the rasteriser for integer class maps: the first step's loss and the encoder
    gradients equal the unfused trainer's (same initial weights, eval-mode statistics so that both see the same
    network), and the silhouette-only step of the alternating schedule runs."""
    from ilps_amd.training import SegTrainer
    B, W = 4, 48
    torch.manual_seed(3)
    images = torch.rand(B, 3, 256, 256, device=dev())
    labels = torch.randint(0, 32, (B, W, W), device=dev())
    silh_labels = torch.randint(0, 2, (B, W, W), device=dev())
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        tr = SegTrainer(smpl_model, output_wh=W, encoder_architecture="enet", use_IEF=True, device=dev(),
                        with_silhouette=True, fused_loss=fused)
        tr.smpl_model.eval()
        loss = tr.step(images, labels, silh_labels)
        grads = [p.grad.detach().clone() for p in tr.smpl_model.parameters() if p.grad is not None]
        res.append((float(loss), grads))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-12
    l3 = tr.step(images, None, silh_labels)                        # silhouettes_model.fit's step
    assert np.isfinite(float(l3))


@pytest.mark.parametrize("B", [3, 128])
def test_silhouette_hint_changes_nothing(layer, B):
    """smplr_seg_raster_ex's vmax = each pixel's largest part score, handed to smplr_silh_fwd_hint as a bound of the
    distance to the nearest vertex: silhouette and arg-max vertex are bit for bit those of smplr_silh_fwd."""
    from ilps_amd import ops
    W = 48
    proj, mask = _proj_mask(layer, B, W, seed=40 + B)
    pt = ops.get_part_table(1, proj.device, 6890)
    ws, rec = ops._seg_bin(proj, mask, W, pt)
    seg = torch.empty(B, W, W, 32, device=dev())
    vmax = torch.empty(B, W, W, device=dev())
    ops._seg_raster_ex(ws, rec, B, W, pt, seg=seg, vmax=vmax)
    assert torch.equal(vmax, seg[..., 1:].max(-1).values)
    s0, a0 = ops._silh_fwd(proj, W)
    s1, a1 = ops._silh_fwd(proj, W, hint=vmax)
    assert torch.equal(s0, s1) and torch.equal(a0, a1)
    # a hint of zeros (no information) and a far-too-generous one (every vertex in reach) change nothing either
    s2, a2 = ops._silh_fwd(proj, W, hint=torch.zeros_like(vmax))
    s3, a3 = ops._silh_fwd(proj, W, hint=torch.full_like(vmax, 1e-30))
    assert torch.equal(s0, s2) and torch.equal(a0, a2) and torch.equal(s0, s3) and torch.equal(a0, a3)


def test_fused_loss_step_is_bit_reproducible_B128(smpl_model):
    """BASELINE configs[2]'s batch through the fused-loss decoder with deterministic=True: per-pixel loss, statistics
    and dx are bit for bit the same on every launch, also with other work and other batch sizes in between (the
    allocator hands out different addresses); and row independence: the loss of rows computed 32 at a time equals the
    full batch's rows bit for bit, their dx to rounding."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    B, W = 128, 48
    x = t(make_x(B, W, seed=9))
    dec = SMPLDecoder(smpl_model, img_wh=W, outputs=(), loss=softmax_focal_loss(2.0, True), deterministic=True)
    lab = torch.randint(0, 32, (B, W, W), device=dev())
    cot = t(np.random.default_rng(2).normal(0, 1, (B, W * W)) / (B * W * W))

    def step(xs, ls, cs):
        xg = xs.detach().requires_grad_(True)
        out = dec(xg, ls)
        (out["seg_loss"] * cs).sum().backward()
        return out["seg_loss"].detach().clone(), xg.grad.clone()

    ref = step(x, lab, cot)
    for k in range(12):
        if k % 3 == 0:
            junk = [torch.full((1 << 22,), float("nan"), device=dev()) for _ in range(3)]
            del junk
            step(x[:40], lab[:40], cot[:40])
        cur = step(x, lab, cot)
        assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), "launch %d differs" % k
    for lo in range(0, B, 32):
        part = step(x[lo:lo + 32], lab[lo:lo + 32], cot[lo:lo + 32])
        assert torch.equal(part[0], ref[0][lo:lo + 32])
        # (dx: the backward groups its partial sums by batch size - 24-row blocks at B = 128, 8-row blocks at B = 32 -
        # so the rows agree to rounding, not bit for bit)
        grad_close(part[1].cpu().numpy(), ref[1][lo:lo + 32].cpu().numpy(), 1e-5, "dx rows %d.." % lo)
