"""Non-finite and degenerate parameter rows: a diverging regressor hands the decoder NaN / huge rows.
One batch of 8 through both heads, forward and backward:
  row 3  a NaN pose parameter            row 5  k_u = 1e30
  row 6  every vertex on ONE pixel (k_u = k_v = 0: a 6 890-way z-buffer collision, one pixel list of 6 877 records)
  row 7  every vertex outside the image and the visibility grid
must (a) not fault, (b) leave rows 0-2 and 4 bit-equal to the same rows run alone (no cell index, slot or list of a
bad row reaches a neighbour), (c) give rows 5-7 the float64 oracle's values (compute_mask.py:86-103 is the behaviour for
empty and colliding cells), and (d) give the NaN row what is STATED here: every vertex NaN, none visible but vertex 1
(compute_mask.py:99's artefact), part scores 0 / background 1 - where the reference's formulation (exp of NaN under
reduce_max) gives NaN scores: a NaN key never wins the rasteriser's minimum - a NaN silhouette (there the NaN key is
all a pixel has), and a NaN gradient row."""
import numpy as np
import pytest
import torch

from _inputs import make_x

pytestmark = pytest.mark.gpu
SEG_RTOL, SEG_ATOL = 1e-3, 1e-6


def _run(dec, x, dseg, dsilh):
    xg = x.clone().requires_grad_(True)
    out = dec(xg)
    torch.autograd.backward([out["seg"], out["silhouette"]], [dseg, dsilh])
    torch.cuda.synchronize()
    return {k: v.detach() for k, v in out.items()}, xg.grad.detach()


def test_bad_rows_do_not_reach_their_neighbours(smpl_model, part_tables):
    from ilps_amd.decoder import SMPLDecoder
    from oracle import np_oracle as o
    dev = torch.device("cuda", 0)
    W, B = 48, 8
    xn = make_x(B, W, 4242)
    xn[3, 10] = np.nan
    xn[5, 0] = 1e30
    xn[6, 0:4] = [0.0, 0.0, 24.0, 24.0]
    xn[7, 2:4] = [1e4, -1e4]
    x = torch.tensor(xn, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    dseg = torch.randn(B, W, W, 32, generator=g).to(dev)
    dsilh = torch.randn(B, W, W, 2, generator=g).to(dev)
    dec = SMPLDecoder(smpl_model, img_wh=W, with_silhouette=True, deterministic=True)
    out, dx = _run(dec, x, dseg, dsilh)
    good = [0, 1, 2, 4]
    out_g, dx_g = _run(dec, x[good].contiguous(), dseg[good].contiguous(), dsilh[good].contiguous())
    # (b) the good rows, bit for bit
    for k in ("verts", "projects", "mask", "seg", "silhouette", "J_transformed"):
        assert torch.equal(out[k][good], out_g[k]), "%s of the good rows changed beside bad rows" % k
    assert torch.equal(dx[good], dx_g), "dx of the good rows changed beside bad rows"
    assert torch.isfinite(dx[good]).all() and float(dx[good].abs().sum()) > 0
    # (c) rows 5, 6, 7 against the oracle
    ids, off = part_tables[1]
    proj = out["projects"].cpu().numpy().astype(np.float64)
    for r in (5, 6, 7):
        assert np.isfinite(proj[r]).all(), "row %d: projected vertices are finite inputs to the rasterisers" % r
        mask = o.compute_mask(proj[r:r + 1])
        assert np.array_equal(mask, out["mask"][r:r + 1].cpu().numpy()), "row %d: visibility mask" % r
        want = o.projects_to_seg(proj[r:r + 1], mask, W, ids, off)
        got = out["seg"][r:r + 1].cpu().numpy()
        assert np.all(np.abs(got - want) <= SEG_RTOL * np.abs(want) + SEG_ATOL), "row %d: seg" % r
        wsil = o.projects_to_silhouette(proj[r:r + 1], W)
        gsil = out["silhouette"][r:r + 1].cpu().numpy()
        assert np.all(np.abs(gsil - wsil) <= SEG_RTOL * np.abs(wsil) + SEG_ATOL), "row %d: silhouette" % r
        assert torch.isfinite(dx[r]).all(), "row %d: gradient" % r
    # row 6: one cell holds every vertex -> its front-most vertex and vertex 1 are visible, 31 parts score 1 there
    assert int((out["mask"][6] == 1.0).sum()) == 2
    assert float(out["seg"][6, W - 1 - 24, 24, 1:].min()) == 1.0 and float(out["seg"][6, W - 1 - 24, 24, 0]) == 0.0
    # row 7: nothing in the grid -> vertex 1 alone "visible", an empty image
    assert int((out["mask"][7] == 1.0).sum()) == 1 and float(out["mask"][7, 1]) == 1.0
    assert float(out["seg"][7, ..., 1:].abs().max()) == 0.0 and float(out["seg"][7, ..., 0].min()) == 1.0
    # (d) the NaN row, as stated
    assert torch.isnan(out["verts"][3]).all() and torch.isnan(out["projects"][3]).all()
    m3 = out["mask"][3]
    assert float(m3[1]) == 1.0 and int((m3 == 500.0).sum()) == m3.numel() - 1
    assert float(out["seg"][3, ..., 1:].abs().max()) == 0.0 and float(out["seg"][3, ..., 0].min()) == 1.0
    assert torch.isnan(out["silhouette"][3]).all()     # (the silhouette keeps the NaN, as the reference's exp(NaN) would)
    assert torch.isnan(dx[3, 4:]).all()


def test_bad_rows_through_the_fused_loss(smpl_model):
    """The same batch through the training path's decoder (loss head inside the rasteriser, no outputs written):
    finite loss and gradient for the good rows, equal to the rows run alone."""
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    dev = torch.device("cuda", 0)
    W, B = 48, 8
    xn = make_x(B, W, 4242)
    xn[3, 10] = np.nan
    xn[5, 0] = 1e30
    xn[6, 0:4] = [0.0, 0.0, 24.0, 24.0]
    xn[7, 2:4] = [1e4, -1e4]
    x = torch.tensor(xn, device=dev)
    labels = torch.randint(0, 32, (B, W, W), device=dev, dtype=torch.int32)
    dec = SMPLDecoder(smpl_model, img_wh=W, outputs=(), loss=softmax_focal_loss(2.0, True), deterministic=True)

    def run(xx, ll):
        xg = xx.clone().requires_grad_(True)
        ls = dec(xg, ll)["seg_loss"]
        ls.sum().backward()
        torch.cuda.synchronize()
        return ls.detach(), xg.grad.detach()
    ls, dx = run(x, labels)
    good = [0, 1, 2, 4]
    ls_g, dx_g = run(x[good].contiguous(), labels[good].contiguous())
    assert torch.equal(ls[good], ls_g) and torch.equal(dx[good], dx_g)
    assert torch.isfinite(ls[[0, 1, 2, 4, 5, 6, 7]]).all() and torch.isfinite(dx[[0, 1, 2, 4, 5, 6, 7]]).all()
    assert torch.isfinite(ls[3]).all()          # (scores 0 / 1 -> a finite loss; its gradient row is NaN through the NaN vertices)
