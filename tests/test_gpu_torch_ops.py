"""torch.ops.smplraster.* (csrc/torch_ops.cpp) against the ctypes path of ops.py: the same launchers,
so every output must be bit-equal; plus the oracle for the pieces a caller would chain by hand."""
import numpy as np
import pytest
import torch

from _inputs import make_x

pytestmark = pytest.mark.gpu


def _setup(smpl_model, B, W, seed):
    from ilps_amd import ops, torch_ops
    dev = torch.device("cuda:0")
    c = ops.SMPLConstants.from_model(smpl_model, dev)
    pt = ops.get_part_table(1, dev, c.V)
    x = torch.tensor(make_x(B, W, seed), device=dev)
    return ops, torch_ops.load(), dev, c, pt, x


def test_decoder_fwd_op_equals_the_autograd_node(smpl_model):
    """decoder_fwd = model.py:108-118 in one host call: bit-equal to SMPLDecoder's training-path forward, and it IS what
    SMPLDecoder runs for a gradient-free forward."""
    from ilps_amd.decoder import SMPLDecoder
    ops, ns, dev, c, pt, x = _setup(smpl_model, 5, 48, 31)
    o = ns.decoder_fwd(x, c.as_list(), pt.part_pos, pt.part_off, 48)
    dec = SMPLDecoder(smpl_model, img_wh=48)
    ref = dec(x.clone().requires_grad_(True))                      # the autograd node (ctypes launch sequence)
    for k, t in (("verts", o[0]), ("projects", o[1]), ("mask", o[2]), ("seg", o[3]), ("J_transformed", o[4])):
        assert torch.equal(ref[k].detach(), t), k
    with torch.no_grad():
        fast = dec(x)                                              # the fast path
    assert all(torch.equal(fast[k], ref[k].detach()) for k in ("verts", "projects", "mask", "seg", "J_transformed"))
    with pytest.raises(RuntimeError):
        ns.decoder_fwd(x.cpu(), c.as_list(), pt.part_pos, pt.part_off, 48)
    with pytest.raises(RuntimeError):
        ns.decoder_fwd(x[:, :80].contiguous(), c.as_list(), pt.part_pos, pt.part_off, 48)   # x must be (B, 86)


def test_single_ops_equal_the_ctypes_calls(smpl_model, part_tables):
    from oracle import np_oracle as o_
    ops, ns, dev, c, pt, x = _setup(smpl_model, 3, 48, 77)
    verts, v_posed, A, Rs, J, Jt = ns.smpl_fwd(x, c.as_list())
    Rs2, J2, A2, Jt2, vp2 = ops._pose_blend_fwd(x, 4, c)
    verts2, proj2 = ops._skin_fwd(vp2, A2, c, cam=x)
    assert torch.equal(verts, verts2) and torch.equal(v_posed, vp2) and torch.equal(A, A2) and torch.equal(Jt, Jt2)
    want = o_.smpl_layer_call(x.cpu().numpy().astype(np.float64), smpl_model)
    assert np.abs(verts.cpu().numpy() - want).max() <= 1e-4         # the north star's vertex bar
    proj = ns.project_fwd(verts, x)
    assert torch.equal(proj, proj2)
    mask = ns.visibility(proj)
    assert torch.equal(mask, ops.visibility(proj))
    seg, arg, rec = ns.seg_fwd(proj, mask, pt.part_pos, pt.part_off, 48)
    seg2, arg2, rec2 = ops._seg_fwd(proj, mask, 48, pt)[:3]
    # (the slot ids of nearest-pixel-only records depend on the order of the binning kernel's atomics: compare vertices)
    assert torch.equal(seg, seg2) and torch.equal(ops.argmin_vertices(arg, rec), ops.argmin_vertices(arg2, rec2))
    g = torch.randn_like(seg)
    dproj = ns.seg_bwd(g, arg, rec, pt.VP, pt.P, pt.K, True)        # deterministic: bit-reproducible
    dproj2 = ops._seg_bwd(g, arg2, rec2, pt.VP, 48, pt, deterministic=True)
    assert torch.equal(dproj, dproj2)
    silh, sarg = ns.silh_fwd(proj, 48)
    silh2, sarg2 = ops._silh_fwd(proj, 48)
    assert torch.equal(silh, silh2) and torch.equal(sarg, sarg2)
    gs = torch.randn_like(silh)
    assert torch.equal(ns.silh_bwd(gs, silh, sarg, proj, True), ops._silh_bwd(gs, silh2, sarg2, proj, 48, True))
    dv, dcam = ns.project_bwd(dproj, verts, x)
    dx = ns.smpl_bwd(dv, None, None, x, c.as_list(), Rs, J, A, v_posed)
    dx2 = torch.empty_like(x)
    ops._smpl_bwd(x, 4, c, Rs2, J2, A2, vp2, dv, None, None, 1, out=dx2)
    assert torch.equal(dx, dx2) and bool(torch.isfinite(dx).all())


def test_mismatched_operands_raise_instead_of_faulting(smpl_model):
    """The ops hand caller-supplied sizes to the launchers, so every shape and device is checked first: a
    wrong `arg` / `rec` / `silh`, a part table of another size, a CPU constant - each a RuntimeError, never a launch."""
    ops, ns, dev, c, pt, x = _setup(smpl_model, 2, 48, 9)
    verts, v_posed, A, Rs, J, Jt = ns.smpl_fwd(x, c.as_list())
    proj = ns.project_fwd(verts, x)
    mask = ns.visibility(proj)
    seg, arg, rec = ns.seg_fwd(proj, mask, pt.part_pos, pt.part_off, 48)
    g = torch.randn_like(seg)
    bad = [
        lambda: ns.seg_bwd(g, arg[:, :24].contiguous(), rec, pt.VP, pt.P, pt.K, False),          # arg of another image size
        lambda: ns.seg_bwd(g, arg, rec[:, :100].contiguous(), pt.VP, pt.P, pt.K, False),          # rec shorter than S(P, K)
        lambda: ns.seg_bwd(g, arg, rec, pt.VP, pt.P, pt.K - 7, False),                            # K of another part table
        lambda: ns.seg_bwd(g, arg.cpu(), rec, pt.VP, pt.P, pt.K, False),
        lambda: ns.seg_fwd(proj, mask, pt.part_pos.cpu(), pt.part_off, 48),
        lambda: ns.project_bwd(proj, verts, x[:, :3].contiguous()),                               # cam narrower than 4
        lambda: ns.smpl_bwd(verts, None, None, x, c.as_list(), Rs[:1].contiguous(), J, A, v_posed),
        lambda: ns.smpl_bwd(verts, None, None, x, [t.cpu() if i == 6 else t for i, t in enumerate(c.as_list())], Rs, J, A, v_posed),
        lambda: ns.decoder_fwd(x, [t.cpu() if i == 7 else t for i, t in enumerate(c.as_list())], pt.part_pos, pt.part_off, 48),
    ]
    silh, sarg = ns.silh_fwd(proj, 48)
    bad += [lambda: ns.silh_bwd(torch.randn(2, 24, 24, 2, device=dev), silh, sarg, proj, False),
            lambda: ns.silh_bwd(torch.randn_like(silh), silh, sarg[:, :24].contiguous(), proj, False),
            lambda: ns.silh_bwd(torch.randn_like(silh), silh, sarg, proj.cpu(), False)]
    for i, f in enumerate(bad):
        with pytest.raises(RuntimeError):
            f()
            pytest.fail("case %d did not raise" % i)
    torch.cuda.synchronize()                                           # nothing was launched, nothing faulted
    assert torch.isfinite(ns.seg_bwd(g, arg, rec, pt.VP, pt.P, pt.K, False)).all()
