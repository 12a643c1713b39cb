"""Host-side logic that needs no GPU: fixtures, synthetic model, part tables, conditioning ops,
loud failure of the HIP ops on CPU tensors."""
import numpy as np
import pytest
import torch


def test_part_tables(part_tables):
    for vs, n in ((1, 6879), (2, 3438), (5, 1376)):
        ids, off = part_tables[vs]
        assert len(ids) == n and len(off) == 32 and off[0] == 0 and off[-1] == n
        assert len(np.unique(ids)) == n and np.all(ids % vs == 0) and ids.max() < 6890
        sizes = np.diff(off)
        assert sizes.min() > 0
    assert np.diff(part_tables[1][1]).min() == 45 and np.diff(part_tables[1][1]).max() == 710


def test_device_part_table():
    from ilps_amd import ops
    from ilps_amd.smpl_model import load_part_tables
    for vs in (1, 2, 5):
        ids, off = load_part_tables(vs)
        pt = ops.build_part_table(ids, off, vs, 6890, "cpu")
        assert pt.P == 31 and pt.K == len(ids) and pt.VP == (6890 + vs - 1) // vs
        assert np.array_equal(pt.part_off.numpy(), off)
        assert np.array_equal(pt.part_pos.numpy(), ids // vs)        # projects_to_seg.py:36-37
        assert len(np.unique(pt.part_pos.numpy())) == pt.K            # still disjoint after // vs


def test_synthetic_model(smpl_model):
    m = smpl_model
    m.validate()
    assert np.allclose(m.weights.sum(1), 1) and (m.weights >= 0).all() and ((m.weights > 0).sum(1) <= 4).all()
    assert np.allclose(m.J_regressor.sum(1), 1) and (m.J_regressor >= 0).all()
    h = m.v_template[:, 1].max() - m.v_template[:, 1].min()
    assert 1.5 < h < 2.0
    from ilps_amd.smpl_model import synthetic_smpl_model
    m2 = synthetic_smpl_model(1234)
    assert np.array_equal(m.v_template, m2.v_template) and np.array_equal(m.posedirs, m2.posedirs)


def test_mean_params_and_conditioning_ops():
    from ilps_amd.smpl_model import load_mean_params, mean86
    from ilps_amd.keras_smpl.set_cam_params import set_cam_params, load_mean_set_cam_params
    from ilps_amd.keras_smpl.concat_mean_param import concat_mean_param
    from oracle import np_oracle as o
    pose, shape = load_mean_params()
    assert pose.shape == (72,) and shape.shape == (10,) and np.all(pose[:3] == 0)
    assert abs(shape[0] - 0.20560974) < 1e-7 and abs(pose[3] + 0.22387259) < 1e-7
    assert mean86(48)[:4].tolist() == [24, 24, 24, 30]
    rng = np.random.default_rng(0)
    s = rng.normal(0, 1, (3, 86)).astype(np.float32)
    for W in (48, 64):
        assert np.allclose(set_cam_params(torch.tensor(s), W).numpy(), o.set_cam_params(s, W), atol=1e-6)
        assert np.allclose(load_mean_set_cam_params(torch.tensor(s), W).numpy(),
                           o.load_mean_set_cam_params(s, W, pose, shape), atol=1e-6)
    f = rng.normal(0, 1, (3, 2048)).astype(np.float32)
    got = concat_mean_param(torch.tensor(f), 48).numpy()
    assert got.shape == (3, 2134)
    assert np.allclose(got, o.concat_mean_param(f, 48, pose, shape), atol=1e-6)


def test_smpl_constants_layout(smpl_model):
    from ilps_amd import ops
    c = ops.SMPLConstants.from_model(smpl_model, "cpu")
    assert c.blend.shape == (220, 20670) and torch.all(c.blend[217:] == 0)
    S = smpl_model.shapedirs.reshape(-1, 10).T
    P = smpl_model.posedirs.reshape(-1, 207).T
    assert np.allclose(c.blend[:10].numpy(), S, atol=1e-7) and np.allclose(c.blend[10:217].numpy(), P, atol=1e-7)
    # J = J_template + J_dirs.beta equals regressing the shaped vertices (batch_smpl.py:106-115)
    beta = np.random.default_rng(1).normal(0, 1, 10)
    v_shaped = (beta @ S).reshape(-1, 3) + smpl_model.v_template
    J = smpl_model.J_regressor @ v_shaped
    J2 = c.J_template.numpy().astype(np.float64) + c.J_dirs.numpy().astype(np.float64) @ beta
    assert np.abs(J - J2).max() < 1e-6


def test_hip_ops_refuse_cpu_tensors(smpl_model):
    """There is no CPU fallback: CPU tensors must raise, not silently compute."""
    from ilps_amd.keras_smpl.batch_smpl import SMPLLayer
    from ilps_amd.keras_smpl.compute_mask import compute_mask
    from ilps_amd.keras_smpl.projection import orthographic_project
    from ilps_amd.keras_smpl.projects_to_seg import projects_to_seg
    from ilps_amd.keras_smpl.projects_to_silhouette import projects_to_silhouette
    layer = SMPLLayer(smpl_model)
    assert layer.compute_output_shape((7, 86)) == (7, 6890, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        layer(torch.zeros(1, 86))
    with pytest.raises(RuntimeError, match="HIP device"):
        orthographic_project([torch.zeros(1, 6890, 3), torch.zeros(1, 86)], None)
    with pytest.raises(RuntimeError, match="HIP device"):
        compute_mask(torch.zeros(1, 6890, 3))
    with pytest.raises(RuntimeError):
        projects_to_seg([torch.zeros(1, 6890, 3), torch.ones(1, 6890)], 48)
    with pytest.raises(RuntimeError, match="HIP device"):
        projects_to_silhouette(torch.zeros(1, 6890, 3), 48)
    with pytest.raises(RuntimeError, match="shape"):
        layer(torch.zeros(1, 85))


def test_shard_range():
    from ilps_amd.sharding import shard_range
    for B, world in ((1024, 8), (10, 4), (3, 8), (0, 2)):
        spans = [shard_range(B, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_batch_norm_dispatch_falls_back_to_stock_modules_on_cpu():
    """ops.batch_norm_act / batch_norm_residual_act route to the HIP kernels only for training-mode fp32 HIP
    tensors; on the CPU (and in eval mode) they are exactly the stock modules' composition."""
    import copy
    import torch
    from ilps_amd import ops
    from ilps_amd.model import PReLU
    torch.manual_seed(0)
    x, other = torch.randn(3, 5, 20, 20), torch.randn(3, 5, 20, 20)
    bn, act = torch.nn.BatchNorm2d(5, eps=1e-3, momentum=0.1).train(), PReLU(5)
    bn2 = copy.deepcopy(bn)
    want = act(bn2(x))
    got = ops.batch_norm_act(x, bn, act)
    assert torch.equal(got, want) and torch.equal(bn.running_mean, bn2.running_mean)
    assert int(bn.num_batches_tracked) == 1
    drop = torch.nn.Dropout2d(0.0)
    want2 = act(drop(bn2(x)) + other)
    got2 = ops.batch_norm_residual_act(x, bn, drop, other, act)
    assert torch.equal(got2, want2)
    bn.eval(); bn2.eval()
    assert torch.equal(ops.batch_norm_act(x, bn), bn2(x))


def test_blend_gemm_mode_env(monkeypatch):
    import pytest
    from ilps_amd import ops
    monkeypatch.delenv("SMPLR_BLEND_GEMM", raising=False)
    assert ops.blend_gemm_mode() == "bf16x3"
    monkeypatch.setenv("SMPLR_BLEND_GEMM", "f32")
    assert ops.blend_gemm_mode() == "f32"
    monkeypatch.setenv("SMPLR_BLEND_GEMM", "fp16")
    with pytest.raises(RuntimeError):
        ops.blend_gemm_mode()


def test_part_tables_match_the_reference_coloured_template():
    """The reference ships the 31-part partition twice: as vertex-id lists (keras_smpl/part_vertices.pkl, what
    projects_to_seg.py:18-24 reads -> data/part_tables.npz) and as per-vertex colours of template-bodyparts.ply
    (-> tests/golden/ply_vertex_colour_class.npz, colour classes only).  They must describe the same partition:
    one colour per part, and the 11 vertices that belong to no part share a 32nd colour (grey)."""
    import os
    from ilps_amd.smpl_model import load_part_tables
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply_vertex_colour_class.npz"))
    cls, colours = g["vertex_class"].astype(np.int64), g["colours"]
    assert cls.shape == (6890,) and colours.shape == (32, 3)
    ids, off = load_part_tables(1)
    part = np.full(6890, -1)
    for p in range(31):
        part[ids[off[p]:off[p + 1]]] = p
    assert int((part < 0).sum()) == 11
    seen = {}
    for p in range(-1, 31):
        c = np.unique(cls[part == p])
        assert len(c) == 1, "part %d carries %d colours" % (p, len(c))
        seen[p] = int(c[0])
    assert len(set(seen.values())) == 32                       # a bijection between parts (+ unassigned) and colours
    assert tuple(colours[seen[-1]]) == (191, 191, 191)
    # the sampled tables are the same partition restricted to every 2nd / 5th vertex
    for vs in (2, 5):
        ids_s, off_s = load_part_tables(vs)
        for p in range(31):
            assert np.all(part[ids_s[off_s[p]:off_s[p + 1]]] == p)


def test_part_table_rejects_repeated_vertices():
    """One record slot per vertex in the backward: a table that lists a position twice (directly, or after the
    division by vertex_sampling) is refused up front instead of producing a wrong gradient."""
    import torch
    from ilps_amd import ops
    ok = ops.build_part_table([0, 2, 4, 6, 8], [0, 2, 5], 2, 10, torch.device("cpu"))
    assert ok.P == 2 and ok.K == 5 and ok.VP == 5
    with pytest.raises(ValueError):
        ops.build_part_table([0, 1, 2, 1], [0, 2, 4], None, 10, torch.device("cpu"))
    with pytest.raises(ValueError):
        ops.build_part_table([0, 1, 4], [0, 1, 3], 2, 10, torch.device("cpu"))      # 0 // 2 == 1 // 2


def test_decoder_options_are_validated_without_a_gpu():
    """SMPLDecoder(heads=, outputs=, loss=): argument errors surface at construction, before any kernel is touched."""
    import pytest
    from ilps_amd.decoder import SMPLDecoder
    from ilps_amd.focal_loss import softmax_focal_loss
    d = SMPLDecoder(None, img_wh=48)
    assert d.heads == ("seg",) and d.outputs == ("verts", "projects", "mask") and d.loss is None
    assert SMPLDecoder(None, with_silhouette=True).heads == ("seg", "silhouette")
    s = SMPLDecoder(None, heads=("silhouette",), outputs=())
    assert s.with_silhouette and s.outputs == ()
    lf = softmax_focal_loss(2.0, True)
    assert lf.gamma == 2.0 and lf.weight_classes is True
    assert SMPLDecoder(None, loss=lf, outputs=()).loss is lf
    for bad in (dict(heads=()), dict(heads=("seg", "depth")), dict(outputs=("seg",)),
                dict(heads=("silhouette",), loss=lf), dict(loss=lambda a, b: a),
                dict(heads=("silhouette",), vertex_sampling=5),
                dict(grid_wh=0), dict(grid_wh=-3), dict(grid_wh=129), dict(grid_wh=0, loss=lf)):
        with pytest.raises(ValueError):
            SMPLDecoder(None, **bad)
    # (a silhouette-only decoder computes no visibility mask: its grid_wh is not looked at)
    assert SMPLDecoder(None, heads=("silhouette",), grid_wh=0).grid_wh == 0
    # the HIP path has no CPU fallback: a CPU tensor is refused, not silently computed elsewhere
    import torch
    with pytest.raises(RuntimeError):
        d(torch.zeros(2, 86))
