"""Restricted SMPL-pkl loader: round trip through a pickle shaped like the real file (chumpy-wrapped
arrays, scipy sparse regressors, uint32 kintree), and refusal of anything else."""
import io
import pickle
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp


def _fake_chumpy():
    mod = types.ModuleType("chumpy")
    sub = types.ModuleType("chumpy.ch")

    class Ch(object):
        def __init__(self, x):
            self.x = x

        def __getstate__(self):
            return {"x": self.x, "dterms": ("x",)}

    Ch.__module__ = "chumpy.ch"
    Ch.__qualname__ = "Ch"
    sub.Ch = Ch
    mod.ch = sub
    return mod, sub, Ch


def test_load_smpl_pkl_roundtrip(tmp_path, smpl_model):
    from ilps_amd.smpl_pkl import load_smpl_pkl
    mod, sub, Ch = _fake_chumpy()
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        kin = np.zeros((2, 24), np.uint32)
        kin[0] = smpl_model.parents.astype(np.int64) % (2 ** 32)
        dd = {"v_template": Ch(smpl_model.v_template), "shapedirs": Ch(smpl_model.shapedirs),
              "posedirs": Ch(smpl_model.posedirs), "weights": Ch(smpl_model.weights),
              "J_regressor": sp.csc_matrix(smpl_model.J_regressor),
              "cocoplus_regressor": sp.csc_matrix(smpl_model.cocoplus_regressor), "kintree_table": kin}
        path = tmp_path / "smpl.pkl"
        path.write_bytes(pickle.dumps(dd, protocol=2))
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    m = load_smpl_pkl(str(path))
    assert np.array_equal(m.v_template, smpl_model.v_template)
    assert np.array_equal(m.posedirs, smpl_model.posedirs) and np.array_equal(m.weights, smpl_model.weights)
    assert np.allclose(m.J_regressor, smpl_model.J_regressor) and m.parents[0] == -1
    assert np.array_equal(m.parents[1:], smpl_model.parents[1:])


def test_load_smpl_pkl_refuses_other_globals(tmp_path):
    from ilps_amd.smpl_pkl import load_smpl_pkl
    path = tmp_path / "evil.pkl"
    path.write_bytes(pickle.dumps({"v_template": io.BytesIO}, protocol=2))
    with pytest.raises(pickle.UnpicklingError, match="not allowed"):
        load_smpl_pkl(str(path))
