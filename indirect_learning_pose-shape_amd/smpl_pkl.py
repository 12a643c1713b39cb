"""Loader for a user-supplied (licence-restricted) SMPL pickle: `neutral_smpl_with_cocoplus_reg.pkl`.

The reference opens it with Python-2 `cPickle` and unwraps chumpy arrays with `.r`
(`keras_smpl/batch_smpl.py:18-20, 34-35`).  Here the file is read with a RESTRICTED unpickler:
only numpy array reconstruction, scipy sparse matrices and chumpy `Ch` objects (replaced by a
stub that keeps their `x` array) are allowed; any other global raises.  No chumpy needed.
"""
from __future__ import annotations

import pickle

import numpy as np

from .smpl_model import SMPLModelData


class _ChStub:
    """Stands in for chumpy.ch.Ch / chumpy.reordering.*: keeps the state dict, exposes `.r`."""

    def __init__(self, *a, **k):
        self._state = {}

    def __setstate__(self, state):
        self._state = state if isinstance(state, dict) else {"x": state}

    @property
    def r(self):
        for key in ("x", "a", "_x"):
            if key in self._state:
                v = self._state[key]
                return np.asarray(v.r if isinstance(v, _ChStub) else v)
        raise ValueError("chumpy object without a value array (keys: %s)" % sorted(self._state))


_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
    ("scipy.sparse.csc", "csc_matrix"), ("scipy.sparse._csc", "csc_matrix"),
    ("scipy.sparse.csr", "csr_matrix"), ("scipy.sparse._csr", "csr_matrix"),
    ("scipy.sparse.coo", "coo_matrix"), ("scipy.sparse._coo", "coo_matrix"),
    ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"), ("__builtin__", "object"),
    ("builtins", "object"), ("_codecs", "encode"),       # bytes payloads of protocol-2 pickles
}


class _Restricted(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChStub
        if (module, name) in _ALLOWED:
            if module.startswith("scipy.sparse"):
                import scipy.sparse as sp
                return getattr(sp, name)
            if module in ("copy_reg", "copyreg"):
                import copyreg
                return copyreg._reconstructor
            if module in ("__builtin__", "builtins"):
                return object
            mod = module.replace("numpy.core", "numpy._core") if module.startswith("numpy.core") else module
            try:
                return getattr(__import__(mod, fromlist=[name]), name)
            except (ImportError, AttributeError):
                return getattr(__import__(module, fromlist=[name]), name)
        raise pickle.UnpicklingError("SMPL pkl: global %s.%s is not allowed" % (module, name))


def _arr(v):
    if isinstance(v, _ChStub):
        return np.asarray(v.r, np.float64)
    if hasattr(v, "todense"):
        return np.asarray(v.todense(), np.float64)
    return np.asarray(v)


def load_smpl_pkl(path: str) -> SMPLModelData:
    """Fields used by `SMPLLayer.build` (`batch_smpl.py:38-87`)."""
    with open(path, "rb") as f:
        dd = _Restricted(f, encoding="latin1").load()
    parents = np.asarray(dd["kintree_table"])[0].astype(np.int64)
    parents = np.where(parents > 1000, -1, parents).astype(np.int32)    # root: uint32 max (:71)
    coco = dd.get("cocoplus_regressor")
    m = SMPLModelData(
        v_template=_arr(dd["v_template"]).astype(np.float64),
        shapedirs=_arr(dd["shapedirs"]).astype(np.float64)[..., :10],
        posedirs=_arr(dd["posedirs"]).astype(np.float64),
        J_regressor=_arr(dd["J_regressor"]).astype(np.float64),
        weights=_arr(dd["weights"]).astype(np.float64),
        parents=parents,
        cocoplus_regressor=_arr(coco).astype(np.float64) if coco is not None else None)
    m.validate()
    return m
