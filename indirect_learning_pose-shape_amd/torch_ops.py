"""`torch.ops.smplraster.*`: the at::Tensor layer over the C ABI (csrc/torch_ops.cpp).

    from ilps_amd import torch_ops
    ops = torch_ops.load()                      # torch.ops.smplraster, after torch.ops.load_library(...)
    mask = ops.visibility(proj)                 # compute_mask.py:12-108
    seg, arg, rec = ops.seg_fwd(proj, mask, part_pos, part_off, 48)

One host call per op (outputs and workspaces allocated in C++, the current HIP stream, TORCH_CHECKed arguments) where
the ctypes binding (`_lib.py`) marshals two dozen arguments per launcher from Python; Meta kernels make the ops
traceable.  Not a fallback and not a second implementation: the same launchers of libsmplraster_hip.so run.
The autograd Functions of `ops.py` keep the ctypes binding (they reuse buffers and fuse calls in ways the single ops
do not express); `SMPLDecoder` uses `decoder_fwd` for gradient-free forwards (predict.py's path).
"""
from __future__ import annotations

import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmplraster_torch.so")

# op name -> schema, as csrc/torch_ops.cpp registers them (tests/test_abi.py compares with the loaded library)
SCHEMAS = {
    "abi_version": "smplraster::abi_version() -> int",
    "build_tag": "smplraster::build_tag() -> str",
    "visibility": "smplraster::visibility(Tensor proj, int grid_wh=64, bool ref_compat=True) -> Tensor",
    "project_fwd": "smplraster::project_fwd(Tensor verts, Tensor cam, int vertex_sampling=1) -> Tensor",
    "project_bwd": "smplraster::project_bwd(Tensor dproj, Tensor verts, Tensor cam, int vertex_sampling=1) -> (Tensor, Tensor)",
    "seg_fwd": "smplraster::seg_fwd(Tensor proj, Tensor mask, Tensor part_pos, Tensor part_off, int W) -> (Tensor, Tensor, Tensor)",
    "seg_bwd": "smplraster::seg_bwd(Tensor dseg, Tensor arg, Tensor rec, int VP, int P, int K, bool deterministic=False) -> Tensor",
    "silh_fwd": "smplraster::silh_fwd(Tensor proj, int W) -> (Tensor, Tensor)",
    "silh_bwd": "smplraster::silh_bwd(Tensor dsilh, Tensor silh, Tensor arg, Tensor proj, bool deterministic=False) -> Tensor",
    "smpl_fwd": "smplraster::smpl_fwd(Tensor x, Tensor[] consts, int num_cam=4) -> Tensor[]",
    "smpl_bwd": ("smplraster::smpl_bwd(Tensor? dverts, Tensor? dproj, Tensor? dJ_transformed, Tensor x, Tensor[] consts, "
                 "Tensor Rs, Tensor J, Tensor A, Tensor v_posed, int num_cam=4, int vertex_sampling=1) -> Tensor"),
    "decoder_fwd": ("smplraster::decoder_fwd(Tensor x, Tensor[] consts, Tensor part_pos, Tensor part_off, int W, "
                    "int grid_wh=64, bool ref_compat=True, int num_cam=4) -> Tensor[]"),
}

_ns = None


def source_tag():
    """sha256(csrc/torch_ops.cpp)[:16] + '-' + torch.__version__, as csrc/Makefile compiles it into build_tag(); None when
    the source is not beside the library."""
    import hashlib
    src = os.path.join(_HERE, "csrc", "torch_ops.cpp")
    if not os.path.exists(src):
        return None
    with open(src, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16] + "-" + torch.__version__


def available() -> bool:
    # (not beside another build of the C-ABI library: this layer links the product one)
    return os.path.exists(LIB_PATH) and os.environ.get("SMPLR_TORCH_OPS", "1") != "0" and not _lib.LIB_OVERRIDE


def load():
    """torch.ops.smplraster (loads libsmplraster_hip.so first - same build-id check as every other entry - then the
    torch layer that links it).  Raises when the library is missing: `make -C indirect_learning_pose-shape_amd/csrc`."""
    global _ns
    if _ns is not None:
        return _ns
    _lib.load()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("torch op library missing: %s not found (make -C indirect_learning_pose-shape_amd/csrc)" % LIB_PATH)
    torch.ops.load_library(LIB_PATH)
    ns = torch.ops.smplraster
    if int(ns.abi_version()) != _lib.ABI_VERSION:
        raise RuntimeError("libsmplraster_torch.so was linked against another ABI version of libsmplraster_hip.so")
    want = source_tag()
    if want is not None and ns.build_tag() != want:
        raise RuntimeError("libsmplraster_torch.so was built from another torch_ops.cpp or for another torch (library %s, "
                           "here %s): rebuild with `make -C indirect_learning_pose-shape_amd/csrc`" % (ns.build_tag(), want))
    _ns = ns
    return ns
