"""ctypes binding of libsmplraster_hip.so (the C ABI declared in include/smplraster.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C indirect_learning_pose-shape_amd/csrc`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMPLR_LIB_PATH: another build of the library (an A/B run against an older or a variant build, tools/ab_*.sh) - chosen
# by path, never copied over the product library; its build id is not compared with the sources beside it, and the torch
# op layer (linked against the product library) is switched off for the process.
LIB_OVERRIDE = os.environ.get("SMPLR_LIB_PATH") or None
LIB_PATH = os.path.abspath(LIB_OVERRIDE) if LIB_OVERRIDE else os.path.join(_HERE, "libsmplraster_hip.so")

KPAD = 220
CHUNK = 8
ABI_VERSION = 7          # SMPLR_ABI_VERSION of include/smplraster.h

P = c_void_p
I = c_int

# name -> (restype, argtypes); mirrors include/smplraster.h one to one.
SIGNATURES = {
    "smplr_abi_version": (c_int, []),
    "smplr_last_error": (c_char_p, []),
    "smplr_build_id": (c_char_p, []),
    "smplr_debug_device_ordinal": (c_int, [I]),
    "smplr_debug_lds_attr_sets": (c_int, []),
    "smplr_coef_ld": (c_int, [I]),
    "smplr_coef3_bytes": (c_size_t, [I]),
    "smplr_pose_fwd": (c_int, [P, I, I, I, P, P, P, P, P, P, P, P, P, P]),
    "smplr_pose_bwd": (c_int, [P, I, I, I, P, P, P, P, P, P, P, P, P, P, P]),
    "smplr_blend_fwd": (c_int, [P, P, P, I, I, P, P]),
    "smplr_blend_bwd_workspace": (c_size_t, [I, I]),
    "smplr_blend_bwd": (c_int, [P, P, I, I, P, P, P]),
    "smplr_coef3_pack": (c_int, [P, I, P, P]),
    "smplr_blend3_fwd_bytes": (c_size_t, [I]),
    "smplr_blend3_bwd_bytes": (c_size_t, [I]),
    "smplr_blend3_pack": (c_int, [P, I, P, P, P]),
    "smplr_blend3_fwd": (c_int, [P, P, P, I, I, P, P]),
    "smplr_pose_blend3_fwd": (c_int, [P, I, I, I, P, P, P, P, P, I, P, P, P, P, P, P]),
    "smplr_blend3_bwd_workspace": (c_size_t, [I, I]),
    "smplr_blend3_bwd": (c_int, [P, P, I, I, P, P, P]),
    "smplr_skin_fwd": (c_int, [P, P, P, P, P, I, I, I, I, P, P, P]),
    "smplr_skin_bwd_workspace": (c_size_t, [I, I]),
    "smplr_skin_bwd": (c_int, [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P]),
    "smplr_smpl_bwd_workspace": (c_size_t, [I, I]),
    "smplr_smpl_bwd": (c_int, [P, P, P, P, I, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "smplr_project_fwd": (c_int, [P, P, I, I, I, I, P, P]),
    "smplr_project_bwd": (c_int, [P, P, P, I, I, I, I, P, P, P]),
    "smplr_visibility": (c_int, [P, I, I, I, I, P, P]),
    "smplr_seg_workspace": (c_size_t, [I, I, I, I, I]),
    "smplr_seg_slots": (c_int, [I, I]),
    "smplr_seg_fwd": (c_int, [P, P, I, I, I, P, P, I, I, P, P, P, P, P, P]),
    "smplr_vis_seg_fwd": (c_int, [P, I, I, I, I, I, P, P, I, I, P, P, P, P, P, P, P]),
    "smplr_skin_vis_seg_fits": (c_int, [I, I, I]),
    "smplr_skin_vis_seg_fwd": (c_int, [P, P, P, P, I, I, I, I, I, I, P, P, I, I, P, P, P, P, P, P, P, P, P]),
    "smplr_seg_bin": (c_int, [P, P, I, I, I, I, I, P, P, I, I, P, P, P, P]),
    "smplr_seg_raster": (c_int, [I, I, I, I, P, P, P, P, P]),
    "smplr_seg_raster_timed": (c_int, [I, I, I, I, P, P, P, P, P, P]),
    "smplr_seg_raster_plan": (c_int, [I, I, I, I, P, P]),
    "smplr_seg_bwd_nsplit": (c_int, [I, I]),
    "smplr_seg_bwd_workspace": (c_size_t, [I, I]),
    "smplr_seg_bwd": (c_int, [P, P, P, I, I, I, I, I, P, P, I, P]),
    "smplr_seg_raster_ex": (c_int, [I, I, I, I, P, P, P, P, c_float, P, P, P, P, P, P]),
    "smplr_skin_vis_seg_fwd_ex": (c_int, [P, P, P, P, I, I, I, I, I, I, P, P, I, I, P, P, P, c_float, P, P, P, P, P, P,
                                          P, P, P, P, P]),
    "smplr_seg_loss_bwd": (c_int, [P, P, P, P, I, I, I, I, I, P, P, I, P]),
    "smplr_silh_workspace": (c_size_t, [I, I, I]),
    "smplr_silh_fwd": (c_int, [P, I, I, I, P, P, P, P]),
    "smplr_silh_fwd_hint": (c_int, [P, P, I, I, I, P, P, P, P]),
    "smplr_silh_bwd": (c_int, [P, P, P, P, I, I, I, P, I, P]),
    "smplr_focal_fwd": (c_int, [P, P, P, P, c_float, c_longlong, I, P, P, P]),
    "smplr_focal_bwd": (c_int, [P, P, P, P, c_float, P, c_longlong, I, P, P]),
    "smplr_prelu_fwd": (c_int, [P, P, c_longlong, I, I, P, P]),
    "smplr_prelu_bwd_workspace": (c_size_t, [c_longlong, I, I]),
    "smplr_prelu_bwd": (c_int, [P, P, P, c_longlong, I, I, P, P, P, P]),
    "smplr_bn_workspace": (c_size_t, [c_longlong, I, I]),
    "smplr_bn_fwd": (c_int, [P, P, P, P, c_longlong, I, I, c_float, c_float, P, P, P, P, P, P, P]),
    "smplr_bn_bwd": (c_int, [P, P, P, P, P, P, P, c_longlong, I, I, P, P, P, P, P, P]),
    "smplr_bn_res_fwd": (c_int, [P, P, P, P, P, P, c_longlong, I, I, c_float, c_float, P, P, P, P, P, P, P]),
    "smplr_bn_res_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_longlong, I, I, P, P, P, P, P, P, P]),
}

_lib = None


def source_build_id():
    """sha256 over csrc/*.hip, csrc/*.h (sorted by name) and include/smplraster.h, concatenated: what
    csrc/Makefile compiles into the library as smplr_build_id().  None when the sources are not beside the
    library (an installed copy without csrc/)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    header = os.path.join(os.path.dirname(_HERE), "include", "smplraster.h")
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    if not names or not os.path.exists(header):
        return None
    h = hashlib.sha256()
    for f in [os.path.join(csrc, n) for n in names] + [header]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def library_build_id(path=None):
    """The id compiled into the library FILE at `path`, read in a child process: a ctypes.CDLL() in this process
    would keep the file mapped, and glibc hands the same (old) mapping back for that path after a rebuild, so a
    probe-then-rebuild-then-load sequence (tests/conftest.py) must not open it here.  None when the file is
    missing, does not load, or has no smplr_build_id."""
    import subprocess
    import sys
    path = LIB_PATH if path is None else path
    if not os.path.exists(path):
        return None
    code = ("import ctypes,sys\n"
            "f = ctypes.CDLL(sys.argv[1]).smplr_build_id\n"
            "f.restype = ctypes.c_char_p\n"
            "sys.stdout.write(f().decode('ascii'))\n")
    try:
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.SubprocessError):
        return None
    return r.stdout.strip() if r.returncode == 0 and r.stdout.strip() else None


def build_id():
    """The id compiled into the loaded library."""
    return load().smplr_build_id().decode("ascii")


def load():
    """Load the HIP library (once) and bind every symbol of the header; fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "HIP extension missing: %s not found. Build it with "
            "`python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc, gfx950)." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if LIB_OVERRIDE and not hasattr(lib, name):
            continue                     # (an older build in an A/B run: entry points added since are simply absent)
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.smplr_abi_version() != ABI_VERSION and not LIB_OVERRIDE:
        raise RuntimeError("libsmplraster_hip.so ABI version mismatch")
    want, got = source_build_id(), lib.smplr_build_id().decode("ascii")
    if LIB_OVERRIDE:
        import sys
        sys.stderr.write("[smplraster] SMPLR_LIB_PATH=%s (build %s...): not the product library\n" % (LIB_PATH, got[:12]))
    elif want is not None and want != got:
        raise RuntimeError(
            "libsmplraster_hip.so was built from other sources than the ones beside it (library %s..., sources "
            "%s...): rebuild with `make -C indirect_learning_pose-shape_amd/csrc`" % (got[:12], want[:12]))
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().smplr_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    """torch's current stream on the CURRENT device: call sites run under `on_device`, which makes the
    operands' device the current one first (a launch on device 0's stream with device-1 pointers faults)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(fn):
    """Run fn with the device of its first HIP tensor argument as the current device, so that `stream()`,
    the kernels' launches and torch's allocations all refer to the device the operands live on (autograd
    switches devices for backward by itself, forward calls do not)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        for a in args:
            if getattr(a, "is_cuda", False) is True:          # a tensor, or an operand bundle exposing .device
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
        return fn(*args, **kw)
    return wrapper


def require_cuda(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """The HIP path only: fp32/int device tensors, contiguous.  No CPU fallback exists."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on a HIP device (got %s); this framework has no CPU path"
                           % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    return t.contiguous()
