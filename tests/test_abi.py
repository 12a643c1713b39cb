"""The C ABI: every function declared in include/smplraster.h is exported by the shared library
and bound by the ctypes table (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smplraster.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smplr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("smplr_pose_fwd", "smplr_pose_bwd", "smplr_blend_fwd", "smplr_blend_bwd", "smplr_skin_fwd",
                 "smplr_skin_bwd", "smplr_project_fwd", "smplr_project_bwd", "smplr_visibility",
                 "smplr_seg_fwd", "smplr_seg_bwd", "smplr_silh_fwd", "smplr_silh_bwd"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from ilps_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libsmplraster_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "symbol %s missing from the shared library" % name


def test_ctypes_table_matches_header():
    from ilps_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.smplr_abi_version() == _lib.ABI_VERSION == 7
    # argument errors are reported without touching the GPU
    rc = lib.smplr_visibility(None, 1, 6890, 0, 1, None, None)
    assert rc == -1 and b"grid_wh" in lib.smplr_last_error()
    assert lib.smplr_blend_bwd_workspace(128, 20670) > 0
    # the skinning + binning call refuses null operands, and an empty batch is a no-op
    rc = lib.smplr_skin_vis_seg_fwd(None, None, None, None, 86, 1, 6890, 48, 64, 1, None, None, 31, 6879, None, None, None,
                                    None, None, None, None, None, None)
    assert rc == -1 and b"smplr_skin_vis_seg_fwd" in lib.smplr_last_error()
    assert lib.smplr_skin_vis_seg_fwd(None, None, None, None, 86, 0, 6890, 48, 64, 1, None, None, 31, 6879, None, None,
                                      None, None, None, None, None, None, None) == 0
    assert lib.smplr_skin_bwd_workspace(128, 6890) == 128 * 27 * 292 * 4
    # the skinning form of the binning kernel needs the mesh's staged (u, v) in LDS beside the 64 x 64 z-buffer, the
    # pixel counters and the slot map: it fits the reference's raster sizes, not the largest ones the ABI accepts
    assert lib.smplr_skin_vis_seg_fits(6890, 48, 64) == 1 and lib.smplr_skin_vis_seg_fits(6890, 96, 64) == 1
    assert lib.smplr_skin_vis_seg_fits(6890, 128, 64) == 0 and lib.smplr_skin_vis_seg_fits(6890, 160, 64) == 0
    assert lib.smplr_skin_vis_seg_fits(7169, 48, 64) == 0 and lib.smplr_skin_vis_seg_fits(6890, 48, 0) == 0


def test_argument_counts_match_header():
    from ilps_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_res, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, src)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), "%s: header has %d parameters, ctypes table %d" % (name, n, len(args))


def test_seg_bwd_row_blocks_by_batch():
    """The segmentation backward cuts a mesh's image into row blocks of 8 rows, or of 24 from the batch on that gives
    every CU (256) a tall block (24 rows, or the largest even divisor of W from 12 to 24); the workspace is sized for whichever applies (host-side arithmetic only)."""
    from ilps_amd import _lib
    lib = _lib.load()
    assert lib.smplr_seg_bwd_nsplit(1, 48) == 6 and lib.smplr_seg_bwd_nsplit(127, 48) == 6
    assert lib.smplr_seg_bwd_nsplit(128, 48) == 2 and lib.smplr_seg_bwd_nsplit(511, 48) == 2
    assert lib.smplr_seg_bwd_nsplit(512, 48) == 2 and lib.smplr_seg_bwd_nsplit(2048, 48) == 2
    assert lib.smplr_seg_bwd_nsplit(64, 96) == 4 and lib.smplr_seg_bwd_nsplit(63, 96) == 12
    # the tall height divides the image where it can: W = 64 -> 16 rows (4 blocks per mesh), from 64 meshes on
    assert lib.smplr_seg_bwd_nsplit(128, 64) == 4 and lib.smplr_seg_bwd_nsplit(64, 64) == 4 and lib.smplr_seg_bwd_nsplit(63, 64) == 8
    assert lib.smplr_seg_bwd_nsplit(300, 20) == 1 and lib.smplr_seg_bwd_nsplit(0, 48) == 0
    for B, W in ((1, 48), (128, 48), (64, 96), (5, 50)):
        assert lib.smplr_seg_bwd_workspace(B, W) == B * lib.smplr_seg_bwd_nsplit(B, W) * 5 * 4096 * 2 * 4


def test_library_is_built_from_the_sources_beside_it(monkeypatch):
    """smplr_build_id() = sha256 over csrc/*.hip, csrc/*.h and the header as csrc/Makefile took it; `_lib.load()`
    recomputes it from the files and refuses a library built from anything else."""
    import subprocess
    from ilps_amd import _lib
    want = _lib.source_build_id()
    assert want is not None and len(want) == 64
    assert _lib.build_id() == want
    # the Makefile's recipe, spelled out independently: cat of the sorted sources + the header | sha256sum
    csrc = os.path.join(ROOT, "indirect_learning_pose-shape_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") or f.endswith(".h"))
    sh = subprocess.run("cat %s ../../include/smplraster.h | sha256sum" % " ".join(names), shell=True, cwd=csrc,
                        stdout=subprocess.PIPE, check=True).stdout.decode().split()[0]
    assert sh == want
    # a library whose id differs from the sources' is refused at load
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_build_id", lambda: "0" * 64)
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib.load()


def test_probing_a_stale_library_does_not_pin_its_mapping(tmp_path):
    """tests/conftest.py reads the built id, rebuilds when it is stale, then `_lib.load()` opens the file: the probe
    must not leave the old file mapped in this process (glibc would hand the same handle back for that path and the
    rebuilt library would never be seen).  Reproduced with a toy library built twice under one path."""
    import shutil
    import subprocess
    from ilps_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    so = str(tmp_path / "libtoy.so")

    def build(tag):
        src = tmp_path / ("toy_%s.c" % tag[:4])
        src.write_text('const char *smplr_build_id(void) { return "%s"; }\n' % tag)
        subprocess.run(["gcc", "-shared", "-fPIC", "-o", so + ".new", str(src)], check=True)
        os.replace(so + ".new", so)

    old, new = "a" * 64, "b" * 64
    build(old)
    assert _lib.library_build_id(so) == old           # the probe conftest makes (child process)
    build(new)                                        # "make -B"
    fn = ctypes.CDLL(so).smplr_build_id               # what _lib.load() does afterwards, in this process
    fn.restype = ctypes.c_char_p
    assert fn().decode("ascii") == new
    assert _lib.library_build_id(str(tmp_path / "missing.so")) is None
    (tmp_path / "junk.so").write_bytes(b"not an ELF file")
    assert _lib.library_build_id(str(tmp_path / "junk.so")) is None


def test_torch_library_registers_the_declared_ops():
    """The at::Tensor layer (TORCH_LIBRARY(smplraster, ...) over the extern "C" launchers): every op
    of `torch_ops.SCHEMAS` is registered with that schema, has a Meta kernel (output shapes without a device) and
    refuses CPU tensors - no compute, runs without a GPU."""
    import torch
    from ilps_amd import torch_ops
    ns = torch_ops.load()
    for name, schema in torch_ops.SCHEMAS.items():
        op = getattr(ns, name)
        assert str(op.default._schema) == schema, (name, str(op.default._schema))
    assert int(ns.abi_version()) == 7
    m = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="meta")
    proj = m(2, 6890, 3)
    assert ns.visibility(proj).shape == (2, 6890)
    assert ns.project_fwd(m(2, 6890, 3), m(2, 86), 5).shape == (2, 1378, 3)
    seg, arg, rec = ns.seg_fwd(proj, m(2, 6890), m(6879, dt=torch.int32), m(32, dt=torch.int32), 48)
    assert seg.shape == (2, 48, 48, 32) and arg.shape == (2, 48, 48, 32) and arg.dtype == torch.int16 and rec.shape[2] == 4
    assert ns.seg_bwd(m(2, 48, 48, 32), arg, rec, 6890, 31, 6879).shape == (2, 6890, 3)
    silh, sarg = ns.silh_fwd(proj, 64)
    assert silh.shape == (2, 64, 64, 2) and sarg.dtype == torch.int32
    consts = [m(24, 3), m(24, 3, 10), m(24, dt=torch.int32), m(20670), m(64, dt=torch.uint8), m(64, dt=torch.uint8),
              m(6890, 24), m(6890, 8), m(20670, 224)]
    outs = ns.decoder_fwd(m(2, 86), consts, m(6879, dt=torch.int32), m(32, dt=torch.int32), 48)
    assert [tuple(o.shape) for o in outs[:5]] == [(2, 6890, 3), (2, 6890, 3), (2, 6890), (2, 48, 48, 32), (2, 24, 3)]
    assert ns.smpl_bwd(m(2, 6890, 3), None, None, m(2, 86), consts, m(2, 24, 9), m(2, 24, 3), m(2, 24, 12),
                       m(2, 6890, 3)).shape == (2, 86)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ns.visibility(torch.zeros(1, 10, 3))                 # a CPU tensor: no kernel registered for it


def test_seg_raster_plan_is_host_arithmetic():
    """smplr_seg_raster_plan (ABI 7) launches nothing: the block shape by batch and the records one pass over a tile's LDS
    table takes, as raster2_fwd_kernel computes them - 128 pair-lanes x 8 part ranges with tables of 800 (tiles over 8
    image rows) / 1 032 records at W = 48 and 1 444 at W = 64, 64 x 10 with 1 228 for the small batches at W <= 48 with the
    full part table; a tile over more than 10 image rows (W = 12 in 64-lane blocks) cannot take the table form at all."""
    import ctypes
    from ilps_amd import _lib
    lib = _lib.load()

    def plan(B, W, K=6879, P=31):
        info = (ctypes.c_int32 * 8)()
        nt = lib.smplr_seg_raster_plan(B, W, P, K, info, None)
        tiles = (ctypes.c_int32 * max(nt, 1))()
        assert lib.smplr_seg_raster_plan(B, W, P, K, info, tiles) == nt
        return nt, list(info), list(tiles)[:nt]
    nt, info, tiles = plan(128, 48)
    assert nt == 9 and info[:4] == [128, 8, 9, 31 + 2 + 32] and tiles == [1032, 800, 1032] * 3
    assert info[4:7] == [1032, 800, 0]
    nt, info, tiles = plan(128, 64)
    assert nt == 16 and set(tiles) == {1444} and info[0] == 128
    nt, info, tiles = plan(8, 48)
    assert nt == 18 and info[:2] == [64, 10] and set(tiles) == {1228}
    nt, info, tiles = plan(128, 48, K=1376)                    # every fifth vertex: the large shape at every batch
    assert info[0] == 128 and nt == 9
    nt, info, tiles = plan(4, 12)
    assert 0 in tiles and info[6] == 1                         # some tile spans more than 10 image rows
    assert lib.smplr_seg_raster_plan(0, 48, 31, 6879, None, None) == 0
    assert lib.smplr_seg_raster_plan(4, 161, 31, 6879, None, None) == 0
    assert lib.smplr_seg_raster_plan(4, 48, 32, 6879, None, None) == 0


def test_debug_device_ordinal_round_trips():
    from ilps_amd import _lib
    lib = _lib.load()
    prev = lib.smplr_debug_device_ordinal(5)
    try:
        assert lib.smplr_debug_device_ordinal(70) == 5          # (clamped to the table: 63)
        assert lib.smplr_debug_device_ordinal(-1) == 63
        assert lib.smplr_debug_device_ordinal(-1) == -1
        assert lib.smplr_debug_lds_attr_sets() >= 0
    finally:
        lib.smplr_debug_device_ordinal(prev)
