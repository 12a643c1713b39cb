#!/usr/bin/env python3
"""Convert the reference's DATA files for the hot path into package fixtures.

Inputs (data only, read from the read-only reference mount; never source code):
  keras_smpl/part_vertices.pkl, 2_sampled_part_vertices.pkl, 5_sampled_part_vertices.pkl
      31 python lists of vertex ids (read by projects_to_seg.py:18-24)
  neutral_smpl_mean_params.h5
      'shape' float64[10] at byte 4192, 'pose' float64[72] at byte 4272
      (read through deepdish at set_cam_params.py:41-47, concat_mean_param.py:9-15)

  template-bodyparts.ply
      the template mesh coloured by body part (the renderer's asset, renderer.py): 6890 vertices with an RGB
      colour each.  Only the COLOUR CLASS of every vertex is kept (no geometry): an independent statement of
      the 31-part partition, used by tests/test_host_logic.py to pin the part tables.

Output: indirect_learning_pose-shape_amd/data/part_tables.npz, mean_params.npz,
        tests/golden/ply_vertex_colour_class.npz

The pickles are parsed with an unpickler that refuses every global (they hold
only INT/LIST opcodes), so nothing from the mount is ever executed.
"""
import io
import os
import pickle
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "indirect_learning_pose-shape_amd", "data")


class NoGlobals(pickle.Unpickler):
    def find_class(self, module, name):  # pragma: no cover - must never trigger
        raise pickle.UnpicklingError("global %s.%s refused" % (module, name))


def load_lists(path):
    with open(path, "rb") as f:
        parts = NoGlobals(io.BytesIO(f.read())).load()
    assert isinstance(parts, list) and len(parts) == 31
    return [np.asarray(p, dtype=np.int32) for p in parts]


def csr(parts):
    off = np.zeros(len(parts) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(p) for p in parts])
    return np.concatenate(parts).astype(np.int32), off


def main():
    os.makedirs(OUT, exist_ok=True)
    tables = {}
    for vs, name in ((1, "part_vertices.pkl"), (2, "2_sampled_part_vertices.pkl"),
                     (5, "5_sampled_part_vertices.pkl")):
        parts = load_lists(os.path.join(REF, "keras_smpl", name))
        ids, off = csr(parts)
        assert ids.min() >= 0 and ids.max() < 6890
        assert len(np.unique(ids)) == len(ids), "parts are disjoint"
        if vs > 1:
            assert np.all(ids % vs == 0)
        tables["ids_vs%d" % vs] = ids          # ORIGINAL vertex ids, part-major
        tables["off_vs%d" % vs] = off          # 32 offsets
        print(name, "ids", len(ids), "sizes", np.diff(off).min(), "..", np.diff(off).max())
    np.savez_compressed(os.path.join(OUT, "part_tables.npz"), **tables)

    with open(os.path.join(REF, "neutral_smpl_mean_params.h5"), "rb") as f:
        blob = f.read()
    assert len(blob) == 4848
    shape = np.frombuffer(blob[4192:4192 + 80], dtype="<f8").copy()
    pose = np.frombuffer(blob[4272:4272 + 576], dtype="<f8").copy()
    assert abs(shape[0] - 0.20560974) < 1e-7
    np.savez(os.path.join(OUT, "mean_params.npz"), shape=shape, pose=pose)
    print("mean shape[:3]", shape[:3], "pose[:6]", pose[:6])

    # template-bodyparts.ply: binary little-endian, vertex = 6 floats (position, normal) + 3 uchar (colour)
    with open(os.path.join(REF, "template-bodyparts.ply"), "rb") as f:
        raw = f.read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex 6890" in head and b"format binary_little_endian" in head
    dt = np.dtype([("pos", "<f4", 3), ("nrm", "<f4", 3), ("rgb", "u1", 3)])
    v = np.frombuffer(body[:6890 * dt.itemsize], dtype=dt)
    colours, cls = np.unique(v["rgb"], axis=0, return_inverse=True)
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    np.savez_compressed(os.path.join(golden, "ply_vertex_colour_class.npz"), colours=colours.astype(np.uint8),
                        vertex_class=cls.astype(np.uint8))
    print("ply: %d colour classes over %d vertices" % (len(colours), len(cls)))


if __name__ == "__main__":
    main()
