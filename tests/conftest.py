import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ilps_amd  # noqa: E402,F401  (alias of the hyphenated package directory)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The shared library is git-ignored: (re)build it in-tree when missing or older than its sources
    (hipcc cross-compiles gfx950 without a GPU).  Building the product is not a fallback: every op
    still fails loudly if the library cannot be loaded."""
    import glob
    import subprocess
    pkg = os.path.join(ROOT, "indirect_learning_pose-shape_amd")
    lib = os.path.join(pkg, "libsmplraster_hip.so")
    srcs = glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) + \
        [os.path.join(ROOT, "include", "smplraster.h")]
    stale = (not os.path.exists(lib)) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs)
    if stale:
        subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-j4"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)


_ensure_built()


@pytest.fixture(scope="session")
def smpl_model():
    from ilps_amd.smpl_model import synthetic_smpl_model
    return synthetic_smpl_model(1234)


@pytest.fixture(scope="session")
def part_tables():
    from ilps_amd.smpl_model import load_part_tables
    return {vs: load_part_tables(vs) for vs in (1, 2, 5)}
