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
    """The shared library is git-ignored: (re)build it in-tree when it is missing or was NOT built from the
    sources beside it - decided by the build id compiled into it (sha256 of csrc/*.hip, csrc/*.h and the header),
    not by file times, which a copy to another box can equalise (hipcc cross-compiles gfx950 without a GPU).
    The id is read by `_lib.library_build_id()` in a child process, so this process never maps the stale file.
    Building the product is not a fallback: every op still fails loudly if the library cannot be loaded, and
    `_lib.load()` itself refuses a library whose id differs from its sources."""
    import subprocess
    from ilps_amd import _lib
    csrc = os.path.join(ROOT, "indirect_learning_pose-shape_amd", "csrc")

    # (read in a child process: a CDLL() here would pin the stale mapping and the load after the rebuild would get it back)
    if _lib.library_build_id() != _lib.source_build_id():
        subprocess.run(["make", "-C", csrc, "-B", "-j4", "PYTHON=" + sys.executable], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.STDOUT)


_ensure_built()


@pytest.fixture(scope="session")
def smpl_model():
    from ilps_amd.smpl_model import synthetic_smpl_model
    return synthetic_smpl_model(1234)


@pytest.fixture(scope="session")
def part_tables():
    from ilps_amd.smpl_model import load_part_tables
    return {vs: load_part_tables(vs) for vs in (1, 2, 5)}
