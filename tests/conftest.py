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


@pytest.fixture(scope="session")
def smpl_model():
    from ilps_amd.smpl_model import synthetic_smpl_model
    return synthetic_smpl_model(1234)


@pytest.fixture(scope="session")
def part_tables():
    from ilps_amd.smpl_model import load_part_tables
    return {vs: load_part_tables(vs) for vs in (1, 2, 5)}
