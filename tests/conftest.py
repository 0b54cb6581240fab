import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zif4():
    from tests.helpers import read_extxyz
    return read_extxyz(os.path.join(GOLDEN, "ZIF-4.xyz"), 0)


@pytest.fixture(scope="session")
def hip_ctx():
    from amof_amd import _hip
    return _hip.get_context(0)
