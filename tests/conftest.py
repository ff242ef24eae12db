import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (only present in the build container)")


@pytest.fixture(scope="session")
def hip_lib():
    """Path of the in-tree HIP library (built on demand; hipcc cross-compiles without a GPU)."""
    from crazyara_amd import build
    return build.build()


@pytest.fixture(scope="session")
def has_reference():
    return os.path.isdir("/root/reference/DeepCrazyhouse")
