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


@pytest.fixture(scope="session")
def lds_poison():
    """ctypes handle of tests/support/lds_poison.hip: poison(pattern, base, lo, hi) fills every CU's LDS before the next launch."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "support", "lds_poison.hip")
    out = os.path.join(ROOT, "tests", "support", "_build", "liblds_poison.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        from crazyara_amd import build
        subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O2", *build.device_flags(), "-shared", "-fPIC", src, "-o", out], check=True, cwd="/tmp")
    lib = ctypes.CDLL(out)
    lib.poison_lds.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    lib.poison_lds.restype = ctypes.c_int
    return lib
