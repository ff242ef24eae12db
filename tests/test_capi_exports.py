"""The C-ABI library loads and exports every symbol include/crazyara_hip.h declares (CPU only, no compute calls)."""
import os
import re

from crazyara_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    header = open(os.path.join(ROOT, "include", "crazyara_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)


def test_errors_are_loud_without_gpu(hip_lib):
    lib = _capi.load()
    assert lib.mi_version().startswith(b"crazyara_amd")
    if lib.mi_device_count() == 0:
        h = lib.mi_net_create(b"/nonexistent", 0, 8, b"float16")
        assert not h and len(_capi.last_error()) > 0
