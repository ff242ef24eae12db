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


def test_header_is_plain_c99_and_struct_layouts_match_the_python_binding(hip_lib, tmp_path):
    """include/crazyara_hip.h compiled as C (not C++) by a translation unit outside the library, -std=c99 -Wall -Werror -pedantic;
    the struct sizes / offsets a C consumer sees equal those of the ctypes mirror (crazyara_amd/search.py)."""
    import ctypes as C
    import shutil
    import subprocess
    from crazyara_amd import search
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    so = os.path.join(str(tmp_path), "libhdrcheck.so")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_header_check.c"), "-o", so, _capi.LIB_PATH], check=True)
    chk = C.CDLL(so)
    for fn in ("cra_sizeof_search_settings", "cra_sizeof_search_stats", "cra_offsetof_settings_virtual_offset_strength",
               "cra_offsetof_settings_version_minor", "cra_offsetof_stats_depth_max", "cra_sizeof_selfplay_settings", "cra_sizeof_selfplay_stats",
               "cra_offsetof_selfplay_seed", "cra_offsetof_selfplay_stats_wins", "cra_sizeof_go_limits", "cra_offsetof_go_limits_move_overhead"):
        getattr(chk, fn).restype = C.c_size_t
    assert chk.cra_sizeof_search_settings() == C.sizeof(search.SearchSettingsC)
    assert chk.cra_sizeof_search_stats() == C.sizeof(search.SearchStatsC)
    assert chk.cra_offsetof_settings_virtual_offset_strength() == search.SearchSettingsC.virtual_offset_strength.offset
    assert chk.cra_offsetof_settings_version_minor() == search.SearchSettingsC.version_minor.offset
    assert chk.cra_offsetof_stats_depth_max() == search.SearchStatsC.depth_max.offset
    from crazyara_amd import selfplay
    assert chk.cra_sizeof_selfplay_settings() == C.sizeof(selfplay.SelfPlaySettingsC)
    assert chk.cra_sizeof_selfplay_stats() == C.sizeof(selfplay.SelfPlayStatsC)
    assert chk.cra_offsetof_selfplay_seed() == selfplay.SelfPlaySettingsC.seed.offset
    assert chk.cra_offsetof_selfplay_stats_wins() == selfplay.SelfPlayStatsC.wins.offset
    assert chk.cra_sizeof_go_limits() == C.sizeof(search.GoLimitsC)
    assert chk.cra_offsetof_go_limits_move_overhead() == search.GoLimitsC.move_overhead.offset


def test_integration_md_quotes_the_compiled_shim_verbatim():
    """INTEGRATION.md section 2 must be the file that is compiled against the reference (integration/hipapi.h), not a paraphrase."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(ROOT, "integration", "hipapi.h")).read()
    assert "```cpp\n" + shim + "```" in md
