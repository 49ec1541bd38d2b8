"""CPU: the C-ABI library builds, loads and exports every symbol include/efg_hip.h declares (no
compute calls without a GPU), and argument validation works on the host side."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from efg_amd import _lib, build

    build.build()
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    from efg_amd import _lib

    header = open(os.path.join(ROOT, "include", "efg_hip.h")).read()
    declared = set(re.findall(r"\b(efg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libefg_hip.so does not export " + name
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_version_and_error_channel(lib):
    assert lib.efg_version().decode().startswith("efg_hip")
    assert b"gfx950" in lib.efg_version()
    # host-side validation: bad arguments are rejected before any device work
    vs, cr = (ctypes.c_float * 3)(0.1, 0.1, 0.15), (ctypes.c_float * 6)(-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
    assert lib.efg_hard_voxelize_workspace_bytes(1000, 1, 5, 0, 10, vs, cr) == 0
    assert lib.efg_hard_voxelize_workspace_bytes(1000, 1, 2, 5, 10, vs, cr) == 0
    assert lib.efg_hard_voxelize_workspace_bytes(180000, 2, 5, 5, 120000, vs, cr) > 0
    assert lib.efg_spconv_packed_weight_bytes(64, 27, 64, 0) >= 27 * 64 * 64 * 4
    assert lib.efg_spconv_wgrad_workspace_bytes(100000, 64, 64, 27) > 0
    shp = (ctypes.c_int * 3)(41, 1504, 1504)
    assert lib.efg_spconv_index_bytes(2, shp) == ((2 * 41 * 1504 * 1504 + 31) // 32) * 8
    big = (ctypes.c_int * 3)(4100, 1504, 1504)
    assert lib.efg_spconv_index_bytes(2, big) == 0  # >= 2^32 cells is refused
    assert b"2^32" in lib.efg_last_error()
    rc = lib.efg_msda_forward_f32(None, None, None, None, None, 1, 4, 2, 6, 1, 1, 1, None, None)  # d % 4 != 0
    assert rc == -1 and b"head dim" in lib.efg_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from efg_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_LIB_PATH", "/nonexistent/libefg_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
