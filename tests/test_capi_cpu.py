"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/suhmo_hip.h declares; without a GPU the product path fails
loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

from suhmo_amd import capi, synthetic as sy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build()
    return capi.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "suhmo_hip.h")).read()
    declared = set(re.findall(r"\b(suhmo_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"suhmo_exchange_fn", "suhmo_allreduce_max_fn"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    # sizes the C compiler gives the ABI structs (LP64): guards the ctypes mirror
    assert ctypes.sizeof(capi.Phys) == 7 * 8 + 3 * 4 + 4
    assert ctypes.sizeof(capi.BC) == 16 + 32 + 8
    assert ctypes.sizeof(capi.SolverParams) == 5 * 4 + 4 + 3 * 8 + 2 * 4
    assert ctypes.sizeof(capi.LevelDesc) == 16 + 16 + 4 + 4 + 8 + 4 + 4 + 16 + 56 + 72 + 4 + 4 + 8 + 8


def test_fails_loudly_without_gpu(lib):
    if lib.suhmo_device_count() > 0:
        pytest.skip("a GPU is present")
    from suhmo_amd import level
    f = sy.shmip_fields(64, 64)
    with pytest.raises(capi.SuhmoError):
        level.HipLevel(64, 64, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS)


def test_product_package_never_imports_oracle():
    # the oracle is test infrastructure: nothing under suhmo_amd/ may import, link or load it
    pat = re.compile(r"import\s+oracle|from\s+oracle|from\s+\.+oracle|liboracle|pyoracle|level_shim|suhmo_oracle")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "suhmo_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                assert not pat.search(open(os.path.join(dirpath, fn)).read()), (dirpath, fn)
