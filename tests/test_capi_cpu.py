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


def test_postproc_host_halves_without_a_gpu(lib):
    """suhmo_postproc_finish / suhmo_postproc_temporal are host arithmetic on column sums (what the ranks add up): the table and the
    daily row of AmrHydro's post-processing (src/AmrHydro.cpp:3778-3810, 4023-4053) against a numpy restatement"""
    import ctypes as C
    import numpy as np
    nx, dx = 256, 6000.0 / 256
    rng = np.random.default_rng(3)
    sums = rng.uniform(0.5, 2.0, size=(8, nx))
    sums[7] = np.round(rng.uniform(10, 60, size=nx))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    out = np.zeros(6)
    assert lib.suhmo_postproc_temporal(dp(sums), nx, dx, dp(out)) == 0
    x = (np.arange(nx) + 0.5) * dx
    band = lambda lo, hi: sums[6][(x > lo) & (x < hi)].sum() / sums[7][(x > lo) & (x < hi)].sum()
    want = [sums[6].sum() / sums[7].sum(), band(600.0, 900.0), band(3000.0, 3300.0), band(5100.0, 5400.0), (sums[4] + sums[5])[1:].sum(), -sums[1][1]]
    assert np.allclose(out, want, rtol=1e-13, atol=0)
    table = np.zeros((nx, 8))
    assert lib.suhmo_postproc_finish(dp(sums), nx, dx, dp(table)) == 0
    assert np.allclose(table[:, 5], np.cumsum(sums[4][::-1])[::-1], rtol=1e-13) and np.allclose(table[:, 2], -sums[1])
