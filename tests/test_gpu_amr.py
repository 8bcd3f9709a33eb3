"""Two AMR levels on the GPU (suhmo_amd/csrc/suhmo_amr.hip) against the oracle's amr2.c on the same inputs:
BITWISE for the coarse-fine interpolation, the fine-level operator update with a coarser level, the composite
residual (reflux), one AMR FAS V-cycle and the solve history.  Config: cfg3 of BASELINE.json
(exec/0_convergence_channelized/2lev_base/input.hydro: 64 x 16 base, y periodic, refined box around the moulin)."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
BC = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])     # 2lev_base/input.hydro:8-13,76
BC_NP = dict(type=[[0, 1], [1, 0]], value=[[3.0, 0.01], [-0.02, 7.0]], periodic=[0, 0])

CASES = [
    ("cfg3", 64, 16, sy.CFG3_PATCH, BC, sy.CFG3_PHYS, {}),
    ("cfg3-x2-patch-at-hi-x", 128, 32, (96, 8, 127, 23), BC, sy.CFG3_PHYS, {}),            # patch touches the x-hi domain side
    ("nonperiodic-values-mask", 64, 32, (0, 10, 15, 21), BC_NP, dict(sy.CFG3_PHYS, use_mask_gradients=1, cutOffbr=0.008, maxOffbr=0.012, cutOffB=1), {}),
]


def pair(oracle, case):
    from suhmo_amd import level
    _, nxc, nyc, patch, bc, ph, kw = case
    c, f = sy.amr2_fields(nxc, nyc, patch, **kw)
    O = oracle.OracleAmr2(nxc, nyc, c["dx"], c["dy"], bc, ph, patch, max_box=32, nthreads=2)
    O.coarse.set_inputs(c); O.coarse.build_mg_coefficients(); O.set_fine_inputs(f)
    G = level.HipAmr2(nxc, nyc, c["dx"], c["dy"], bc, ph, patch, max_box=32)
    G.coarse.set_inputs(c); G.coarse.build_mg_coefficients(); G.fine.set_inputs(f)
    return O, G, patch


def eq(a, b, what):
    assert np.array_equal(a, b), (what, float(np.max(np.abs(a - b))))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_amr2_pieces_bitwise(oracle, case):
    from suhmo_amd.level import F_PHI, F_RES, F_BX, F_BY
    O, G, patch = pair(oracle, case)
    ci0, cj0, ci1, cj1 = patch
    # coarse-fine interpolation of the head
    O.cf_interp(); G.cf_interp()
    a, b = O.fine_get(oracle.F_PHI, ghosted=True), G.fine.get(F_PHI, ghosted=True)
    eq(a[1:-1, :], b[1:-1, :], "cf ghosts x"); eq(a[:, 1:-1], b[:, 1:-1], "cf ghosts y")
    # operator of the fine level with the coarser level
    O.fine_update_operator(); G.fine_update_operator()
    eq(O.fine_get(oracle.F_BX), G.fine.get(F_BX), "fine bx"); eq(O.fine_get(oracle.F_BY), G.fine.get(F_BY), "fine by")
    # the base level's operator, then the composite residual with reflux
    O.coarse.update_operator(); G.coarse.update_operator()
    ro, rg = O.residual(), G.residual()
    eq(O.fine_get(oracle.F_RES), G.fine.get(F_RES), "fine residual")
    co, cg = O.coarse.get(oracle.F_RES), G.coarse.get(F_RES)
    # oracle/amr2.c keeps the composite coarse residual in its own array: compare through the norm and the cycle below
    assert ro == rg, (ro, rg)
    # fine relaxation with stored coarse-fine ghosts
    O.fine_gsrb(2); G.fine.gsrb(2)
    eq(O.fine_get(oracle.F_PHI), G.fine.get(F_PHI), "fine gsrb")
    O.close(); G.close()


def test_amr2_base_level_on_the_fused_kernels(oracle, monkeypatch):
    """512 x 128 base level relaxed by the fused K=2 GSRB / fused bCoef kernels, 256 x 128 fine patch"""
    from suhmo_amd.level import F_PHI
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "10000")
    O, G, patch = pair(oracle, ("big", 512, 128, (128, 32, 255, 95), BC, sy.CFG3_PHYS, dict(lx=512.0, ly=128.0)))
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (ho, hg)
    eq(O.fine_get(oracle.F_PHI), G.fine.get(F_PHI), "fine head")
    eq(O.coarse.get(oracle.F_PHI), G.coarse.get(F_PHI), "coarse head")
    O.close(); G.close()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_amr2_vcycle_and_solve_bitwise(oracle, case):
    from suhmo_amd.level import F_PHI
    O, G, patch = pair(oracle, case)
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=6, imin=30)
    O.vcycle(sp); G.vcycle(sp)
    eq(O.fine_get(oracle.F_PHI), G.fine.get(F_PHI), "fine head after one AMR V-cycle")
    eq(O.coarse.get(oracle.F_PHI), G.coarse.get(F_PHI), "coarse head after one AMR V-cycle")
    eq(O.coarse.get(oracle.F_RHS), G.coarse.get(1), "coarse rhs restored")
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (ho, hg)
    eq(O.fine_get(oracle.F_PHI), G.fine.get(F_PHI), "fine head after the solve")
    eq(O.coarse.get(oracle.F_PHI), G.coarse.get(F_PHI), "coarse head after the solve")
    O.close(); G.close()


PATCHES3 = (sy.CFG3_PATCH, (22, 11, 37, 20))      # level 2 nested (2+ cells) in level 1 = [16..47] x [8..23]
NCASES = [
    ("two-levels", 64, 16, (sy.CFG3_PATCH,), BC, sy.CFG3_PHYS),
    ("three-levels", 64, 16, PATCHES3, BC, sy.CFG3_PHYS),
    ("three-levels-nonperiodic-mask", 64, 32, ((0, 8, 23, 23), (4, 20, 35, 43)), BC_NP,
     dict(sy.CFG3_PHYS, use_mask_gradients=1, cutOffbr=0.008, maxOffbr=0.012, cutOffB=1)),
]


@pytest.mark.parametrize("case", NCASES, ids=[c[0] for c in NCASES])
def test_amr_n_levels_bitwise(oracle, case):
    from suhmo_amd import level
    from suhmo_amd.level import F_PHI
    _, nx0, ny0, patches, bc, ph = case
    fs = sy.amr_fields(nx0, ny0, patches)
    O = oracle.OracleAmr(nx0, ny0, fs[0]["dx"], fs[0]["dy"], bc, ph, patches, max_box=32, nthreads=2)
    G = level.HipAmr(nx0, ny0, fs[0]["dx"], fs[0]["dy"], bc, ph, patches, max_box=32)
    O.coarse.set_inputs(fs[0]); O.coarse.build_mg_coefficients()
    G.levels[0].set_inputs(fs[0]); G.levels[0].build_mg_coefficients()
    for l in range(1, len(fs)):
        O.set_patch_inputs(l, fs[l])
        G.levels[l].set_inputs(fs[l])
    assert O.residual() == G.residual()
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=5, imin=30)
    O.vcycle(sp); G.vcycle(sp)
    for l in range(1, len(fs)):
        eq(O.patch_get(l, oracle.F_PHI), G.levels[l].get(F_PHI), "level %d head after one AMR V-cycle" % l)
    eq(O.coarse.get(oracle.F_PHI), G.levels[0].get(F_PHI), "base head after one AMR V-cycle")
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (ho, hg)
    for l in range(1, len(fs)):
        eq(O.patch_get(l, oracle.F_PHI), G.levels[l].get(F_PHI), "level %d head after the solve" % l)
    eq(O.coarse.get(oracle.F_PHI), G.levels[0].get(F_PHI), "base head after the solve")
    O.close(); G.close()
