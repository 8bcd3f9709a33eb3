"""Moulin source term on the GPU (suhmo_level_moulin_source; Calc_moulin_integral +
Calc_moulin_source_term_distributed, src/AmrHydro.cpp:1866-2066) against the oracle.  exp() comes from two
different math libraries, so the bar here is a stated tolerance: 1e-13 relative to the largest value (the
integrals additionally differ in summation order), not bit equality.  The time step that consumes the term is
compared bit for bit by feeding both sides the same source array."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
RTOL = 1e-13


def moulins(n, seed, lx=1.0e5, ly=2.0e4):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(0.05 * lx, 0.95 * lx, n), rng.uniform(0.05 * ly, 0.95 * ly, n)], axis=1)
    return pos, np.full(n, 200.0), np.full(n, 90.0 / n)          # exec/B_SHMIP/B<k>/input.hydro:37-40


@pytest.mark.parametrize("nx,ny,n", [(320, 64, 1), (320, 64, 100), (1280, 256, 10)])
def test_moulin_source_matches_oracle(oracle, nx, ny, n):
    from suhmo_amd import model
    st = sy.shmip_initial_state(nx, ny)
    pos, sg, fl = moulins(n, 5) if n > 1 else (np.array([[59000.0, 8000.0]]), np.array([200.0]), np.array([90.0]))
    G = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, sy.A3_MODEL)
    integ_g = G.moulin_source(pos, sg, fl, 0.75)
    src_g = G.get("msrc")
    src_o, integ_o = oracle.moulin_source(nx, ny, st["dx"], st["dy"], pos, sg, fl, 0.75)
    assert np.max(np.abs(integ_g - integ_o)) <= RTOL * np.max(integ_o)
    assert np.max(np.abs(src_g - src_o)) <= RTOL * np.max(src_o)
    # each moulin delivers its flux (times the time factor): sum(src) dx dy = 0.75 * sum(flux)
    assert abs(src_g.sum() * st["dx"] * st["dy"] - 0.75 * fl.sum()) < 1e-11 * fl.sum()
    assert np.count_nonzero(src_g) < src_g.size or n == 100          # far cells are exactly zero (underflow), as in the reference
    G.close()


def test_timestep_with_moulin_source_bitwise(oracle):
    from suhmo_amd import model, level as lv
    nx, ny = 160, 32
    m = dict(sy.A3_MODEL, use_moulin_source=1, ramp=0.8, distributed_input=7.93e-11)   # B_SHMIP: background input
    st = sy.shmip_initial_state(nx, ny)
    pos, sg, fl = moulins(3, 9)
    sg = sg * 8.0                                                    # a few cells wide on this coarse grid
    src, _ = oracle.moulin_source(nx, ny, st["dx"], st["dy"], pos, sg, fl, 1.0)
    O = oracle.OracleModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=32, nthreads=2)
    G = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=32)
    O.set_state(st); G.set_state(st)
    O.field(oracle.OM_MSRC)[1:-1, 1:-1] = src
    G.level.set(lv.F_MSRC, src)
    for k in range(3):
        assert O.timestep(m["dt"]) == G.timestep(m["dt"])
        for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("rhs_h", oracle.OM_RHSH), ("mR", oracle.OM_MR)):
            a, b = np.array(O.field(fid))[1:-1, 1:-1], G.get(nm)
            assert np.array_equal(a, b), (k, nm, float(np.max(np.abs(a - b))))
    assert np.all(np.isfinite(G.get("head")))
    O.close(); G.close()


@pytest.mark.parametrize("impl", [1, 0])
def test_timestep_with_diffusion_bitwise(oracle, impl):
    """suhmo.diffFactor = 1: the diffusive term div(D grad b) in RHS_h and, with solver.use_ImplDiff, the implicit
    gap-height solve (exec/B_SHMIP/B<k>/input.hydro:31,56) -- head, gap height, D and the term itself bit for bit"""
    from suhmo_amd import model, level as lv
    nx, ny = 160, 32
    m = dict(sy.A3_MODEL, use_moulin_source=1, ramp=1.0, distributed_input=7.93e-11, diffFactor=1.0, use_impl_diff=impl)
    st = sy.shmip_initial_state(nx, ny)
    pos, sg, fl = moulins(3, 9)
    sg = sg * 8.0
    src, _ = oracle.moulin_source(nx, ny, st["dx"], st["dy"], pos, sg, fl, 1.0)
    O = oracle.OracleModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=32, nthreads=2)
    G = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=32)
    O.set_state(st); G.set_state(st)
    O.field(oracle.OM_MSRC)[1:-1, 1:-1] = src
    G.level.set(lv.F_MSRC, src)
    for k in range(4):
        assert O.timestep(m["dt"]) == G.timestep(m["dt"]), k
        for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("rhs_h", oracle.OM_RHSH), ("mR", oracle.OM_MR)):
            a, b = np.array(O.field(fid))[1:-1, 1:-1], G.get(nm)
            assert np.array_equal(a, b), (k, nm, float(np.max(np.abs(a - b))))
    assert np.all(np.isfinite(G.get("B"))) and G.get("B").min() > 0.0
    O.close(); G.close()
