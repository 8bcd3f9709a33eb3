"""The time step on an AMR hierarchy (suhmo_amr_timestep: per-level phases + PiecewiseLinearFillPatch / QuadCFInterp
ghosts, AMR solve, average down, Picard test over the uncovered cells) against oracle/amr_step.c on the same inputs:
BITWISE on every level over several steps; a one-level hierarchy equals suhmo_level_timestep."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu

CASES = [
    ("2-levels", 64, 32, ((16, 8, 47, 23),), dict(), 3),
    ("3-levels", 64, 32, ((16, 8, 47, 23), (40, 22, 79, 41)), dict(), 3),
    ("2-levels-patch-on-the-boundary", 64, 32, ((0, 0, 31, 15),), dict(), 2),
    ("2-levels-diffusion", 64, 32, ((16, 8, 47, 23),), dict(diffFactor=1.0), 2),
    ("2-levels-implicit-gap", 64, 32, ((16, 8, 47, 23),), dict(diffFactor=1.0, use_impl_diff=1), 3),
    ("3-levels-implicit-gap", 64, 32, ((16, 8, 47, 23), (40, 22, 79, 41)), dict(diffFactor=1.0, use_impl_diff=1), 2),
]


@pytest.mark.parametrize("name,nx0,ny0,patches,mpo,nsteps", CASES, ids=[c[0] for c in CASES])
def test_amr_timestep_bitwise(oracle, name, nx0, ny0, patches, mpo, nsteps):
    from suhmo_amd import model
    m = dict(sy.A3_MODEL, **mpo)
    sts = sy.shmip_amr_states(nx0, ny0, patches, rough=0.5)
    O = oracle.OracleAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16, nthreads=2)
    G = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16)
    for l, st in enumerate(sts):
        O.set_state(l, st)
        G.set_state(l, st)
    v = lambda a: np.array(a)[1:-1, 1:-1]
    for k in range(nsteps):
        co, cg = O.timestep(m["dt"]), G.timestep(m["dt"])
        assert co == cg, (k, co, cg)
        for l in range(len(sts)):
            for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("mR", oracle.OM_MR), ("Pw", oracle.OM_PW), ("rhs_h", oracle.OM_RHSH),
                            ("Re", oracle.OM_RE)):
                a, b = v(O.field(l, fid)), G.get(l, nm)
                assert np.array_equal(a, b, equal_nan=True), (name, k, l, nm, float(np.nanmax(np.abs(a - b))))
            for nm, fid in (("qwx", oracle.OM_QWX), ("qwy", oracle.OM_QWY)):
                a, b = np.array(O.field(l, fid)), G.get(l, nm)
                assert np.array_equal(a, b, equal_nan=True), (name, k, l, nm, float(np.nanmax(np.abs(a - b))))
            # ghost cells of the gap height after the step: PiecewiseLinearFillPatch on coarse-fine sides, copies on domain sides
            a, b = np.array(O.field(l, oracle.OM_B)), G.get(l, "B", ghosted=True)
            assert np.array_equal(a[1:-1, :], b[1:-1, :]) and np.array_equal(a[:, 1:-1], b[:, 1:-1]), (name, k, l, "B ghosts")
    O.close()
    G.close()


def test_amr_moulin_source_and_timestep(oracle):
    """moulins on the hierarchy: integrals over the uncovered cells of all levels, source term per level, covered cells =
    average of the finer level (1e-13: two exp libraries); the time step then consumes the oracle's source term bit for bit"""
    from suhmo_amd import model, level as lv
    from test_gpu_moulin import moulins
    nx0, ny0, patches = 64, 32, ((16, 8, 47, 23), (40, 22, 79, 41))
    m = dict(sy.A3_MODEL, use_moulin_source=1, distributed_input=7.93e-11)
    sts = sy.shmip_amr_states(nx0, ny0, patches, rough=0.5)
    pos, sg, fl = moulins(6, 3)
    pos[0] = (52000.0, 10500.0)                       # one moulin inside the finest patch
    sg = np.full(6, 1500.0)
    O = oracle.OracleAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16, nthreads=2)
    G = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16)
    for l, st in enumerate(sts):
        O.set_state(l, st)
        G.set_state(l, st)
    io, ig = O.moulin_source(pos, sg, fl, 0.8), G.moulin_source(pos, sg, fl, 0.8)
    assert np.max(np.abs(io - ig)) <= 1e-13 * np.max(io)
    tot = 0.0
    for l in range(3):
        a, b = np.array(O.field(l, oracle.OM_MSRC))[1:-1, 1:-1], G.get(l, "msrc")
        assert np.max(np.abs(a - b)) <= 1e-13 * np.max(a), l
        # flux delivered by the uncovered cells of this level
        cov = np.zeros_like(b, dtype=bool)
        if l < 2:
            ci0, cj0, ci1, cj1 = patches[l]
            cov[cj0 - sts[l]["j0"]:cj1 + 1 - sts[l]["j0"], ci0 - sts[l]["i0"]:ci1 + 1 - sts[l]["i0"]] = True
        tot += b[~cov].sum() * sts[l]["dx"] * sts[l]["dy"]
        G.levels[l].set(lv.F_MSRC, a)                 # continue from identical source terms
    assert abs(tot - 0.8 * fl.sum()) < 1e-11 * fl.sum()   # every moulin delivers its flux over the composite grid
    for k in range(2):
        assert O.timestep(m["dt"]) == G.timestep(m["dt"])
        for l in range(3):
            for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("rhs_h", oracle.OM_RHSH)):
                assert np.array_equal(np.array(O.field(l, fid))[1:-1, 1:-1], G.get(l, nm)), (k, l, nm)
    O.close()
    G.close()


def test_one_level_hierarchy_equals_the_level_timestep():
    from suhmo_amd import model
    nx, ny, m = 64, 32, dict(sy.A3_MODEL)
    st = sy.shmip_amr_states(nx, ny, (), rough=0.5)[0]
    A = model.HipAmrModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, (), max_box=16)
    S = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=16)
    A.set_state(0, st)
    S.set_state(st)
    for k in range(3):
        assert A.timestep(m["dt"]) == S.timestep(m["dt"])
        for nm in ("head", "B", "mR", "qwx", "qwy"):
            assert np.array_equal(A.get(0, nm), S.get(nm)), (k, nm)
    A.close()
    S.close()


def test_amr_timestep_refuses_what_is_not_built():
    from suhmo_amd import capi, model
    sts = sy.shmip_amr_states(64, 32, ((16, 8, 47, 23),))
    G = model.HipAmrModel(64, 32, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, dict(sy.A3_MODEL, diffFactor=0.0, use_impl_diff=1),
                          ((16, 8, 47, 23),), max_box=16)
    for l, st in enumerate(sts):
        G.set_state(l, st)
    with pytest.raises(capi.SuhmoError):
        G.timestep(3600.0)
    G.close()


def test_error_paths_of_the_new_entry_points():
    """refusals carry a message instead of wrong numbers: moulin boxes that do not match the handles, strips without boxes,
    a boundary-condition change that would alter the periodicity, the time-varying recharge without a surface field"""
    import ctypes as C
    from suhmo_amd import capi, model, level as lv
    patches = ((16, 8, 47, 23),)
    sts = sy.shmip_amr_states(64, 32, patches)
    G = model.HipAmrModel(64, 32, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, dict(sy.A3_MODEL), patches, max_box=16)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    pos, sg, fl, integ = np.array([3.0e4, 1.0e4]), np.array([500.0]), np.array([10.0]), np.zeros(1)
    bad = (C.c_int * 4)(16, 8, 47, 25)                        # not the box the fine handle was created with
    rc = capi.lib().suhmo_amr_moulin_source(G.amr._arr, 2, bad, 1, dp(pos), dp(sg), dp(fl), 1.0, dp(integ), None)
    assert rc == -1 and b"patch_boxes" in capi.lib().suhmo_last_error()
    good = (C.c_int * 4)(*patches[0])
    assert capi.lib().suhmo_amr_moulin_source(G.amr._arr, 2, good, 1, dp(pos), dp(sg), dp(fl), 1.0, dp(integ), None) == 0
    i2 = G.moulin_source(pos.reshape(1, 2), sg, fl, 1.0)      # geometry from the handles: the same integral
    assert integ[0] == i2[0] > 0.0
    bc = lv._bc(dict(type=[[0, 0], [1, 1]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1]))
    assert capi.lib().suhmo_level_set_bc(G.levels[0].h, C.byref(bc)) == -1 and b"periodicity" in capi.lib().suhmo_last_error()
    assert capi.lib().suhmo_level_time_varying_recharge(G.levels[0].h, 3.0, 1e-10, None) == -1
    # a hierarchy whose patch is not properly nested is refused by the time step
    G2 = model.HipAmrModel(64, 32, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, dict(sy.A3_MODEL), ((16, 8, 47, 23), (33, 17, 92, 45)), max_box=16)
    with pytest.raises(capi.SuhmoError):
        G2.timestep(3600.0)
    G.close()
    G2.close()
