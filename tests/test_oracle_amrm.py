"""CPU checks of the multi-box AMR restatement (oracle/amrm.c): with one box per level it IS oracle/amrn.c bit for bit;
a level cut into more (abutting) boxes gives the same bits (fine-fine exchange = the neighbour's cell); on unions with
re-entrant corners, disjoint boxes and boxes on the domain boundary the coarse-fine interpolation is exact for quadratics
(centred and one-sided second-order stencils), and the 4-level FAS solve converges."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

BC = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])     # 2lev_base/input.hydro:8-13,76
BC_NP = dict(type=[[0, 1], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])  # run_C_3lev/input.hydro: no periodic side

# one box per level (= tests/test_oracle_amr.py PATCHES3, in each level's own index space)
ONE = ([(16, 8, 47, 23)], [(44, 22, 75, 41)])
# the same level-1 rectangle cut into three abutting boxes, level 2 cut into two
CUT = ([(16, 8, 31, 23), (32, 8, 47, 15), (32, 16, 47, 23)], [(44, 22, 59, 41), (60, 22, 75, 41)])
# unions: level 1 = an L (two abutting boxes of different height) + a disjoint box on the x-lo domain side;
# level 2 = two boxes inside the L (one of them against the re-entrant corner region) + one in the disjoint box;
# level 3 = one box in each of two level-2 boxes
UNION = ([(16, 8, 31, 23), (32, 8, 47, 15), (0, 2, 11, 13)],
         [(36, 20, 59, 27), (36, 28, 51, 43), (4, 8, 15, 19)],
         [(80, 44, 103, 51), (12, 20, 23, 31)])


def make(oracle, boxes, bc=BC, nx0=64, ny0=16, ph=sy.CFG3_PHYS, **kw):
    fs = sy.amrm_fields(nx0, ny0, boxes, **kw)
    A = oracle.OracleAmrM(nx0, ny0, fs[0]["dx"], fs[0]["dy"], bc, ph, boxes, max_box=32, nthreads=2)
    A.set_inputs(fs)
    return A, fs


def test_one_box_per_level_is_the_nested_patch_code(oracle):
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=4, imin=30)
    AM, fs = make(oracle, ONE)
    patches = ((8, 4, 23, 11), (22, 11, 37, 20))
    AN = oracle.OracleAmr(64, 16, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, patches, max_box=32, nthreads=2)
    AN.coarse.set_inputs(fs[0]); AN.coarse.build_mg_coefficients()
    for l in (1, 2):
        AN.set_patch_inputs(l, fs[l][0])
    assert AM.residual() == AN.residual()
    nm, hm = AM.solve(sp)
    nn, hn = AN.solve(sp)
    assert nm == nn and np.array_equal(hm, hn)
    for l in (1, 2):
        assert np.array_equal(AM.box_get(l, 0, oracle.F_PHI), AN.patch_get(l, oracle.F_PHI))
        assert np.array_equal(AM.box_get(l, 0, oracle.F_BX), AN.patch_get(l, oracle.F_BX))
    assert np.array_equal(AM.coarse.get(oracle.F_PHI), AN.coarse.get(oracle.F_PHI))
    AM.close(); AN.close()


def test_cutting_a_level_into_boxes_changes_no_bit(oracle):
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
    A1, _ = make(oracle, ONE)
    A3, _ = make(oracle, CUT)
    assert A1.residual() == A3.residual()
    n1, h1 = A1.solve(sp)
    n3, h3 = A3.solve(sp)
    assert n1 == n3 and np.array_equal(h1, h3)
    for l in (1, 2):
        for f in (oracle.F_PHI, oracle.F_RES):
            a, b = A1.level_array(l, f), A3.level_array(l, f)
            assert np.array_equal(a, b, equal_nan=True), (l, f)
    assert np.array_equal(A1.coarse.get(oracle.F_PHI), A3.coarse.get(oracle.F_PHI))
    A1.close(); A3.close()


def test_bad_hierarchies_are_refused(oracle):
    fs = sy.amrm_fields(64, 16, ())
    for boxes in (([(16, 8, 31, 23), (30, 8, 47, 15)],),            # overlap
                  ([(15, 8, 30, 23)],),                               # not coarse-aligned
                  ([(16, 8, 47, 23)], [(32, 16, 63, 47)])):           # level 2 touches the edge of level 1
        with pytest.raises(ValueError):
            oracle.OracleAmrM(64, 16, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, boxes, max_box=32)


def test_cf_interp_on_a_union_is_exact_for_quadratics(oracle):
    A, fs = make(oracle, UNION[:1], bc=BC_NP)
    q = lambda X, Y: 3.0 + 0.2 * X - 0.1 * Y + 0.01 * X * X + 0.02 * Y * Y
    dx0, dy0 = fs[0]["dx"], fs[0]["dy"]
    A.coarse.set(oracle.F_PHI, q(*np.meshgrid((np.arange(64) + 0.5) * dx0, (np.arange(16) + 0.5) * dy0)))
    exact = {}
    for k, (lo0, lo1, hi0, hi1) in enumerate(UNION[0]):
        X, Y = np.meshgrid((np.arange(lo0 - 1, hi0 + 2) + 0.5) * dx0 / 2, (np.arange(lo1 - 1, hi1 + 2) + 0.5) * dy0 / 2)
        exact[k] = q(X, Y)
        A.box_set(1, k, oracle.F_PHI, exact[k][1:-1, 1:-1])
    A.cf_interp_phi(1)
    A.exchange(1, oracle.F_PHI)
    scale = max(np.max(np.abs(e)) for e in exact.values())
    ncf = nff = 0
    for k, (lo0, lo1, hi0, hi1) in enumerate(UNION[0]):
        g = A.box_get(1, k, oracle.F_PHI, ghosted=True)
        for (jj, ii) in [(j, 0) for j in range(1, g.shape[0] - 1)] + [(j, g.shape[1] - 1) for j in range(1, g.shape[0] - 1)] + \
                        [(0, i) for i in range(1, g.shape[1] - 1)] + [(g.shape[0] - 1, i) for i in range(1, g.shape[1] - 1)]:
            gi, gj = lo0 - 1 + ii, lo1 - 1 + jj
            if not (0 <= gi < 128 and 0 <= gj < 32):
                continue                                     # domain ghost
            inside_other = any(b[0] <= gi <= b[2] and b[1] <= gj <= b[3] for b in UNION[0])
            err = abs(g[jj, ii] - exact[k][jj, ii])
            if inside_other:
                nff += 1
                assert err == 0.0                            # the neighbour's cell itself
            else:
                ncf += 1                                     # next to the other box of the L the stencil is one-sided: still exact
                assert err < 1e-12 * scale, (k, gi, gj, err)
    assert nff == 2 * 8 and ncf > 100, (nff, ncf)
    A.close()


def test_four_level_solve_on_unions_converges(oracle):
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=40, imin=40)
    A, fs = make(oracle, UNION, bc=BC_NP)
    r0 = A.residual()
    n, hist = A.solve(sp)
    assert hist[-1] < 1e-6 * r0 and hist[-1] < 1e-9 * np.max(hist), hist
    # covered cells hold the average of the finer level (up to the last post-smoothing)
    for l in (3, 2):
        fine, coarse = A.level_array(l, oracle.F_PHI), A.level_array(l - 1, oracle.F_PHI)
        avg = 0.25 * (fine[0::2, 0::2] + fine[0::2, 1::2] + fine[1::2, 0::2] + fine[1::2, 1::2])
        m = ~np.isnan(avg)
        assert m.sum() > 0 and np.max(np.abs(avg[m] - coarse[m])) < 1e-3 * np.max(np.abs(avg[m]))
    A.close()
