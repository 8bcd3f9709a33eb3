"""Known-answer tests that pin the CPU oracle without the reference (SURVEY.md 8c, items
1-10): the reference ships no kernel-level vectors for this path and cannot be built
here, so the oracle is checked against (i) analytic identities, (ii) an independent
numpy restatement (tests/npref.py) bit for bit, (iii) a hand-derived 4x4 fixture.
CPU only."""
import json
import os

import numpy as np
import pytest

from suhmo_amd import synthetic as sy
from tests import npref

HERE = os.path.dirname(os.path.abspath(__file__))


def mk(oracle, f, bc, ph, alpha=0.0, beta=-1.0, max_box=16, nthreads=1):
    L = oracle.OracleLevel(f["nx"], f["ny"], f["dx"], f["dy"], bc, ph, alpha, beta, max_box, nthreads)
    L.set_inputs(f)
    return L


def v(a):
    return a[1:-1, 1:-1]


def test_laplacian_of_quadratic_exact(oracle):
    # item 1: b = -1, beta = -1, A = 0  ->  L = -lap(phi); quadratic, power-of-two spacing: exact
    nx, ny, dx, dy = 32, 16, 1.0, 0.5
    f = sy.random_fields(nx, ny, dx, dy)
    x = (np.arange(nx) + 0.5) * dx
    y = (np.arange(ny) + 0.5) * dy
    X, Y = np.meshgrid(x, y)
    f["phi"] = X * X + 2.0 * Y * Y
    f["bx"][:] = -1.0
    f["by"][:] = -1.0
    ph = dict(sy.RANDOM_PHYS, A=0.0)
    L = mk(oracle, f, sy.RANDOM_BC, ph)
    L.apply_op()
    out = L.get(oracle.F_LPHI)
    assert np.all(out[1:-1, 1:-1] == -(2.0 + 4.0))
    # constant phi, A = 0, periodic-free interior: L = 0
    f["phi"] = np.full((ny, nx), 3.25)
    L = mk(oracle, f, sy.RANDOM_BC, ph)
    L.apply_op()
    assert np.all(L.get(oracle.F_LPHI)[1:-1, 1:-1] == 0.0)


@pytest.mark.parametrize("bc", [sy.RANDOM_BC, sy.CONV_BC, sy.A3_BC])
def test_apply_residual_lambda_match_numpy_bitwise(oracle, bc):
    f = sy.random_fields(48, 32)
    ph = sy.RANDOM_PHYS
    alpha, beta = 0.7, -1.0
    L = mk(oracle, f, bc, ph, alpha, beta)
    L.apply_op()
    L.residual()
    L.reset_lambda()
    pg = npref.fill_ghosts(f["phi"], bc, f["dx"], f["dy"])
    nl, dnl = npref.nl_terms(f["phi"], v(f["B"]), v(f["Pi"]), v(f["zb"]), v(f["mask"]), ph)
    Lnp = npref.op(pg, f["aCoef"], f["bx"], f["by"], nl, alpha, beta, f["dx"], f["dy"])
    assert np.array_equal(L.get(oracle.F_NL), nl)
    assert np.array_equal(L.get(oracle.F_DNL), dnl)
    assert np.array_equal(L.get(oracle.F_LPHI), Lnp)
    # item 3: res == rhs - applyOp bitwise
    assert np.array_equal(L.get(oracle.F_RES), f["rhs"] - Lnp)
    assert np.array_equal(L.get(oracle.F_LAMBDA),
                          npref.lam(f["aCoef"], f["bx"], f["by"], alpha, beta, f["dx"], f["dy"]))


@pytest.mark.parametrize("bc", [sy.RANDOM_BC, sy.CONV_BC])
@pytest.mark.parametrize("max_box", [8, 16, 48])
def test_gsrb_matches_numpy_and_is_decomposition_independent(oracle, bc, max_box):
    # item 5: colour = parity of GLOBAL i+j+pass, result independent of the boxes
    f = sy.random_fields(48, 32)
    ph = sy.RANDOM_PHYS
    alpha, beta = 0.3, -1.0
    L = mk(oracle, f, bc, ph, alpha, beta, max_box=max_box, nthreads=2)
    L.gsrb(3)
    phi = f["phi"]
    for _ in range(3):
        phi = npref.gsrb_sweep(phi, f["rhs"], f["aCoef"], f["bx"], f["by"], f["B"], f["Pi"], f["zb"],
                               f["mask"], ph, bc, alpha, beta, f["dx"], f["dy"])
    assert np.array_equal(L.get(oracle.F_PHI), phi)


def test_gsrb_red_pass_touches_only_even_cells(oracle):
    # run one sweep with rhs chosen so black cells are at a fixed point -> only red move.
    f = sy.random_fields(16, 16, with_mask_holes=False)
    ph = dict(sy.RANDOM_PHYS, A=0.0)
    L = mk(oracle, f, sy.CONV_BC, ph)
    L.gsrb(1)
    new = L.get(oracle.F_PHI)
    # independent: numpy sweep, then verify that after pass 0 exactly the even cells changed
    jj, ii = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    pg = npref.fill_ghosts(f["phi"], sy.CONV_BC, f["dx"], f["dy"])
    nl = np.zeros((16, 16))
    Lp = npref.op(pg, f["aCoef"], f["bx"], f["by"], nl, 0.0, -1.0, f["dx"], f["dy"])
    lm = npref.lam(f["aCoef"], f["bx"], f["by"], 0.0, -1.0, f["dx"], f["dy"])
    red = f["phi"] + (f["rhs"] - Lp) / (1e-16 + lm + 0.0)
    even = (ii + jj) % 2 == 0
    assert np.array_equal(new[even], red[even])          # red cells: one Jacobi step from old data
    assert not np.array_equal(new[~even], f["phi"][~even])  # black cells moved afterwards


def test_gsrb_fixed_point(oracle):
    # item 2: rhs = L(phi)  ->  phi unchanged to a few ulp
    f = sy.shmip_fields(64, 32)
    L = mk(oracle, f, sy.A3_BC, sy.A3_PHYS)
    L.update_operator()
    L.apply_op()
    L.set(oracle.F_RHS, L.get(oracle.F_LPHI))
    L.gsrb(2)
    out = L.get(oracle.F_PHI)
    assert np.max(np.abs(out - f["phi"]) / np.abs(f["phi"])) < 1e-12


def test_restrict_residual_is_four_cell_mean(oracle):
    # item 4
    f = sy.random_fields(32, 32)
    L = mk(oracle, f, sy.RANDOM_BC, sy.RANDOM_PHYS, 0.2, -1.0)
    L.residual()
    r = L.get(oracle.F_RES)
    L.restrict_residual()
    rc = L.get(oracle.F_RES, depth=1)
    assert np.array_equal(rc, npref.restrict_sum4(r))
    mean = 0.25 * (r[0::2, 0::2] + r[0::2, 1::2] + r[1::2, 0::2] + r[1::2, 1::2])
    assert np.allclose(rc, mean, rtol=1e-14, atol=1e-20)
    L.restrict_r()
    assert np.array_equal(L.get(oracle.F_PHI, depth=1), npref.restrict_sum4(f["phi"]))


def test_prolong_restrict_identity_and_bilinear_weights(oracle):
    # item 6
    f = sy.random_fields(32, 16)
    L = mk(oracle, f, sy.RANDOM_BC, sy.RANDOM_PHYS)
    L.set(oracle.F_PHI, np.zeros((16, 32)))
    rng = np.random.default_rng(3)
    c = rng.uniform(-1, 1, size=(8, 16))
    L.prolong_increment(c)
    fine = L.get(oracle.F_PHI)
    assert np.array_equal(fine, np.repeat(np.repeat(c, 2, axis=0), 2, axis=1))
    L.restrict_r()
    assert np.array_equal(L.get(oracle.F_PHI, depth=1), c)
    # PROLONG_2_NL: weights 9/16, 3/16, 3/16, 1/16 sum to one and reproduce linear functions
    ny, nx = 16, 32
    cg = np.ones((ny // 2 + 2, nx // 2 + 2))
    out = oracle.prolong2(np.zeros((ny, nx)), cg)
    assert np.all(out == 1.0)
    xc = (np.arange(-1, nx // 2 + 1) + 0.5) * 2.0
    yc = (np.arange(-1, ny // 2 + 1) + 0.5) * 2.0
    XC, YC = np.meshgrid(xc, yc)
    cg = 3.0 * XC - 0.5 * YC + 1.0
    out = oracle.prolong2(np.zeros((ny, nx)), cg)
    XF, YF = np.meshgrid(np.arange(nx) + 0.5, np.arange(ny) + 0.5)
    assert np.allclose(out, 3.0 * XF - 0.5 * YF + 1.0, rtol=0, atol=1e-12)


def test_dnl_is_derivative_of_nl(oracle):
    # item 7
    f = sy.random_fields(24, 24, with_mask_holes=True)
    ph = sy.RANDOM_PHYS
    L = mk(oracle, f, sy.RANDOM_BC, ph)
    L.nonlinear()
    dnl = L.get(oracle.F_DNL)
    h = 1e-4
    vals = []
    for s in (+1, -1):
        L.set(oracle.F_PHI, f["phi"] + s * h)
        L.nonlinear()
        vals.append(L.get(oracle.F_NL))
    fd = (vals[0] - vals[1]) / (2 * h)
    # where B is clamped (cutOffbr > B or maxOffbr < B) the reference scales nl and dnl by the
    # same factor B/br, so the identity holds there as well
    assert np.allclose(dnl, fd, rtol=1e-6, atol=1e-30)
    assert np.all(dnl[v(f["mask"]) < 0] == 0.0)


def test_lambda_is_operator_diagonal(oracle):
    # item 8 (linear part: A = 0)
    f = sy.random_fields(16, 16)
    ph = dict(sy.RANDOM_PHYS, A=0.0)
    alpha, beta = 0.4, -1.0
    L = mk(oracle, f, sy.CONV_BC, ph, alpha, beta)
    L.apply_op()
    L0 = L.get(oracle.F_LPHI)
    L.reset_lambda()
    lam = L.get(oracle.F_LAMBDA)
    for (j, i) in [(5, 7), (8, 8), (3, 12)]:
        p = f["phi"].copy()
        p[j, i] += 1.0
        L.set(oracle.F_PHI, p)
        L.apply_op()
        d = L.get(oracle.F_LPHI)[j, i] - L0[j, i]
        assert abs(d - lam[j, i]) <= 1e-9 * abs(lam[j, i])


def test_reynolds_solves_quadratic_via_bcoef(oracle):
    # item 9: uniform head gradient c, uniform B  ->  b = -B^3 g / (12 nu (1 + omega Re)),
    # omega Re^2 + Re - B^3 g c / (12 nu^2) = 0
    nx, ny, dx, dy, c, B0 = 32, 16, 10.0, 5.0, 2.5e-3, 0.02
    f = sy.random_fields(nx, ny, dx, dy, with_mask_holes=False)
    x = (np.arange(nx) + 0.5) * dx
    f["phi"] = np.tile(c * x, (ny, 1))
    f["B"][:] = B0
    ph = dict(sy.A3_PHYS)
    bc = dict(type=[[0, 1], [1, 1]], value=[[0.0, c], [0.0, 0.0]], periodic=[0, 1])
    L = mk(oracle, f, bc, ph)
    L.update_operator()
    bx, by = L.get(oracle.F_BX), L.get(oracle.F_BY)
    om, nu, g = ph["omega"], ph["nu"], ph["grav"]
    k = B0 ** 3 * g * c / (12 * nu * nu)
    Re = (-1 + np.sqrt(1 + 4 * om * k)) / (2 * om)
    assert abs(om * Re * Re + Re - k) < 1e-9 * k
    b = -(B0 ** 3 * g) / (12 * nu * (1 + om * Re))
    assert np.allclose(bx, b, rtol=1e-11)
    assert np.allclose(by, b, rtol=1e-11)
    bxn, byn = npref.bcoef_update(f["phi"], f["B"], f["mask"], ph, bc, dx, dy)
    assert np.array_equal(bx, bxn) and np.array_equal(by, byn)


@pytest.mark.parametrize("bc,ph", [(sy.RANDOM_BC, sy.RANDOM_PHYS), (sy.CONV_BC, sy.A3_PHYS),
                                   (sy.A3_BC, dict(sy.RANDOM_PHYS, use_mask_gradients=0))])
def test_bcoef_update_matches_numpy_bitwise(oracle, bc, ph):
    f = sy.random_fields(48, 32)
    L = mk(oracle, f, bc, ph, max_box=16, nthreads=2)
    L.update_operator()
    bxn, byn = npref.bcoef_update(f["phi"], f["B"], f["mask"], ph, bc, f["dx"], f["dy"])
    assert np.array_equal(L.get(oracle.F_BX), bxn)
    assert np.array_equal(L.get(oracle.F_BY), byn)
    assert np.array_equal(L.get(oracle.F_LAMBDA), npref.lam(f["aCoef"], bxn, byn, 0.0, -1.0, f["dx"], f["dy"]))


def test_hand_fixture_4x4(oracle):
    # item 10: fixture generated by tests/golden/make_gsrb_4x4.py (scalar Python, no numpy
    # vector ops, no oracle) from the formulas in SURVEY.md Appendix C.
    with open(os.path.join(HERE, "golden", "gsrb_4x4.json")) as fh:
        fx = json.load(fh)
    f = {k: np.array(val) for k, val in fx["inputs"].items() if isinstance(val, list)}
    f.update(nx=4, ny=4, dx=fx["inputs"]["dx"], dy=fx["inputs"]["dy"])
    L = mk(oracle, f, fx["bc"], fx["phys"], fx["alpha"], fx["beta"], max_box=2)
    L.gsrb(1)
    assert np.array_equal(L.get(oracle.F_PHI), np.array(fx["expected"]["phi_after_one_sweep"]))
    L2 = mk(oracle, f, fx["bc"], fx["phys"], fx["alpha"], fx["beta"], max_box=4)
    L2.residual()
    assert np.array_equal(L2.get(oracle.F_RES), np.array(fx["expected"]["residual_before"]))


def test_divergence_difterm_getflux(oracle):
    rng = np.random.default_rng(11)
    nx, ny, dx, dy = 12, 10, 0.5, 0.25
    ux, uy = rng.normal(size=(ny, nx + 1)), rng.normal(size=(ny + 1, nx))
    d = oracle.divergence(ux, uy, dx, dy)
    ref = np.zeros((ny, nx)) + (1.0 / dx) * (ux[:, 1:] - ux[:, :-1])
    ref = ref + (1.0 / dy) * (uy[1:, :] - uy[:-1, :])
    assert np.array_equal(d, ref)
    pg = rng.normal(size=(ny + 2, nx + 2))
    dt = oracle.difterm(pg, ux, uy, dx, dy)
    rdx, rdy = 1 / (dx * dx), 1 / (dy * dy)
    c = pg[1:-1, 1:-1]
    ref = (ux[:, 1:] * (pg[1:-1, 2:] - c) * rdx - ux[:, :-1] * (c - pg[1:-1, :-2]) * rdx
           + uy[1:, :] * (pg[2:, 1:-1] - c) * rdy - uy[:-1, :] * (c - pg[:-2, 1:-1]) * rdy)
    assert np.array_equal(dt, ref)
    fl = oracle.getflux(pg, ux, 0, -1.0, dx, 2)
    assert np.array_equal(fl, -ux * ((pg[1:-1, 1:] - pg[1:-1, :-1]) * (-1.0 * 2 / dx)))


def test_fas_vcycle_converges_and_depth_rule(oracle):
    f = sy.shmip_fields(128, 64)
    L = mk(oracle, f, sy.A3_BC, sy.A3_PHYS, max_box=64)
    assert L.ndepth == 6          # 64^2 boxes -> depths 0..5
    L2 = mk(oracle, f, sy.A3_BC, sy.A3_PHYS, max_box=32)
    assert L2.ndepth == 5
    L.build_mg_coefficients()
    n, hist = L.solve(dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-12))
    assert n >= 2 and hist[-1] < 1e-3 * hist[0]
    assert all(hist[k + 1] < hist[k] for k in range(1, len(hist) - 1))
