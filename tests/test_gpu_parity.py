"""Parity tests proper: every HIP operator is driven through the C-ABI (libsuhmo_hip.so)
and compared with the CPU oracle on the same seeded inputs.  Bar: BITWISE equality for
every fp64 field (both sides are compiled with FP contraction off and follow the
reference's expression association); the only tolerance is on the l2 norm, whose
summation order differs (1e-12 relative, stated below)."""
import os

import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from suhmo_amd import capi, level
    assert capi.lib().suhmo_device_count() > 0, "no GPU visible: the product path has no fallback"
    return level


def pair(oracle, hip, f, bc, ph, alpha=0.0, beta=-1.0, max_box=16):
    O = oracle.OracleLevel(f["nx"], f["ny"], f["dx"], f["dy"], bc, ph, alpha, beta, max_box, 2)
    G = hip.HipLevel(f["nx"], f["ny"], f["dx"], f["dy"], bc, ph, alpha, beta, max_box)
    O.set_inputs(f)
    G.set_inputs(f)
    return O, G


CASES = [
    ("random-mixedbc", lambda: sy.random_fields(48, 32), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.7, -1.0, 16),
    ("random-yperiodic", lambda: sy.random_fields(64, 32, seed=3), sy.CONV_BC, sy.RANDOM_PHYS, 0.0, -1.0, 16),
    ("random-allperiodic", lambda: sy.random_fields(32, 32, seed=5),
     dict(type=[[0, 0], [0, 0]], value=[[0, 0], [0, 0]], periodic=[1, 1]), sy.RANDOM_PHYS, 0.25, -1.0, 8),
    ("ragged-odd", lambda: sy.random_fields(50, 34, seed=9), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, 64),
    ("shmip-a3", lambda: sy.shmip_fields(128, 64), sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64),
    ("no-nl", lambda: sy.random_fields(32, 16, seed=2), sy.A3_BC, dict(sy.RANDOM_PHYS, use_NL=0), 0.0, -1.0, 16),
]
IDS = [c[0] for c in CASES]


def prep(oracle, hip, case, need_b=True):
    _, mk, bc, ph, alpha, beta, mb = case
    f = mk()
    O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
    if need_b and "bx" not in f:
        O.update_operator()
        G.update_operator()
    return f, O, G


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_nonlinear_and_lambda(oracle, hip, case):
    f, O, G = prep(oracle, hip, case)
    O.nonlinear(); G.nonlinear()
    O.reset_lambda(); G.compute_lambda()
    assert np.array_equal(G.get(hip.F_NL), O.get(oracle.F_NL))
    assert np.array_equal(G.get(hip.F_DNL), O.get(oracle.F_DNL))
    assert np.array_equal(G.get(hip.F_LAMBDA), O.get(oracle.F_LAMBDA))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_apply_and_residual(oracle, hip, case):
    f, O, G = prep(oracle, hip, case)
    for homog in (False, True):
        O.apply_op(homog); G.apply_op(homog)
        assert np.array_equal(G.get(hip.F_LPHI), O.get(oracle.F_LPHI))
    O.residual(); G.residual()
    assert np.array_equal(G.get(hip.F_RES), O.get(oracle.F_RES))
    assert G.norm(hip.F_RES, 0) == O.norm(oracle.F_RES, 0)
    assert abs(G.norm(hip.F_RES, 2) - O.norm(oracle.F_RES, 2)) <= 1e-12 * O.norm(oracle.F_RES, 2)


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("sweeps", [1, 4])
def test_gsrb(oracle, hip, case, sweeps):
    f, O, G = prep(oracle, hip, case)
    O.gsrb(sweeps); G.gsrb(sweeps)
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))
    # ghosts are left with the HOMOGENEOUS BC applied (VCAMRNonLinearPoissonOp.cpp:757-759)
    go, gg = O.get(oracle.F_PHI, ghosted=True), G.get(hip.F_PHI, ghosted=True)
    bc = case[2]
    if not bc["periodic"][0]:
        assert np.array_equal(gg[1:-1, 0], go[1:-1, 0]) and np.array_equal(gg[1:-1, -1], go[1:-1, -1])
    if not bc["periodic"][1]:
        assert np.array_equal(gg[0, 1:-1], go[0, 1:-1]) and np.array_equal(gg[-1, 1:-1], go[-1, 1:-1])


FUSED_CASES = [
    ("wide-3strips", lambda: sy.random_fields(1100, 48, seed=21), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.7, -1.0, 2048),
    ("allperiodic", lambda: sy.random_fields(128, 96, seed=22),
     dict(type=[[0, 0], [0, 0]], value=[[0, 0], [0, 0]], periodic=[1, 1]), sy.RANDOM_PHYS, 0.0, -1.0, 32),
    ("yperiodic-tall", lambda: sy.random_fields(64, 200, seed=23), sy.CONV_BC, sy.RANDOM_PHYS, 0.0, -1.0, 8),
    ("shmip-512", lambda: sy.shmip_fields(512, 256), sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64),
]


@pytest.mark.parametrize("case", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
@pytest.mark.parametrize("variant,hc", [(0, 0), (1, 0), (1, 16), (2, 0), (2, 32)])
@pytest.mark.parametrize("sweeps", [1, 2, 5])
def test_gsrb_fused_variants(oracle, hip, case, variant, hc, sweeps, monkeypatch):
    """every relaxation kernel variant (two-pass, fused red+black, two sweeps fused; several
    chunk heights = many workgroup seams) gives the oracle's phi bit for bit"""
    monkeypatch.setenv("SUHMO_GSRB_VARIANT", str(variant))
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    monkeypatch.setenv("SUHMO_FUSED_HC", str(hc))
    f, O, G = prep(oracle, hip, case)
    O.gsrb(sweeps); G.gsrb(sweeps)
    a, b = G.get(hip.F_PHI), O.get(oracle.F_PHI)
    assert np.array_equal(a, b), "mismatch at %s" % (np.argwhere(a != b)[:5],)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_restrict_prolong(oracle, hip, case):
    f, O, G = prep(oracle, hip, case)
    if O.ndepth < 2:
        pytest.skip("boxes not coarsenable")
    assert G.ndepth == O.ndepth
    O.build_mg_coefficients(); G.build_mg_coefficients()
    for fid in (oracle.F_ACOEF, oracle.F_B, oracle.F_PI, oracle.F_ZB, oracle.F_MASK, oracle.F_BX, oracle.F_BY):
        for d in range(1, O.ndepth):
            assert np.array_equal(G.get(fid, depth=d), O.get(fid, depth=d)), (fid, d)
    O.restrict_residual(); G.restrict_residual()
    assert np.array_equal(G.get(hip.F_RES, depth=1), O.get(oracle.F_RES, depth=1))
    O.restrict_r(); G.restrict_r()
    assert np.array_equal(G.get(hip.F_PHI, depth=1), O.get(oracle.F_PHI, depth=1))
    rng = np.random.default_rng(1)
    corr = rng.normal(size=O.shape(oracle.F_PHI, 1))
    O.prolong_increment(corr)
    G.set(hip.F_CORR, corr, depth=1); G.prolong_increment()
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))
    # PROLONG_2_NL with a ghosted coarse correction
    cg = rng.normal(size=O.shape(oracle.F_PHI, 1, ghosted=True))
    fine0 = G.get(hip.F_PHI)
    G.set(hip.F_CORR, cg, depth=1, ghosted=True); G.prolong_bilinear()
    assert np.array_equal(G.get(hip.F_PHI), oracle.prolong2(fine0, cg))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_update_and_average_operator(oracle, hip, case):
    _, mk, bc, ph, alpha, beta, mb = case
    f = mk()
    O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
    O.update_operator(); G.update_operator()
    assert np.array_equal(G.get(hip.F_BX), O.get(oracle.F_BX))
    assert np.array_equal(G.get(hip.F_BY), O.get(oracle.F_BY))
    for d in range(1, O.ndepth):
        O.average_operator(d); G.average_operator(d)
        assert np.array_equal(G.get(hip.F_BX, depth=d), O.get(oracle.F_BX, depth=d))
        assert np.array_equal(G.get(hip.F_BY, depth=d), O.get(oracle.F_BY, depth=d))


def test_divergence_flux_axby(oracle, hip):
    f = sy.random_fields(40, 24, seed=4)
    O, G = pair(oracle, hip, f, sy.RANDOM_BC, sy.RANDOM_PHYS)
    d0 = np.random.default_rng(0).normal(size=(24, 40))
    G.set(hip.F_LPHI, d0)
    G.divergence(hip.F_LPHI)
    assert np.array_equal(G.get(hip.F_LPHI), oracle.divergence(f["bx"], f["by"], f["dx"], f["dy"], d0))
    O.bc(oracle.F_PHI, False)
    pg = O.get(oracle.F_PHI, ghosted=True)
    for direction, b, h in ((0, f["bx"], f["dx"]), (1, f["by"], f["dy"])):
        assert np.array_equal(G.get_flux(direction, ref=2), oracle.getflux(pg, b, direction, -1.0, h, 2))
    G.axby(hip.F_RES, hip.F_PHI, hip.F_RHS, 1.0, -1.0)
    assert np.array_equal(G.get(hip.F_RES), f["phi"] - f["rhs"])


def test_box_traffic_roundtrip(oracle, hip):
    # LevelData<FArrayBox> drop-in: scatter per-box fabs (1 ghost), gather them back with ghosts
    nx, ny, mb = 32, 16, 8
    f = sy.random_fields(nx, ny, seed=8)
    boxes = [(bi * mb, bj * mb, bi * mb + mb - 1, bj * mb + mb - 1) for bj in range(ny // mb) for bi in range(nx // mb)]
    G = hip.HipLevel(nx, ny, f["dx"], f["dy"], sy.RANDOM_BC, sy.RANDOM_PHYS, boxes=boxes)
    Bg = f["B"]
    for k, (l0, l1, h0, h1) in enumerate(boxes):
        fab = np.full((mb + 2, mb + 2), np.nan)           # stale interior ghosts must be ignored
        fab[1:-1, 1:-1] = f["phi"][l1:h1 + 1, l0:h0 + 1]
        G.put_box(hip.F_PHI, k, fab, (l0 - 1, l1 - 1), (h0 + 1, h1 + 1))
        G.put_box(hip.F_B, k, Bg[l1:h1 + 3, l0:h0 + 3], (l0 - 1, l1 - 1), (h0 + 1, h1 + 1), with_domain_ghosts=True)
    assert np.array_equal(G.get(hip.F_PHI), f["phi"])
    gb = G.get(hip.F_B, ghosted=True)
    assert np.array_equal(gb[1:-1, :], Bg[1:-1, :]) and np.array_equal(gb[:, 1:-1], Bg[:, 1:-1])
    G.fill_ghosts(hip.F_PHI, False)
    O = oracle.OracleLevel(nx, ny, f["dx"], f["dy"], sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, mb, 1)
    O.set_inputs(f); O.exchange(oracle.F_PHI); O.bc(oracle.F_PHI, False)
    og = O.get(oracle.F_PHI, ghosted=True)
    l0, l1, h0, h1 = boxes[0]
    fab = G.get_box(hip.F_PHI, 0, (l0 - 1, l1 - 1), (h0 + 1, h1 + 1))
    assert np.array_equal(fab[1:, 1:][:-1, :-1], f["phi"][l1:h1 + 1, l0:h0 + 1])
    assert np.array_equal(fab[1:-1, 0], og[1:mb + 1, 0])          # x-lo domain ghosts (Dirichlet)
    assert np.array_equal(fab[1:-1, -1], f["phi"][l1:h1 + 1, h0 + 1])  # interior ghost = neighbour's valid cells


SOLVE_CASES = [
    ("shmip-a3", lambda: sy.shmip_fields(128, 64), sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64),
    ("conv-yperiodic", lambda: sy.shmip_fields(256, 64, lx=8.0e4), sy.CONV_BC, dict(sy.A3_PHYS, A=2.5e-25), 0.0, -1.0, 32),
]


@pytest.mark.parametrize("case", SOLVE_CASES, ids=[c[0] for c in SOLVE_CASES])
def test_vcycle_and_solve(oracle, hip, case):
    _, mk, bc, ph, alpha, beta, mb = case
    f = mk()
    f.pop("bx", None); f.pop("by", None)
    O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=6, imin=6)
    O.vcycle(sp); G.vcycle(sp)
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))      # one V-cycle: bitwise
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert ng == no
    assert np.array_equal(hg, ho)
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))      # converged head: bitwise


FUSED_VCYCLE_CASES = [
    ("wide-3strips", lambda: sy.random_fields(1100, 48, seed=21), sy.RANDOM_BC, dict(sy.RANDOM_PHYS), 0.0, -1.0, 16),
    ("allperiodic", lambda: sy.random_fields(128, 96, seed=22),
     dict(type=[[0, 0], [0, 0]], value=[[0, 0], [0, 0]], periodic=[1, 1]), sy.RANDOM_PHYS, 0.0, -1.0, 32),
    ("yperiodic-tall", lambda: sy.random_fields(64, 200, seed=23), sy.CONV_BC, sy.RANDOM_PHYS, 0.0, -1.0, 8),
    ("shmip-512", lambda: sy.shmip_fields(512, 256), sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64),
    ("mixedbc-helmholtz", lambda: sy.random_fields(256, 64, seed=24), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.6, -1.0, 16),   # alpha != 0: no fused restriction
]


@pytest.mark.parametrize("case", FUSED_VCYCLE_CASES, ids=[c[0] for c in FUSED_VCYCLE_CASES])
@pytest.mark.parametrize("fused_restrict,rhs_in_relax", [(1, 3), (0, 3), (1, 0)], ids=["restrict+rhs", "rhs", "restrict"])
@pytest.mark.parametrize("hc", [0, 6, 10])
def test_vcycle_on_fused_kernels(oracle, hip, case, fused_restrict, rhs_in_relax, hc, monkeypatch):
    """every depth on the streaming kernel (SUHMO_FUSED_MIN_CELLS = 1), with the restriction fused into the launch that
    ends the pre-smoothing and with the separate restriction kernel, the FAS right-hand side of a coarse depth formed by the
    first launch of its pre-smoothing or by its own kernel: all bitwise equal to the oracle's V-cycle; chunk heights
    6 / 10 / automatic move the coarse cells' row pairs relative to the chunk boundaries"""
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    monkeypatch.setenv("SUHMO_FUSED_RESTRICT", str(fused_restrict))
    monkeypatch.setenv("SUHMO_FAS_RHS_IN_RELAX", str(rhs_in_relax))
    monkeypatch.setenv("SUHMO_FUSED_HC", str(hc))
    _, mk, bc, ph, alpha, beta, mb = case
    f = mk()
    f.pop("bx", None); f.pop("by", None)
    O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=6)
    for k in range(2):
        O.vcycle(sp); G.vcycle(sp)
        assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)), (k, float(np.max(np.abs(G.get(hip.F_PHI) - O.get(oracle.F_PHI)))))
    for d in range(1, G.ndepth):
        assert np.array_equal(G.get(hip.F_RES, depth=d), O.get(oracle.F_RES, depth=d)), ("coarse residual", d)
        assert np.array_equal(G.get(hip.F_RHS, depth=d), O.get(oracle.F_RHS, depth=d)), ("coarse right-hand side", d)
        assert np.array_equal(G.get(hip.F_PHIOLD, depth=d), O.get(oracle.F_PHIOLD, depth=d)), ("R phi kept for the prolongation", d)
        assert np.array_equal(G.get(hip.F_PHI, depth=d), O.get(oracle.F_PHI, depth=d)), ("coarse phi", d)
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert ng == no and np.array_equal(hg, ho)
    if case[0] in ("wide-3strips", "allperiodic", "shmip-512"):          # (depth 1 is wide enough for the streaming kernel there, alpha = 0)
        assert (G.get_option("rhs_in_streaming_launches") > 0) == (rhs_in_relax == 3), G.get_option("rhs_in_streaming_launches")
    # the solve loop's residual evaluation: left behind by the launch that ends each V-cycle (alpha = 0, two-sweep launches), its own pass otherwise
    assert np.array_equal(G.get(hip.F_RES), O.get(oracle.F_RES)), "residual of the converged head"
    assert (G.get_option("residual_in_relax_launches") > 0) == (alpha == 0.0), G.get_option("residual_in_relax_launches")


TILE_CASES = CASES + FUSED_VCYCLE_CASES[1:] + [
    ("tiny-mixedbc", lambda: sy.random_fields(8, 6, seed=31), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, 8),
    ("odd-rows-70x51", lambda: sy.random_fields(70, 51, seed=32), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, 64),
    ("allperiodic-smaller-than-halo", lambda: sy.random_fields(12, 10, seed=33),
     dict(type=[[0, 0], [0, 0]], value=[[0, 0], [0, 0]], periodic=[1, 1]), sy.RANDOM_PHYS, 0.3, -1.0, 4),
    ("yperiodic-33-tiles", lambda: sy.random_fields(66, 36, seed=34), sy.CONV_BC, sy.RANDOM_PHYS, 0.0, -1.0, 64),
]


@pytest.mark.parametrize("case", TILE_CASES, ids=[c[0] for c in TILE_CASES])
@pytest.mark.parametrize("tile_t", [16, 32])
def test_gsrb_tile_kernel(oracle, hip, case, tile_t, monkeypatch):
    """cache-resident depths: S = 4 / 2 / 1 sweeps per launch on LDS tiles with a 2S halo (k_gsrb_tile), tile edge 16 and
    32: bitwise the colour passes, for every sweep count's decomposition, domains that are not a multiple of the tile,
    odd row counts, periodic domains smaller than tile + halo (the halo then holds several images of a cell)"""
    monkeypatch.setenv("SUHMO_GSRB_TILE", "1")
    monkeypatch.setenv("SUHMO_TILE_T", str(tile_t))
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "100000000")
    for sweeps in (1, 2, 3, 4, 7, 9, 16):           # 9, 16: a level that is one tile does them in one launch (2 / 4 chunks of 4)
        f, O, G = prep(oracle, hip, case)
        O.gsrb(sweeps); G.gsrb(sweeps)
        assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)), sweeps
        go, gg = O.get(oracle.F_PHI, ghosted=True), G.get(hip.F_PHI, ghosted=True)
        if not case[2]["periodic"][0]:
            assert np.array_equal(gg[1:-1, 0], go[1:-1, 0]) and np.array_equal(gg[1:-1, -1], go[1:-1, -1])
        if not case[2]["periodic"][1]:
            assert np.array_equal(gg[0, 1:-1], go[0, 1:-1]) and np.array_equal(gg[-1, 1:-1], go[-1, 1:-1])


@pytest.mark.parametrize("case", FUSED_VCYCLE_CASES + SOLVE_CASES, ids=[c[0] for c in FUSED_VCYCLE_CASES + SOLVE_CASES])
@pytest.mark.parametrize("tile,fused_restrict,tile_t,rhs_in_relax", [(1, 1, 0, 1), (1, 1, 32, 1), (1, 1, 16, 1), (1, 0, 0, 1), (1, 1, 0, 0), (0, 1, 0, 1)],
                         ids=["tile+restrict+rhs", "tile32+restrict+rhs", "tile16+restrict+rhs", "tile+rhs", "tile+restrict", "colour-passes"])
def test_vcycle_on_tile_kernels(oracle, hip, case, tile, fused_restrict, tile_t, rhs_in_relax, monkeypatch):
    """V-cycles and a solve with every depth on the tile kernel (prolongation fused into its load, restriction into the
    launch that ends the pre-smoothing or as a separate kernel, the FAS right-hand side of a coarse depth formed by its first
    relaxation or by its own kernel) and with the tile kernel off (colour passes): all bitwise the oracle's"""
    monkeypatch.setenv("SUHMO_GSRB_TILE", str(tile))
    monkeypatch.setenv("SUHMO_FUSED_RESTRICT", str(fused_restrict))
    monkeypatch.setenv("SUHMO_TILE_RESTRICT", str(fused_restrict))      # (default 0: a separate kernel)
    monkeypatch.setenv("SUHMO_TILE_T", str(tile_t))
    monkeypatch.setenv("SUHMO_FAS_RHS_IN_RELAX", str(rhs_in_relax))
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "100000000")
    _, mk, bc, ph, alpha, beta, mb = case
    f = mk()
    f.pop("bx", None); f.pop("by", None)
    O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=6)
    for k in range(5):                                                   # from the second on: captured / replayed HIP graphs (two ping-pong states)
        O.vcycle(sp); G.vcycle(sp)
        assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)), (k, float(np.max(np.abs(G.get(hip.F_PHI) - O.get(oracle.F_PHI)))))
    for d in range(1, G.ndepth):
        assert np.array_equal(G.get(hip.F_RES, depth=d), O.get(oracle.F_RES, depth=d)), ("coarse residual", d)
        assert np.array_equal(G.get(hip.F_RHS, depth=d), O.get(oracle.F_RHS, depth=d)), ("coarse right-hand side", d)
        assert np.array_equal(G.get(hip.F_PHI, depth=d), O.get(oracle.F_PHI, depth=d)), ("coarse phi", d)
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert ng == no and np.array_equal(hg, ho)
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))


def _random_tile_case(seed):
    rng = np.random.default_rng(1000 + seed)
    nx = 2 * int(rng.integers(1, 60))
    per = [int(rng.integers(0, 2)), int(rng.integers(0, 2))]
    ny = int(rng.integers(1, 90))
    if per[1]:
        ny += ny & 1                                         # periodic y: even number of rows (the colour of an image)
    bc = dict(type=[[int(rng.integers(0, 2)), int(rng.integers(0, 2))], [int(rng.integers(0, 2)), int(rng.integers(0, 2))]],
              value=[[float(rng.uniform(-5, 5)), float(rng.uniform(-0.05, 0.05))], [float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-5, 5))]],
              periodic=per)
    alpha = float(rng.choice([0.0, 0.0, 0.37]))
    return nx, ny, bc, alpha, int(rng.integers(1, 12)), int(rng.choice([0, 16, 32]))


@pytest.mark.parametrize("seed", range(24))
def test_tile_kernel_random_shapes(hip, seed, monkeypatch):
    """tile kernel against the colour passes (both on the device) on random level shapes (2..118 x 1..90, narrower and lower
    than a tile, ragged last tiles), boundary types / values, periodicity, alpha and sweep counts: bitwise, ghosts included"""
    nx, ny, bc, alpha, sweeps, tile_t = _random_tile_case(seed)
    f = sy.random_fields(nx, ny, seed=2000 + seed)
    out = []
    for tile in (0, 1):
        monkeypatch.setenv("SUHMO_GSRB_TILE", str(tile))
        monkeypatch.setenv("SUHMO_TILE_T", str(tile_t))
        monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "100000000")
        G = hip.HipLevel(nx, ny, f["dx"], f["dy"], bc, sy.RANDOM_PHYS, alpha, -1.0, 64)
        G.set_inputs(f)
        G.gsrb(sweeps)
        out.append(G.get(hip.F_PHI, ghosted=True))
        G.close()
    assert np.array_equal(out[0][1:-1, 1:-1], out[1][1:-1, 1:-1]), (nx, ny, bc, alpha, sweeps, tile_t)
    assert np.array_equal(out[0][1:-1, :], out[1][1:-1, :]) and np.array_equal(out[0][:, 1:-1], out[1][:, 1:-1])


def _random_stream_case(seed):
    rng = np.random.default_rng(7000 + seed)
    nx = 4 * int(rng.integers(32, 100))                      # (two streaming depths: nx / 2 even and >= 64 as well)
    per = [int(rng.integers(0, 2)), int(rng.integers(0, 2))]
    ny = 4 * int(rng.integers(8, 40))
    bc = dict(type=[[int(rng.integers(0, 2)), int(rng.integers(0, 2))], [int(rng.integers(0, 2)), int(rng.integers(0, 2))]],
              value=[[float(rng.uniform(-5, 5)), float(rng.uniform(-0.05, 0.05))], [float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-5, 5))]],
              periodic=per)
    return nx, ny, bc, int(rng.choice([0, 6, 10, 14])), bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(12))
def test_streaming_cycle_random_shapes(oracle, hip, seed, monkeypatch):
    """the streaming kernel with everything it carries -- prolongation while loading, the FAS right-hand side of a coarse depth in its
    first launch, restriction and the solve loop's residual + partial norms in its last -- on random shapes, boundary conditions, chunk heights
    and ice-free patches (so that some V-cycles read the mask array and some skip it): V-cycles, solve history and the residual field are the
    oracle's, bit for bit"""
    nx, ny, bc, hc, icefree = _random_stream_case(seed)
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    monkeypatch.setenv("SUHMO_FUSED_HC", str(hc))
    f = sy.random_fields(nx, ny, seed=7100 + seed)
    f.pop("bx", None); f.pop("by", None)
    if not icefree:
        f["mask"][:] = 1.0
    ph = dict(sy.RANDOM_PHYS)
    O, G = pair(oracle, hip, f, bc, ph, 0.0, -1.0, 64)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-12, norm_thresh=1e-14, max_iter=3, imin=6)
    for k in range(2):
        O.vcycle(sp); G.vcycle(sp)
        assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)), (seed, nx, ny, bc, hc, k)
    for d in range(1, G.ndepth):
        assert np.array_equal(G.get(hip.F_RHS, depth=d), O.get(oracle.F_RHS, depth=d)), (seed, "coarse right-hand side", d)
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert ng == no and np.array_equal(hg, ho), (seed, nx, ny, bc, hc, hg, ho)
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)) and np.array_equal(G.get(hip.F_RES), O.get(oracle.F_RES)), (seed, nx, ny, bc, hc)
    assert G.get_option("residual_in_relax_launches") > 0 and (G.ndepth < 2 or G.get_option("rhs_in_streaming_launches") > 0)


def test_full_size_properties(hip):
    """BASELINE size (4096^2): size-independent properties instead of the (slow) oracle:
    GSRB fixed point, residual == rhs - applyOp, restriction of a constant, idempotent
    operator update."""
    n = 4096
    f = sy.shmip_fields(n, n)
    G = hip.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS)
    G.set_inputs(f)
    G.update_operator()
    bx1 = G.get(hip.F_BX)
    G.update_operator()
    assert np.array_equal(bx1, G.get(hip.F_BX))
    G.apply_op()
    lphi = G.get(hip.F_LPHI)
    G.residual()
    assert np.array_equal(G.get(hip.F_RES), f["rhs"] - lphi)
    G.set(hip.F_RHS, lphi)                       # rhs = L(phi)  ->  fixed point of the relaxation
    G.gsrb(2)
    out = G.get(hip.F_PHI)
    assert np.max(np.abs(out - f["phi"]) / np.abs(f["phi"])) < 1e-11
    G.set(hip.F_PHI, np.full((n, n), 2.5))
    G.restrict_r()
    assert np.all(G.get(hip.F_PHI, depth=1) == 2.5)
    assert G.ndepth == 6


def test_bench_size_vcycle_bitwise(oracle, hip):
    """The headline configuration at its own size: bench.py's workload (4096 x 4096 cells on square cells of 24.4 m, 64^2 boxes,
    6 depths, kernel options as the library picks them -- the occupancy-derived chunk height, one-wave strips, the restricting
    launch, the mask skip, the tile kernel at depths 1-5) against the oracle: two V-cycles, the head after each and every coarse
    residual after the second, bit for bit."""
    n = 4096
    f = sy.shmip_fields(n, n, ly=1.0e5)
    f.pop("bx", None); f.pop("by", None)
    O = oracle.OracleLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64, nthreads=min(16, os.cpu_count() or 1))
    G = hip.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64)
    O.set_inputs(f); G.set_inputs(f)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    assert G.ndepth == 6
    sp = dict(sy.SOLVER_DEFAULT)
    for k in range(2):
        O.vcycle(sp); G.vcycle(sp)
        a, b = G.get(hip.F_PHI), O.get(oracle.F_PHI)
        assert np.array_equal(a, b), (k, float(np.max(np.abs(a - b))))
    for d in range(1, G.ndepth):
        assert np.array_equal(G.get(hip.F_RES, depth=d), O.get(oracle.F_RES, depth=d)), ("coarse residual", d)
    O.residual(); G.residual()
    assert np.array_equal(G.get(hip.F_RES), O.get(oracle.F_RES))
    O.close(); G.close()


@pytest.mark.parametrize("max_box", [64, 1024], ids=["max-box-64", "max-box-is-the-level"])
def test_converged_solve_bitwise(oracle, hip, max_box):
    """bench.py's `converged_solve` at 1024^2: suhmo_level_solve from SHMIP-A's initial head to the reference's tolerances of a step >= 50
    (src/AmrHydro.cpp:737-762) with the headline's 64^2 boxes (6 depths, the cycle bottoms out at 32^2 cells) and with max_box_size = the
    level (exec/A_SHMIP/A3/input.hydro:71-72 is an ordinary input: 10 depths down to 2^2 cells) -- cycle count, residual history and head
    against the oracle, bit for bit"""
    n = 1024
    f = sy.shmip_fields(n, n, ly=1.0e5)
    f.pop("bx", None); f.pop("by", None)
    O = oracle.OracleLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=max_box, nthreads=min(16, os.cpu_count() or 1))
    G = hip.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=max_box)
    O.set_inputs(f); G.set_inputs(f)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    assert G.ndepth == O.ndepth == (6 if max_box == 64 else 10)
    sp = dict(sy.SOLVER_DEFAULT)
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (no, ng, ho[-3:], hg[-3:])
    assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI))
    if max_box == n:
        assert hg[-1] <= sp["norm_thresh"] and ng < 40, (ng, hg[-1])     # the deep cycle converges; with 64^2 boxes it contracts slowly
    O.close(); G.close()


@pytest.mark.parametrize("case", FUSED_VCYCLE_CASES[:4], ids=[c[0] for c in FUSED_VCYCLE_CASES[:4]])
def test_residual_left_behind_by_the_last_launch_equals_the_residual_pass(oracle, hip, case, monkeypatch):
    """suhmo_level_solve with the residual riding on the cycle's last launch (level option resid_in_relax, default 1) and with the
    separate pass: the same history, head and residual field, all the oracle's; chunk height 6 puts chunk seams everywhere"""
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    monkeypatch.setenv("SUHMO_FUSED_HC", "6")
    _, mk, bc, ph, alpha, beta, mb = case
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=4, imin=6)
    got = []
    for on in (1, 0):
        f = mk()
        f.pop("bx", None); f.pop("by", None)
        O, G = pair(oracle, hip, f, bc, ph, alpha, beta, mb)
        G.set_option("resid_in_relax", on)
        O.build_mg_coefficients(); G.build_mg_coefficients()
        no, ho = O.solve(sp)
        ng, hg = G.solve(sp)
        assert ng == no and np.array_equal(hg, ho), (on, hg, ho)
        assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)) and np.array_equal(G.get(hip.F_RES), O.get(oracle.F_RES)), on
        assert (G.get_option("residual_in_relax_launches") > 0) == bool(on)
        got.append(G.get(hip.F_RES))
    assert np.array_equal(got[0], got[1])


def test_kernel_selection_through_the_option_api(oracle):
    """suhmo_level_set_option replaces the SUHMO_* environment variables as the way to pick kernels: the same V-cycle on colour
    passes, tiles and the streaming kernel gives the same bits; unknown keys and bad values are refused"""
    from suhmo_amd import level, capi
    f = sy.shmip_fields(256, 256, ly=1.0e5)
    res = []
    for opts in (dict(), dict(gsrb_tile=0, gsrb_variant=0), dict(gsrb_variant=2, fused_min_cells=0), dict(tile_t=16, tile_s=2)):
        G = level.HipLevel(256, 256, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64)
        for k, v in opts.items():
            G.set_option(k, v)
            assert G.get_option(k) == v
        G.set_inputs(f); G.build_mg_coefficients()
        G.vcycle(sy.SOLVER_DEFAULT); G.vcycle(sy.SOLVER_DEFAULT)
        res.append(G.get(level.F_PHI))
        with pytest.raises(capi.SuhmoError):
            G.set_option("no_such_knob", 1)
        with pytest.raises(capi.SuhmoError):
            G.set_option("tile_t", 24)
        G.close()
    for r in res[1:]:
        assert np.array_equal(res[0], r)


def test_cfg2_shmip_a3_on_1024x1024_bitwise(oracle):
    """configs[1] of BASELINE.json literally: SHMIP A3 (100 km x 20 km, steady distributed source) on a 1024 x 1024 single level
    with 64 x 64 boxes -- operator update, two V-cycles and the residual history of a short solve against the oracle, bit for bit"""
    from suhmo_amd import level
    f = sy.shmip_fields(1024, 1024)
    O = oracle.OracleLevel(1024, 1024, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64, 8)
    G = level.HipLevel(1024, 1024, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64)
    O.set_inputs(f); G.set_inputs(f)
    O.build_mg_coefficients(); G.build_mg_coefficients()
    assert O.ndepth == G.ndepth == 6
    sp = dict(sy.SOLVER_DEFAULT, max_iter=3, imin=10, eps=1e-12, norm_thresh=1e-30)
    O.vcycle(sp); G.vcycle(sp)
    assert np.array_equal(O.get(oracle.F_PHI), G.get(level.F_PHI))
    assert np.array_equal(O.get(oracle.F_BX), G.get(level.F_BX)) and np.array_equal(O.get(oracle.F_BY), G.get(level.F_BY))
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng == 3 and np.array_equal(ho, hg), (ho, hg)
    assert np.array_equal(O.get(oracle.F_PHI), G.get(level.F_PHI))
    O.close(); G.close()


def test_single_precision_literals_of_the_fortran_kernels(oracle):
    """The .ChF kernels spell g and rho_w g as 9.8 and 1000.0 * 9.8 (src/AmrHydroF.ChF:45-52, 103, 217): a build whose Fortran
    compiler does not promote real literals to real*8 computes with float(9.8) = 9.80000019073486328125.  Both the library and
    the oracle take g and rho_w g as parameters: the nonlinear terms, the operator update (Re, bCoef) and a V-cycle with the
    single-precision-literal values are bitwise equal too, and differ from the double-literal results (1000.0 * 9.8 is 9800 exactly
    in single precision too, so only the terms with a bare 9.8 move: dNL, Re, bCoef)."""
    from suhmo_amd import level
    g32 = float(np.float32(9.8))
    ph32 = dict(sy.RANDOM_PHYS, grav=g32, rho_w_g=float(np.float32(1000.0) * np.float32(9.8)))
    f = sy.random_fields(96, 64)
    out = {}
    for name, ph in (("f64", sy.RANDOM_PHYS), ("f32", ph32)):
        O = oracle.OracleLevel(96, 64, f["dx"], f["dy"], sy.RANDOM_BC, ph, 0.0, -1.0, 32, 2)
        G = level.HipLevel(96, 64, f["dx"], f["dy"], sy.RANDOM_BC, ph, max_box=32)
        O.set_inputs(f); G.set_inputs(f)
        O.build_mg_coefficients(); G.build_mg_coefficients()
        O.nonlinear(); G.nonlinear()
        assert np.array_equal(O.get(oracle.F_NL), G.get(level.F_NL)) and np.array_equal(O.get(oracle.F_DNL), G.get(level.F_DNL))
        O.update_operator(); G.update_operator()
        assert np.array_equal(O.get(oracle.F_BX), G.get(level.F_BX)) and np.array_equal(O.get(oracle.F_BY), G.get(level.F_BY))
        O.vcycle(sy.SOLVER_DEFAULT); G.vcycle(sy.SOLVER_DEFAULT)
        assert np.array_equal(O.get(oracle.F_PHI), G.get(level.F_PHI))
        out[name] = (G.get(level.F_DNL), G.get(level.F_BX), G.get(level.F_PHI))
        O.close(); G.close()
    assert not np.array_equal(out["f64"][0], out["f32"][0]) and not np.array_equal(out["f64"][1], out["f32"][1])
    assert np.max(np.abs(out["f64"][2] - out["f32"][2])) < 1e-5 * np.max(np.abs(out["f64"][2]))


def test_named_timers_report_the_reference_labels():
    """CH_TIME equivalent: with the timers on, a solve reports the reference's scope names (src/VCAMRNonLinearPoissonOp.cpp:40, 69, 103, 660)"""
    from suhmo_amd import level, capi
    f = sy.shmip_fields(128, 64)
    G = level.HipLevel(128, 64, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=32)
    G.set_inputs(f); G.build_mg_coefficients()
    capi.lib().suhmo_timers_reset(); capi.lib().suhmo_timers_enable(2)
    try:
        G.solve(dict(sy.SOLVER_DEFAULT, max_iter=2)); G.gsrb(1); G.residual()
        rep = capi.timers_report()
    finally:
        capi.lib().suhmo_timers_enable(0)
    for label in ("AMRFASMultiGrid::solve", "AMRFASMultiGrid::VCycle", "VCAMRNonLinearPoissonOp::levelGSRB", "VCAMRNonLinearPoissonOp::residualI"):
        assert label in rep, rep
    first = rep.splitlines()[0].split()
    assert first[0] == "AMRFASMultiGrid::solve" and int(first[1]) == 1 and float(first[2]) > 0.0
    G.close()


def test_mask_report_follows_the_mask(oracle, hip, monkeypatch):
    """the streaming relaxation skips the ice-mask array when the V-cycle's UpdateOperator found no negative cell (a device word
    written by k_bcoef_fused, valid for that cycle only): a mask that turns negative between two cycles is seen by the next one"""
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    f = sy.shmip_fields(512, 256)
    f.pop("bx", None); f.pop("by", None)
    assert (f["mask"][1:-1, 1:-1] > 0).all()          # (the ghost column beyond x = 0 is -1: the kernel's report looks at valid cells only)
    O, G = pair(oracle, hip, f, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64)
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=6)
    masks = [f["mask"].copy() for _ in range(3)]
    masks[1][60:140, 200:330] = -1.0                  # an ice-free island appears ...
    masks[2][:] = f["mask"]                           # ... and is gone again
    for k, m in enumerate(masks):
        O.set(oracle.F_MASK, m, ghosted=True); G.set(hip.F_MASK, m, ghosted=True)
        O.build_mg_coefficients(); G.build_mg_coefficients()
        for it in range(2):
            O.vcycle(sp); G.vcycle(sp)
            assert np.array_equal(G.get(hip.F_PHI), O.get(oracle.F_PHI)), (k, it)
    assert G.get_option("skip_mask") == 1
