"""Multi-rank path on a 1-GPU box: the level is cut into row strips, each strip is driven by
its own thread ("rank") through the same C-ABI + exchange hooks bench.py uses with RCCL,
and the gathered result must equal the single-level result (and hence the oracle) BIT FOR
BIT -- GSRB is colour-Jacobi, so the partition must not change a single bit."""
import threading

import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu


def split_fields(f, j0, ny):
    s = dict(nx=f["nx"], ny=ny, dx=f["dx"], dy=f["dy"])
    for k in ("phi", "rhs", "aCoef"):
        s[k] = f[k][j0:j0 + ny]
    for k in ("B", "Pi", "zb", "mask"):
        s[k] = f[k][j0:j0 + ny + 2]
    if "bx" in f:
        s["bx"] = f["bx"][j0:j0 + ny]
        s["by"] = f["by"][j0:j0 + ny + 1]
    return s


wrap_ghosts = sy.wrap_ghosts


def run_strips(world, f, bc, ph, alpha, beta, body, halo=4, max_box=32):
    """body(level, rank) runs on every rank-thread; returns list of per-rank results"""
    from suhmo_amd import level, multigpu
    ny_tot = f["ny"]
    assert ny_tot % world == 0
    ny = ny_tot // world
    tr = multigpu.ThreadTransport(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            j0 = rank * ny
            G = level.HipLevel(f["nx"], ny, f["dx"], f["dy"], bc, ph, alpha, beta, max_box, j0=j0,
                               ny_global=ny_tot, halo_rows=halo)
            G.set_inputs(split_fields(f, j0, ny))
            ex = multigpu.StripExchanger(G, tr, rank, world, bool(bc["periodic"][1]))
            ex.exchange_static()
            out[rank] = body(G, rank)
            G.synchronize()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)
            tr.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    return out


def single(f, bc, ph, alpha, beta, body, max_box=32):
    from suhmo_amd import level
    G = level.HipLevel(f["nx"], f["ny"], f["dx"], f["dy"], bc, ph, alpha, beta, max_box)
    G.set_inputs(f)
    return body(G, 0)


CASES = [
    ("random-2ranks", lambda: sy.random_fields(128, 128, seed=31), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.5, -1.0, 2),
    ("random-yperiodic-2ranks", lambda: sy.random_fields(128, 128, seed=32), sy.CONV_BC, sy.RANDOM_PHYS, 0.0, -1.0, 2),
    ("random-3ranks", lambda: sy.random_fields(192, 192, seed=33), sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, 3),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("halo", [4, 9, 16, 24])
def test_strips_gsrb_and_operators(case, variant, halo, monkeypatch):
    from suhmo_amd import level as lv
    monkeypatch.setenv("SUHMO_GSRB_VARIANT", str(variant))
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
    _, mk, bc, ph, alpha, beta, world = case
    f = wrap_ghosts(mk(), bc)

    def body(G, rank):
        G.gsrb(3)
        phi = G.get(lv.F_PHI)
        G.residual()
        res = G.get(lv.F_RES)
        G.update_operator()
        return phi, res, G.get(lv.F_BX), G.get(lv.F_BY), G.norm(lv.F_RES, 0)

    ref = single(f, bc, ph, alpha, beta, body)
    parts = run_strips(world, f, bc, ph, alpha, beta, body, halo=halo)
    assert np.array_equal(np.vstack([p[0] for p in parts]), ref[0])
    assert np.array_equal(np.vstack([p[1] for p in parts]), ref[1])
    assert np.array_equal(np.vstack([p[2] for p in parts]), ref[2])
    by = np.vstack([p[3][:-1] for p in parts] + [parts[-1][3][-1:]])
    assert np.array_equal(by, ref[3])
    assert all(p[4] == ref[4] for p in parts)       # MAX all-reduce of the norm


@pytest.mark.parametrize("world,halo,fused", [(2, 4, 0), (4, 4, 0), (2, 16, 0), (4, 9, 0), (2, 16, 1), (4, 12, 1), (2, 24, 0), (2, 24, 1), (4, 24, 1), (2, 1, 0),
                                              (2, 24, "no-tile"), (4, 12, "no-tile"), (2, 24, "rhs-exchanged"), (2, 24, "copy-readback"),
                                              (2, 16, "overlap"), (4, 12, "overlap"), (2, 16, "no-overlap"), (2, 24, "fused-restriction"), (4, 12, "fused-restriction")])
def test_strips_vcycle_and_solve(world, halo, fused, oracle, monkeypatch):
    """strips of 2 / 4 ranks against the oracle's whole level, bitwise.  Halo >= 10 rows: the tile kernel relaxes the strips
    (halo rows advanced redundantly), R phi + RES travel together and the halo rows' right-hand side is computed locally; the
    named variants keep the paths they replace covered (colour passes, exchanged RHS, copy + synchronise read-back).
    "overlap": streaming kernel with chunks tall enough that the halo exchange travels on the second stream while the inner
    chunks relax, the two end chunks after it (what a 4096^2 strip does); "no-overlap": the same launches, exchange first.
    (The peer-direct transport, whose kernels wait for the neighbour's kernels, is tested where ranks have queues of their own: as a strip that
    is its own neighbour, tests/test_gpu_rccl.py, and as processes, tests/test_gpu_multiproc.py.)"""
    from suhmo_amd import level as lv
    if fused == 1:
        monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "4000")     # fused K=2 launches down to 64 x 64 strips
        monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    elif fused == "no-tile":
        monkeypatch.setenv("SUHMO_TILE_STRIPS", "0")
    elif fused == "rhs-exchanged":
        monkeypatch.setenv("SUHMO_STRIPS_RHS_LOCAL", "0")
    elif fused == "fused-restriction":
        monkeypatch.setenv("SUHMO_TILE_RESTRICT", "1")       # the tile launch that ends the pre-smoothing also restricts (default: its own kernel)
    elif fused == "copy-readback":
        monkeypatch.setenv("SUHMO_POLL_READBACK", "0")
    elif fused in ("overlap", "no-overlap"):
        monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "4000")
        monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
        monkeypatch.setenv("SUHMO_FUSED_HC", "16" if world == 4 else "32")
        monkeypatch.setenv("SUHMO_OVERLAP_HALO", "2" if fused == "overlap" else "0")   # 2: also with a host transport (this one fences the device); 1 = native RCCL only
    f = sy.shmip_fields(256, 256)
    bc, ph = sy.A3_BC, sy.A3_PHYS
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=4, imin=4)

    def body(G, rank):
        G.build_mg_coefficients()
        G.vcycle(sp)
        p1 = G.get(lv.F_PHI)
        n, hist = G.solve(sp)
        return (p1, G.get(lv.F_PHI), n, hist, G.ndepth, G.get_option("overlapped_launches"), G.get_option("rhs_in_streaming_launches"),
                G.get_option("residual_in_relax_launches"), G.get(lv.F_RES))

    parts = run_strips(world, f, bc, ph, 0.0, -1.0, body, halo=halo, max_box=64)
    if fused == "overlap":
        assert all(p[5] > 0 for p in parts), [p[5] for p in parts]
    elif fused == "no-overlap":
        assert all(p[5] == 0 for p in parts)
    if fused == 1 and halo >= 12:
        # streaming kernel on the strips of every depth that is wide enough: the first launch of a coarse depth's pre-smoothing forms the
        # depth's FAS right-hand side, in the strip's halo rows too (R phi and RES arrived together)
        assert all(p[6] > 0 for p in parts), [p[6] for p in parts]
        # ... and the launch that ends each V-cycle of the solve leaves the residual of the strip's rows behind (one more halo row than the sweeps need)
        assert all(p[7] > 0 for p in parts), [p[7] for p in parts]
    O = oracle.OracleLevel(256, 256, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 64, 4)
    O.set_inputs(f)
    O.build_mg_coefficients()
    assert parts[0][4] == O.ndepth
    O.vcycle(sp)
    assert np.array_equal(np.vstack([p[0] for p in parts]), O.get(oracle.F_PHI))
    n, hist = O.solve(sp)
    assert all(p[2] == n for p in parts)
    assert np.array_equal(parts[0][3], hist)
    assert np.array_equal(np.vstack([p[1] for p in parts]), O.get(oracle.F_PHI))
    assert np.array_equal(np.vstack([p[8] for p in parts]), O.get(oracle.F_RES)), "residual of the converged head"


@pytest.mark.parametrize("where", ["none", "first-rows-of-rank-1", "inside-rank-1", "last-rows-of-rank-0"])
@pytest.mark.parametrize("world", [2, 4])
def test_strips_mask_report_covers_the_halo_rows(world, where, oracle, monkeypatch):
    """a strip's streaming relaxation skips the ice-mask array when UpdateOperator's report found no negative cell among the strip's
    own cells AND its stored halo rows on every depth (the neighbours' cells it reads): ice-free cells that sit only in a halo,
    only deep inside another rank, or nowhere -- all bitwise the oracle's whole level"""
    from suhmo_amd import level as lv
    monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "4000")
    monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
    f = sy.shmip_fields(256, 256)
    f.pop("bx", None); f.pop("by", None)
    ny = 256 // world
    m = f["mask"]                                           # ghosted: row r of the array is cell row r - 1
    if where == "first-rows-of-rank-1":
        m[1 + ny:1 + ny + 2, 40:90] = -1.0
    elif where == "inside-rank-1":
        m[1 + ny + ny // 2:1 + ny + ny // 2 + 3, 100:180] = -1.0
    elif where == "last-rows-of-rank-0":
        m[1 + ny - 3:1 + ny, 10:200] = -1.0
    bc, ph = sy.A3_BC, sy.A3_PHYS
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=3)

    def body(G, rank):
        G.build_mg_coefficients()
        G.vcycle(sp); G.vcycle(sp)
        return G.get(lv.F_PHI), G.get_option("skip_mask")

    parts = run_strips(world, f, bc, ph, 0.0, -1.0, body, halo=16, max_box=64)
    assert all(p[1] == 1 for p in parts)
    O = oracle.OracleLevel(256, 256, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 64, 4)
    O.set_inputs(f)
    O.build_mg_coefficients()
    O.vcycle(sp); O.vcycle(sp)
    assert np.array_equal(np.vstack([p[0] for p in parts]), O.get(oracle.F_PHI))


@pytest.mark.parametrize("world", [2, 4])
def test_strips_l2_norm_and_dot_product(world):
    """norm(ord 2) and dotProduct reduce with a SUM over the ranks (src/AMRNonLinearPoissonOp.cpp:660-666, 519-551, 1222-1264): on 2 / 4
    thread ranks through the reduce hook (suhmo_level_set_reduce_hook) equal to the whole level to 1e-12 (different summation
    order); with the MAX-only hook of suhmo_level_set_hooks alone the call is refused (rc -5), not answered per rank"""
    from suhmo_amd import level as lv, capi
    f = wrap_ghosts(sy.random_fields(128, 128, seed=41), sy.RANDOM_BC)

    def body(G, rank):
        G.residual()
        return G.norm(lv.F_RES, 2), G.dot(lv.F_RES, lv.F_PHI), G.norm(lv.F_RES, 0)

    ref = single(f, sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, body)
    parts = run_strips(world, f, sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, body)
    for p in parts:
        assert abs(p[0] - ref[0]) <= 1e-12 * abs(ref[0]) and abs(p[1] - ref[1]) <= 1e-12 * abs(ref[1]) and p[2] == ref[2]
    assert len({p[0] for p in parts}) == 1 and len({p[1] for p in parts}) == 1          # every rank holds the same bits
    # numpy twin of the definition
    G = lv.HipLevel(128, 128, f["dx"], f["dy"], sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, 32)
    G.set_inputs(f); G.residual()
    r, p = G.get(lv.F_RES), G.get(lv.F_PHI)
    assert abs(ref[0] - np.sqrt(np.sum(r * r))) <= 1e-12 * ref[0] and abs(ref[1] - np.sum(r * p)) <= 1e-12 * abs(np.sum(r * p))
    G.close()

    def body_max_only(G, rank):
        check = capi.check
        check(capi.lib().suhmo_level_set_reduce_hook(G.h, capi.REDUCE_FN(0)))           # back to the MAX-only hook
        G.residual()
        with pytest.raises(capi.SuhmoError):
            G.norm(lv.F_RES, 2)
        return G.norm(lv.F_RES, 0)

    parts = run_strips(2, f, sy.RANDOM_BC, sy.RANDOM_PHYS, 0.0, -1.0, body_max_only)
    assert parts[0] == parts[1] == ref[2]


@pytest.mark.parametrize("world,n,agg_min_cells,expect_da", [(2, 256, 40000, 1), (4, 256, 2000, 2), (2, 256, 1000, 3), (4, 512, 70000, 1), (2, 256, 100, 5),
                                                              (4, 256, 10, 0)])
@pytest.mark.parametrize("periodic", [0, 1])
def test_strips_agglomerated_coarse_depths(world, n, agg_min_cells, expect_da, periodic, oracle, monkeypatch):
    """multigrid depths whose strip holds fewer than agg_min_cells cells run redundantly on a whole-level copy on every
    rank (suhmo_agg.hip: all-gather of the coarse coefficients per build, of the coarse faces and of R phi + RES per V-cycle) -- from
    depth 1, 2, 3, from the bottom depth only, and not at all: V-cycle and solve equal the oracle's whole level bit for bit, y-periodic
    and not; the halo exchanges of the agglomerated depths are gone"""
    from suhmo_amd import level as lv, capi
    monkeypatch.setenv("SUHMO_AGG_MIN_CELLS", str(agg_min_cells))
    bc = sy.CONV_BC if periodic else sy.A3_BC
    f = wrap_ghosts(sy.shmip_fields(n, n, ly=1.0e5), bc)
    ph = sy.A3_PHYS
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=4)

    def body(G, rank):
        da = capi.lib().suhmo_level_agglomerated_depth(G.h)
        G.build_mg_coefficients()
        G.vcycle(sp)
        p1 = G.get(lv.F_PHI)
        n_, hist = G.solve(sp)
        return p1, G.get(lv.F_PHI), n_, hist, da, G.get_option("agg_gathers")

    parts = run_strips(world, f, bc, ph, 0.0, -1.0, body, halo=24, max_box=64)
    assert all(p[4] == expect_da for p in parts), [p[4] for p in parts]
    if expect_da:
        assert all(p[5] >= 3 for p in parts)               # coefficients + (faces, state) per V-cycle
    else:
        assert all(p[5] == 0 for p in parts)
    O = oracle.OracleLevel(n, n, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 64, 4)
    O.set_inputs(f)
    O.build_mg_coefficients()
    O.vcycle(sp)
    assert np.array_equal(np.vstack([p[0] for p in parts]), O.get(oracle.F_PHI))
    n_, hist = O.solve(sp)
    assert all(p[2] == n_ for p in parts)
    assert np.array_equal(parts[0][3], hist)
    assert np.array_equal(np.vstack([p[1] for p in parts]), O.get(oracle.F_PHI))


def test_strips_agglomerate_at_the_library_default_threshold(oracle, monkeypatch):
    """the suite switches the agglomeration off by default (tests/conftest.py); here the shipped default (agg_min_cells = 100000) decides: strips of
    512 x 256 cells keep depth 0 (131 k cells) and run every coarser depth on the whole-level copy -- V-cycle and solve equal the oracle bit for bit"""
    from suhmo_amd import level as lv, capi
    monkeypatch.delenv("SUHMO_AGG_MIN_CELLS", raising=False)
    n, world = 512, 2
    f = wrap_ghosts(sy.shmip_fields(n, n, ly=1.0e5), sy.A3_BC)
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=4)

    def body(G, rank):
        da = capi.lib().suhmo_level_agglomerated_depth(G.h)
        G.build_mg_coefficients()
        n_, hist = G.solve(sp)
        return G.get(lv.F_PHI), n_, hist, da, G.get_option("agg_min_cells")

    parts = run_strips(world, f, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, body, halo=24, max_box=64)
    assert all(p[4] == 100000 and p[3] == 1 for p in parts), [(p[3], p[4]) for p in parts]
    O = oracle.OracleLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64, 4)
    O.set_inputs(f); O.build_mg_coefficients()
    n_, hist = O.solve(sp)
    assert all(p[1] == n_ for p in parts) and np.array_equal(parts[0][2], hist)
    assert np.array_equal(np.vstack([p[0] for p in parts]), O.get(oracle.F_PHI))
    O.close()


@pytest.mark.parametrize("world,agg_min_cells,expect_da", [(2, 40000, 1), (4, 2000, 2)])
def test_agglomeration_set_after_the_coefficient_build_and_operator_changes_reach_it(world, agg_min_cells, expect_da, oracle):
    """the agglomerated copy created AFTER suhmo_build_mg_coefficients (set_option("agg_min_cells") on a level whose environment default is
    off): the cycle gathers the coarse coefficients before it enters the copy; suhmo_level_set_alpha_beta / set_bc through the C-ABI reach
    the agglomerated depths too (an operator with beta = -2 and other boundary values equals an oracle level created that way)"""
    import ctypes as C
    from suhmo_amd import level as lv, capi
    n = 256
    bc, bc2 = sy.A3_BC, dict(sy.A3_BC, value=[[3.0, 0.0], [0.0, 0.0]])
    f = wrap_ghosts(sy.shmip_fields(n, n, ly=1.0e5), bc)
    ph = sy.A3_PHYS
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-10, norm_thresh=1e-13, max_iter=3, imin=4)

    def body(G, rank):
        G.build_mg_coefficients()
        assert capi.lib().suhmo_level_agglomerated_depth(G.h) == 0
        G.set_option("agg_min_cells", agg_min_cells)
        da = capi.lib().suhmo_level_agglomerated_depth(G.h)
        G.vcycle(sp)
        p1 = G.get(lv.F_PHI)
        capi.check(capi.lib().suhmo_level_set_alpha_beta(G.h, 0.0, -2.0))
        b = lv._bc(bc2)
        capi.check(capi.lib().suhmo_level_set_bc(G.h, C.byref(b)))
        G.vcycle(sp)
        return p1, G.get(lv.F_PHI), da

    parts = run_strips(world, f, bc, ph, 0.0, -1.0, body, halo=24, max_box=64)
    assert all(p[2] == expect_da for p in parts), [p[2] for p in parts]
    O = oracle.OracleLevel(n, n, f["dx"], f["dy"], bc, ph, 0.0, -1.0, 64, 4)
    O.set_inputs(f); O.build_mg_coefficients(); O.vcycle(sp)
    phi1 = O.get(oracle.F_PHI)
    assert np.array_equal(np.vstack([p[0] for p in parts]), phi1)
    O2 = oracle.OracleLevel(n, n, f["dx"], f["dy"], bc2, ph, 0.0, -2.0, 64, 4)
    f2 = dict(f, phi=phi1)
    O2.set_inputs(f2); O2.build_mg_coefficients()
    for k, fid in (("bx", oracle.F_BX), ("by", oracle.F_BY)):
        O2.set(fid, O.get(fid))                           # the face coefficients the first cycle's UpdateOperator left (bcoeff_otf rebuilds them anyway)
    O2.vcycle(sp)
    assert np.array_equal(np.vstack([p[1] for p in parts]), O2.get(oracle.F_PHI))
    O.close(); O2.close()
