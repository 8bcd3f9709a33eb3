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
    chunks relax, the two end chunks after it (what a 4096^2 strip does); "no-overlap": the same launches, exchange first"""
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
        return p1, G.get(lv.F_PHI), n, hist, G.ndepth, G.get_option("overlapped_launches")

    parts = run_strips(world, f, bc, ph, 0.0, -1.0, body, halo=halo, max_box=64)
    if fused == "overlap":
        assert all(p[5] > 0 for p in parts), [p[5] for p in parts]
    elif fused == "no-overlap":
        assert all(p[5] == 0 for p in parts)
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
