"""Hierarchies of box unions with level 0 cut into rank strips (one process per GPU: a rank holds its rows of level 0 and every
box of the finer levels; level 1 reads level 0 through an all-gather of exactly the coarse cells its stencils touch and writes
only the rank's own rows).  Thread-"ranks" on one GPU drive suhmo_hier_timestep / suhmo_hier_solve through the hooks the
multi-GPU run uses; head, gap height, melt rate, fluxes and the iteration counts of every box on every rank must equal the
single-process hierarchy BIT FOR BIT."""
import threading

import numpy as np
import pytest

from suhmo_amd import synthetic as sy
from test_gpu_hier_timestep import B5ISH, MOULINS, UNION

pytestmark = pytest.mark.gpu
MB = 16
NAMES = ("head", "B", "mR", "rhs_h", "qwx")
NEAR_SIDES = ([(0, 0, 31, 31), (32, 0, 63, 15), (96, 40, 127, 63)], [(8, 4, 47, 27), (200, 96, 239, 119)])   # boxes on the domain sides, two strips apart


def strip_state(st, r0, ny):
    return {k: (v[r0:r0 + ny + 2] if isinstance(v, np.ndarray) else v) for k, v in st.items()}


def run_strips(world, nx0, ny0, boxes, sts, m, nsteps, mou, mb, halo_rows=4, options=None):
    from suhmo_amd import level as lv, model, multigpu
    n0 = ny0 // world
    tr = multigpu.ThreadTransport(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            G = model.HipHierModel(nx0, n0, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, boxes, max_box=mb,
                                   j0=rank * n0, ny_global=ny0, halo_rows=halo_rows, options=options)
            G.set_state(0, 0, strip_state(sts[0][0], rank * n0, n0))
            for l in range(1, len(sts)):
                for k, st in enumerate(sts[l]):
                    G.set_state(l, k, st)
            ex = multigpu.StripExchanger(G.level[0][0], tr, rank, world, False)
            ex.exchange_static()
            import os
            from suhmo_amd import capi
            da = capi.lib().suhmo_level_agglomerated_depth(G.level[0][0].h)
            assert (da > 0) == (int(os.environ.get("SUHMO_AGG_MIN_CELLS", "0")) > 0), da      # the variant really runs what its name says
            ag = multigpu.HierGather(G.hier, tr, rank)
            integ = G.moulin_source(**mou) if mou else None
            msrc = [[G.get(l, k, "msrc") for k in range(len(G.level[l]))] for l in range(len(sts))] if mou else None
            counts = [G.timestep(m["dt"]) for _ in range(nsteps)]
            res = [[{nm: G.get(l, k, nm) for nm in NAMES} for k in range(len(G.level[l]))] for l in range(len(sts))]
            part = [(G.hier.get_option("partitioned_level_%d" % l), G.hier.get_option("own_boxes_level_%d" % l)) for l in range(1, len(sts))]
            stats = [{k: G.hier.get_option("%s_level_%d" % (k, l)) for k in ("ghost_exchange_bytes", "ghost_exchange_bound_bytes", "held_boxes", "owned_cells", "canvas_bytes")}
                     for l in range(1, len(sts))]
            out[rank] = (counts, res, integ, msrc, ag.calls, G.hier.gathers() + G.hier.get_option("partition_gathers"), part, G.hier.get_option("partition_gathers"), stats)
            G.close()
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


def check_partition_stats(out, sts, world):
    """owner computes: a colour-pass ghost exchange moves at most the side cells of the rank's boxes (4 sides x 8 B, in fact one colour of the
    sides that face another rank's box); every cell of a level has one owner; a rank keeps canvases for its boxes and the neighbours' it reads"""
    for l in range(1, len(sts)):
        cells = sum(st["head"][1:-1, 1:-1].size for st in sts[l])
        assert sum(out[r][8][l - 1]["owned_cells"] for r in range(world)) == cells
        for r in range(world):
            q = out[r][8][l - 1]
            assert q["ghost_exchange_bytes"] <= q["ghost_exchange_bound_bytes"] // 2 + 8, (l, r, q)     # one colour of the sides
            assert q["held_boxes"] <= len(sts[l]) and (q["owned_cells"] == 0) == (out[r][6][l - 1][1] == 0)


CASES = [("union-2-ranks", 2, UNION, dict(), 2), ("union-4-ranks", 4, UNION, dict(), 2),
         ("union-2-ranks-moulins-implicit-gap", 2, UNION, B5ISH, 2), ("sides-4-ranks-moulins-implicit-gap", 4, NEAR_SIDES, B5ISH, 2)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("agg,part", [(0, 0), (600, 0), (0, 1), (600, 1)], ids=["", "coarse-depths-agglomerated", "boxes-partitioned", "agglomerated+partitioned"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hier_timestep_on_strips_bitwise(case, agg, part, monkeypatch):
    """part: the boxes of every finer level are dealt to the ranks (creation option partition_min_cells = 1: owner computes the colour
    passes, the operator and the residual of its boxes, the canvases travel by all-gather) -- the same bits as every rank relaxing all of them"""
    from suhmo_amd import model
    monkeypatch.setenv("SUHMO_AGG_MIN_CELLS", str(agg))      # > 0: the base strips' coarse multigrid depths run agglomerated (suhmo_agg.hip), also in the gap-height hierarchy
    name, world, boxes, mpo, nsteps = case
    nx0, ny0 = 64, 32
    m = dict(sy.A3_MODEL, **mpo)
    sts = sy.shmip_amrm_states(nx0, ny0, boxes, rough=0.5)
    mb = min(MB, ny0 // world)            # the strips must hold whole boxes of level 0: the same multigrid depths as the single process
    A = model.HipHierModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, boxes, max_box=mb)
    A.set_states(sts)
    mou = MOULINS if m.get("use_moulin_source") else None
    iref = A.moulin_source(**mou) if mou else None
    mref = [[A.get(l, k, "msrc") for k in range(len(A.level[l]))] for l in range(len(sts))] if mou else None
    ref_counts = [A.timestep(m["dt"]) for _ in range(nsteps)]
    ref = [[{nm: A.get(l, k, nm) for nm in NAMES} for k in range(len(A.level[l]))] for l in range(len(sts))]
    A.close()
    out = run_strips(world, nx0, ny0, boxes, sts, m, nsteps, mou, mb, options="partition_min_cells=1" if part else None)
    for l in range(1, len(sts)):
        flags = [out[r][6][l - 1][0] for r in range(world)]
        owned = [out[r][6][l - 1][1] for r in range(world)]
        assert flags == [part] * world, flags
        assert sum(owned) == (len(sts[l]) if part else world * len(sts[l])), owned     # every box has exactly one owner / is relaxed everywhere
    assert all((out[r][7] > 0) == bool(part) for r in range(world))
    for r in range(world):
        assert out[r][0] == ref_counts, (r, out[r][0], ref_counts)
        assert out[r][4] == out[r][5] > 0                        # every all-gather went through the hook
        if mou:
            assert np.array_equal(out[r][2], iref)               # integrals: the single-process bits on every rank
    for nm in NAMES:
        got = np.vstack([out[r][1][0][0][nm] for r in range(world)])
        assert np.array_equal(got, ref[0][0][nm], equal_nan=True), (name, 0, nm, float(np.nanmax(np.abs(got - ref[0][0][nm]))))
    if mou:
        got = np.vstack([out[r][3][0][0] for r in range(world)])
        assert np.array_equal(got, mref[0][0]), (name, "msrc level 0")
    for l in range(1, len(sts)):
        for k in range(len(sts[l])):
            holders = [r for r in range(world) if out[r][1][l][k]["head"] is not None]
            assert len(holders) == (1 if part else world), (name, l, k, holders)          # dealt to the ranks: exactly the owner answers for a box
            for r in holders:
                for nm in NAMES:
                    a, b = out[r][1][l][k][nm], ref[l][k][nm]
                    assert np.array_equal(a, b, equal_nan=True), (name, r, l, k, nm, float(np.nanmax(np.abs(a - b))))
                if mou:
                    assert np.array_equal(out[r][3][l][k], mref[l][k]), (name, r, l, k, "msrc")
    if part:
        check_partition_stats(out, sts, world)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed,world", [(1, 2), (2, 4), (3, 2), (4, 4)])
def test_random_box_layouts_partitioned_bitwise(seed, world):
    """box layouts grown around random points (boxes_around: nested, coarse-aligned, disjoint; some ranks may own no box of a level),
    every finer level dealt to its owners (partition_min_cells = 1): two time steps with moulins, diffusion and the implicit gap-height
    solve equal the single-process hierarchy on the device, bit for bit, on every rank"""
    from suhmo_amd import model
    rng = np.random.default_rng(4200 + seed)
    nx0, ny0, lx, ly = 128, 64, 1.0e5, 2.0e4
    pts = [(float(rng.uniform(0.1, 0.9) * lx), float(rng.uniform(0.1, 0.9) * ly)) for _ in range(int(rng.integers(2, 6)))]
    boxes = sy.boxes_around(pts, nx0, ny0, 3, lx, ly, radius_cells=(int(rng.integers(3, 7)), int(rng.integers(2, 5))), max_box=int(rng.choice([16, 32])))
    assert len(boxes) >= 1 and all(len(b) > 0 for b in boxes)
    m = dict(sy.A3_MODEL, **B5ISH)
    sts = sy.shmip_amrm_states(nx0, ny0, boxes, rough=0.5)
    mb = min(MB, ny0 // world)
    A = model.HipHierModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, boxes, max_box=mb)
    A.set_states(sts)
    mou = dict(positions=pts[:2], sigma=[900.0, 700.0], flux=[8.0, 5.0])
    iref = A.moulin_source(**mou)
    ref_counts = [A.timestep(m["dt"]) for _ in range(2)]
    ref = [[{nm: A.get(l, k, nm) for nm in NAMES} for k in range(len(A.level[l]))] for l in range(len(sts))]
    A.close()
    out = run_strips(world, nx0, ny0, boxes, sts, m, 2, mou, mb, options="partition_min_cells=1")
    for l in range(1, len(sts)):
        assert sum(out[r][6][l - 1][1] for r in range(world)) == len(sts[l]) and all(out[r][6][l - 1][0] == 1 for r in range(world))
    for r in range(world):
        assert out[r][0] == ref_counts, (r, out[r][0], ref_counts)
        assert np.array_equal(out[r][2], iref)
    for nm in NAMES:
        got = np.vstack([out[r][1][0][0][nm] for r in range(world)])
        assert np.array_equal(got, ref[0][0][nm], equal_nan=True), (seed, 0, nm)
    for l in range(1, len(sts)):
        for k in range(len(sts[l])):
            holders = [r for r in range(world) if out[r][1][l][k]["head"] is not None]
            assert len(holders) == 1, (seed, l, k, holders)
            for nm in NAMES:
                assert np.array_equal(out[holders[0]][1][l][k][nm], ref[l][k][nm], equal_nan=True), (seed, holders[0], l, k, nm)
    check_partition_stats(out, sts, world)


@pytest.mark.timeout(300)
def test_hier_solve_on_strips_bitwise():
    """suhmo_hier_solve alone (cfg4 physics, frozen gap height) on 2 strips: iteration count, residual history and head"""
    from suhmo_amd import level as lv, multigpu
    nx0, ny0, world = 64, 32, 2
    fs = sy.amrm_fields(nx0, ny0, UNION, seed=5)
    dx0, dy0 = fs[0]["dx"], fs[0]["dy"]
    sp = dict(num_smooth=4, num_bottom=16, max_iter=40, iter_min=2, imin=5, eps=1e-10, hang=1e-4, norm_thresh=1e-10, bcoeff_otf=1, max_depth=-1)
    H = lv.HipHier(nx0, ny0, dx0, dy0, sy.A3_BC, sy.A3_PHYS, UNION, max_box=MB)
    H.set_inputs(fs)
    n_ref, hist_ref = H.solve(sp)
    ref = [[H.level[l][k].get(lv.F_PHI) for k in range(len(H.level[l]))] for l in range(H.nlev)]
    H.close()
    n0 = ny0 // world
    tr = multigpu.ThreadTransport(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            G = lv.HipHier(nx0, n0, dx0, dy0, sy.A3_BC, sy.A3_PHYS, UNION, max_box=MB, j0=rank * n0, ny_global=ny0, halo_rows=4)
            f0 = fs[0]
            cut = {k: (v[rank * n0:rank * n0 + n0 + (2 if v.shape[0] == ny0 + 2 else (1 if v.shape[0] == ny0 + 1 else 0))] if isinstance(v, np.ndarray) else v)
                   for k, v in f0.items()}
            G.coarse.set_inputs(cut)
            for l in range(1, G.nlev):
                for k, f in enumerate(fs[l]):
                    G.level[l][k].set_inputs(f)
            ex = multigpu.StripExchanger(G.coarse, tr, rank, world, False)
            ex.exchange_static()
            G.coarse.build_mg_coefficients()
            ag = multigpu.HierGather(G, tr, rank)
            n, hist = G.solve(sp)
            out[rank] = (n, hist, [[G.level[l][k].get(lv.F_PHI) for k in range(len(G.level[l]))] for l in range(G.nlev)])
            G.close()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)
            tr.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    for r in range(world):
        assert out[r][0] == n_ref and np.array_equal(out[r][1], hist_ref), (r, out[r][0], n_ref)
        for l in range(1, len(ref)):
            for k in range(len(ref[l])):
                assert np.array_equal(out[r][2][l][k], ref[l][k]), (r, l, k)
    assert np.array_equal(np.vstack([out[r][2][0][0] for r in range(world)]), ref[0][0])


def test_native_allgather_single_rank():
    """The shadow path over the native transport: the creation option shadow=1 routes level 1's reads of an uncut level 0 through the
    pack kernel, ncclAllGather (one rank) on the kernels' stream and the unpack kernel; the time step must not change a bit."""
    from suhmo_amd import capi, model, multigpu
    m = dict(sy.A3_MODEL, **B5ISH)
    sts = sy.shmip_amrm_states(64, 32, UNION, rough=0.5)
    res = []
    for shadow in (0, 1):
        G = model.HipHierModel(64, 32, sts[0][0]["dx"], sts[0][0]["dy"], sy.A3_BC, sy.A3_PHYS, m, UNION, max_box=MB, options="shadow=%d" % shadow)
        G.set_states(sts)
        if shadow:
            multigpu.attach_rccl(G.level[0][0], 0, 1)
            capi.check(capi.lib().suhmo_hier_attach_rccl(G.hier.h))
        integ = G.moulin_source(**MOULINS)
        counts = [G.timestep(m["dt"]) for _ in range(2)]
        res.append((counts, integ, [[{nm: G.get(l, k, nm) for nm in NAMES} for k in range(len(G.level[l]))] for l in range(len(sts))], G.hier.gathers()))
        G.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert res[0][3] == 0 and res[1][3] > 50
    for l in range(len(sts)):
        for k in range(len(sts[l])):
            for nm in NAMES:
                assert np.array_equal(res[0][2][l][k][nm], res[1][2][l][k][nm], equal_nan=True), (l, k, nm)
