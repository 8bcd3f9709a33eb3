"""cfg5 of BASELINE.json (exec/AMR_multiMoulins: 63 moulins on 100 km x 100 km, dino bed of MountainSetupIBC without its
unseeded noise, diffusion + implicit gap-height solve, transient head + gap height) on base + 3 AMR levels whose levels are
unions of boxes around the moulins: the device hierarchy against oracle/amr_step_m.c, BITWISE over the first steps on a
64 x 64 base and on the reference's own 256 x 256 base (run_C_3lev/input.hydro:67), where also the moulins deliver their flux
over the composite grid and the covered cells hold the average of the finer level."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,max_box", [(64, 16), (256, 64)], ids=["base-64", "base-256-reference-grid"])
def test_cfg5_physics_on_four_levels_bitwise(oracle, nb, max_box):
    """base-256 = AmrHydro.num_cells and max_box_size of exec/AMR_multiMoulins/run_C_3lev/input.hydro:64-72 (base + 3 levels of 70 / 73 / 63 boxes):
    2 steps, every box of every level bit for bit"""
    from suhmo_amd import model, level as lv
    bc, ph, m, mo = sy.multimoulins_setup()
    nx0 = ny0 = nb
    boxes = sy.boxes_around(mo["positions"], nx0, ny0, 4, 1.0e5, 1.0e5)
    assert len(boxes) == 3 and all(len(bl) >= 3 for bl in boxes)
    sts = sy.mountain_amrm_states(nx0, ny0, boxes)
    O = oracle.OracleAmrMModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, m, boxes, max_box=max_box, nthreads=8)
    G = model.HipHierModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, m, boxes, max_box=max_box)
    O.set_states(sts); G.set_states(sts)
    io, ig = O.moulin_source(**mo), G.moulin_source(**mo)
    assert np.max(np.abs(io - ig)) <= 1e-12 * np.max(io)
    for l in range(O.nlev):
        for k in range(len(O.boxes[l])):
            a, b = np.array(O.field(l, k, oracle.OM_MSRC))[1:-1, 1:-1], G.get(l, k, "msrc")
            assert np.max(np.abs(a - b)) <= 1e-12 * max(np.max(np.abs(a)), 1e-300), (l, k)
            G.level[l][k].set(lv.F_MSRC, a)           # continue from identical source terms (two exp libraries)
    v = lambda a: np.array(a)[1:-1, 1:-1]
    for step in range(2):
        co, cg = O.timestep(m["dt"]), G.timestep(m["dt"])
        assert co == cg, (step, co, cg)
        for l in range(O.nlev):
            for k in range(len(O.boxes[l])):
                for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("mR", oracle.OM_MR), ("Re", oracle.OM_RE)):
                    a, b = v(O.field(l, k, fid)), G.get(l, k, nm)
                    assert np.array_equal(a, b, equal_nan=True), (step, l, k, nm, float(np.nanmax(np.abs(a - b))))
    O.close(); G.close()


def test_cfg5_on_the_reference_base_grid():
    from suhmo_amd import model
    bc, ph, m, mo = sy.multimoulins_setup()
    nx0 = ny0 = 256                                     # AmrHydro.num_cells of run_C_3lev/input.hydro
    boxes = sy.boxes_around(mo["positions"], nx0, ny0, 4, 1.0e5, 1.0e5)
    sts = sy.mountain_amrm_states(nx0, ny0, boxes)
    G = model.HipHierModel(nx0, ny0, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, m, boxes, max_box=64)
    G.set_states(sts)
    integ = G.moulin_source(**mo)
    assert np.all(integ > 0)
    total = 0.0
    for l in range(4):
        for k, b in enumerate(G.hier.boxes[l - 1] if l else [(0, 0, nx0 - 1, ny0 - 1)]):
            src = G.get(l, k, "msrc")
            cov = np.zeros(src.shape, dtype=bool)
            if l < 3:
                for (f0, f1, g0, g1) in G.hier.boxes[l]:
                    a0, a1, c0, c1 = max(f0 // 2, b[0]), min(g0 // 2, b[2]), max(f1 // 2, b[1]), min(g1 // 2, b[3])
                    if a0 <= a1 and c0 <= c1:
                        cov[c0 - b[1]:c1 - b[1] + 1, a0 - b[0]:a1 - b[0] + 1] = True
            total += src[~cov].sum() * sts[l][0]["dx"] * sts[l][0]["dy"]
    assert abs(total - mo["flux"].sum()) < 1e-9 * mo["flux"].sum()
    for step in range(2):
        pi, nv = G.timestep(m["dt"])
        assert 1 <= pi <= 30 and nv >= 2
    for l in range(4):
        for k in range(len(G.level[l])):
            for nm in ("head", "B", "mR"):
                assert np.all(np.isfinite(G.get(l, k, nm)))
    G.close()


@pytest.mark.timeout(600)
def test_cfg5_at_the_north_star_size_properties():
    """BASELINE north_star: 4096^2 base + 3 AMR levels (boxes around the 63 moulins of exec/AMR_multiMoulins), transient head + gap
    height.  No oracle finishes this size in test time; what must hold at any size: the moulin source term delivers the moulins'
    flux over the composite grid; after a step the head of level l under level l+1 IS the average of the finer head (CoarseAverage,
    src/AmrHydro.cpp:3138-3141), bit for bit; every field and the composite residual stay finite, the Picard loop converges."""
    from suhmo_amd import level as lv, model
    bc, ph, m, mo = sy.multimoulins_setup()
    nb = 4096
    boxes = sy.boxes_around(mo["positions"], nb, nb, 4, 1.0e5, 1.0e5)
    sts = sy.mountain_amrm_states(nb, nb, boxes)
    G = model.HipHierModel(nb, nb, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, m, boxes, max_box=64)
    G.set_states(sts)
    G.moulin_source(**mo)
    total = 0.0
    for l in range(4):
        for k, b in enumerate(G.hier.boxes[l - 1] if l else [(0, 0, nb - 1, nb - 1)]):
            src = G.get(l, k, "msrc")
            cov = np.zeros(src.shape, dtype=bool)
            if l < 3:
                for (f0, f1, g0, g1) in G.hier.boxes[l]:
                    a0, a1, c0, c1 = max(f0 // 2, b[0]), min(g0 // 2, b[2]), max(f1 // 2, b[1]), min(g1 // 2, b[3])
                    if a0 <= a1 and c0 <= c1:
                        cov[c0 - b[1]:c1 - b[1] + 1, a0 - b[0]:a1 - b[0] + 1] = True
            total += src[~cov].sum() * sts[l][0]["dx"] * sts[l][0]["dy"]
    assert abs(total - mo["flux"].sum()) < 1e-9 * mo["flux"].sum()
    pi, nv = G.timestep(m["dt"])
    assert 1 <= pi <= 30 and nv >= 2
    assert np.isfinite(G.hier.residual())
    # CoarseAverage: level 0 under the boxes of level 1
    h0 = G.get(0, 0, "head")
    for k, (f0, f1, g0, g1) in enumerate(G.hier.boxes[0][:8]):
        hf = G.get(1, k, "head")
        s = 0.0 + hf[0::2, 0::2]
        s = s + hf[0::2, 1::2]; s = s + hf[1::2, 0::2]; s = s + hf[1::2, 1::2]      # FORT_AVERAGE's visiting order: (0,0), (1,0), (0,1), (1,1)
        assert np.array_equal(h0[f1 // 2:g1 // 2 + 1, f0 // 2:g0 // 2 + 1], s * 0.25), k
    for l in range(4):
        for k in range(min(len(G.level[l]), 6)):
            for nm in ("head", "B", "mR"):
                assert np.all(np.isfinite(G.get(l, k, nm)))
    G.close()


def _cfg5_solver_inputs(nb, boxes, mo, m):
    """cfg5's state (mountain bed, ice, gap height, initial head) as inputs of the head solve on every box: the right-hand side is the
    distributed input plus the moulins' Gaussians sampled at the cell centres (numpy, the same array for checker and device)"""
    sts = sy.mountain_amrm_states(nb, nb, boxes)
    pos, sg, fl = mo["positions"], mo["sigma"], mo["flux"]
    out = []
    for l, bl in enumerate([[(0, 0, nb - 1, nb - 1)]] + [list(b) for b in boxes]):
        lev = []
        for k, (lo0, lo1, hi0, hi1) in enumerate(bl):
            st = sts[l][k]
            nx, ny, dx, dy = hi0 - lo0 + 1, hi1 - lo1 + 1, st["dx"], st["dy"]
            x = (np.arange(lo0, hi0 + 1) + 0.5) * dx
            y = (np.arange(lo1, hi1 + 1) + 0.5) * dy
            rhs = np.full((ny, nx), float(m["distributed_input"]))
            for (px, py), s, q in zip(pos, sg, fl):
                if px + 6 * s < x[0] or px - 6 * s > x[-1] or py + 6 * s < y[0] or py - 6 * s > y[-1]:
                    continue
                rhs = rhs + (q / (2.0 * np.pi * s * s)) * np.exp(-((x[None, :] - px) ** 2 + (y[:, None] - py) ** 2) / (2.0 * s * s))
            lev.append(dict(nx=nx, ny=ny, dx=dx, dy=dy, box=(lo0, lo1, hi0, hi1), phi=np.ascontiguousarray(st["head"][1:-1, 1:-1]), rhs=rhs,
                            aCoef=np.zeros((ny, nx)), B=st["B"], Pi=st["Pi"], zb=st["zb"], mask=st["mask"]))
        out.append(lev if l else lev[0])
    return out


@pytest.mark.timeout(1500)
def test_cfg5_north_star_size_amr_vcycles_bitwise(oracle):
    """BASELINE north_star's own size: 4096^2 base + 3 AMR levels of boxes around the 63 moulins (exec/AMR_multiMoulins/run_C_3lev/input.hydro,
    src/AmrHydro.cpp:3119-3141: SolveForHead_nl on the hierarchy).  The head solve's loop -- composite residual, two AMR FAS V-cycles, the
    residual after each -- against oracle/amrm.c on the host's cores: residual history, the head of EVERY box of EVERY level and of the
    16.8 M cells of level 0, bit for bit.  This is the size where level 0 relaxes on the streaming kernel whose last launch leaves
    L(phi) and the TRUE rhs - L(phi) behind (resout_rhs), where the cycle's composite residual of level 0 re-evaluates only the rectangles
    the average from level 1 changed, and where level 0's gradient is evaluated only at the cells level 1's stencils read: the counters
    say that all three ran."""
    from suhmo_amd import level as lv
    bc, ph, m, mo = sy.multimoulins_setup()
    nb = 4096
    boxes = sy.boxes_around(mo["positions"], nb, nb, 4, 1.0e5, 1.0e5)
    fs = _cfg5_solver_inputs(nb, boxes, mo, m)
    sp = dict(num_smooth=4, num_bottom=16, max_iter=2, iter_min=2, imin=5, eps=1e-10, hang=1e-4, norm_thresh=1e-10, bcoeff_otf=1, max_depth=-1)
    G = lv.HipHier(nb, nb, fs[0]["dx"], fs[0]["dy"], bc, ph, boxes, max_box=64)
    G.set_inputs(fs)
    base = G.coarse
    c0 = (G.get_option("incremental_residual_passes"), G.get_option("residuals_left_by_relax"), G.get_option("sparse_gradient_passes"),
          base.get_option("residual_in_relax_launches"))
    ng, hg = G.solve(sp)
    c1 = (G.get_option("incremental_residual_passes"), G.get_option("residuals_left_by_relax"), G.get_option("sparse_gradient_passes"),
          base.get_option("residual_in_relax_launches"))
    assert ng == 2
    assert c1[0] - c0[0] >= 1, "the cycle's composite residual of level 0 was not the incremental one"
    assert c1[1] - c0[1] == 2, "the solve loop's residual of level 0 was not the one the cycle's last launch left behind"
    assert c1[2] - c0[2] == 2, "level 0's gradient was evaluated over the whole level"
    assert c1[3] - c0[3] >= 2, "the streaming launch did not write the residual (resout_rhs)"
    got = [[G.level[l][k].get(lv.F_PHI) for k in range(len(G.level[l]))] for l in range(G.nlev)]
    G.close()
    import os
    O = oracle.OracleAmrM(nb, nb, fs[0]["dx"], fs[0]["dy"], bc, ph, boxes, max_box=64, nthreads=min(16, len(os.sched_getaffinity(0))))
    O.set_inputs(fs)
    no, ho = O.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (no, ng, ho, hg)
    assert np.array_equal(O.coarse.get(oracle.F_PHI), got[0][0]), "level 0"
    for l in range(1, O.nlev):
        for k in range(len(boxes[l - 1])):
            a = O.box_get(l, k, oracle.F_PHI)
            assert np.array_equal(a, got[l][k]), (l, k, float(np.max(np.abs(a - got[l][k]))))
    O.close()
