"""AMR levels that are unions of boxes on the GPU (suhmo_amd/csrc/suhmo_hier.hip) against the oracle's amrm.c on the same
inputs, BITWISE: fine-fine exchange, coarse-fine interpolation with coverage-aware stencils, the fine operator update,
the composite residual (reflux over box unions), AMR FAS V-cycles and the solve history on base + 3 levels with an L-shaped
union (re-entrant corner), a disjoint box and a box on the domain side; the same level cut into more boxes gives the same
bits; one box per level equals the nested-patch code (suhmo_amr_*)."""
import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
BC = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])     # 2lev_base/input.hydro:8-13,76
BC_NP = dict(type=[[0, 1], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])  # run_C_3lev/input.hydro: no periodic side
BC_V = dict(type=[[0, 1], [1, 0]], value=[[3.0, 0.01], [-0.02, 7.0]], periodic=[0, 0])
ONE = ([(16, 8, 47, 23)], [(44, 22, 75, 41)])
CUT = ([(16, 8, 31, 23), (32, 8, 47, 15), (32, 16, 47, 23)], [(44, 22, 59, 41), (60, 22, 75, 41)])
# periodic in y: two boxes that are neighbours THROUGH the wrap (rows 0-7 and 24-31 of a level of 32 rows), one that spans the period and is its own
# neighbour, and a finer box on the wrap inside it
WRAP = ([(16, 0, 31, 7), (16, 24, 31, 31), (40, 0, 55, 31)], [(84, 0, 99, 11), (84, 52, 99, 63)])
UNION = ([(16, 8, 31, 23), (32, 8, 47, 15), (0, 2, 11, 13)],
         [(36, 20, 59, 27), (36, 28, 51, 43), (4, 8, 15, 19)],
         [(80, 44, 103, 51), (12, 20, 23, 31)])
MASKPH = dict(sy.CFG3_PHYS, use_mask_gradients=1, cutOffbr=0.008, maxOffbr=0.012, cutOffB=1)


def pair(oracle, boxes, bc, ph=sy.CFG3_PHYS, nx0=64, ny0=16, options=None, **kw):
    from suhmo_amd import level
    fs = sy.amrm_fields(nx0, ny0, boxes, **kw)
    O = oracle.OracleAmrM(nx0, ny0, fs[0]["dx"], fs[0]["dy"], bc, ph, boxes, max_box=32, nthreads=2)
    O.set_inputs(fs)
    G = level.HipHier(nx0, ny0, fs[0]["dx"], fs[0]["dy"], bc, ph, boxes, max_box=32, options=options)
    G.set_inputs(fs)
    return O, G, fs


def eq(a, b, what):
    assert np.array_equal(a, b), (what, float(np.nanmax(np.abs(a - b))))


def uncovered(O, l, k):
    """cells of box k of level l that no box of level l + 1 covers (the device zeroes RES under finer levels, AMRNorm)"""
    lo0, lo1, hi0, hi1 = O.boxes[l - 1][k]
    m = np.ones((hi1 - lo1 + 1, hi0 - lo0 + 1), dtype=bool)
    if l < O.nlev - 1:
        for (f0, f1, g0, g1) in O.boxes[l]:
            a0, a1, c0, c1 = max(f0 // 2, lo0), min(g0 // 2, hi0), max(f1 // 2, lo1), min(g1 // 2, hi1)
            if a0 <= a1 and c0 <= c1:
                m[c0 - lo1:c1 - lo1 + 1, a0 - lo0:a1 - lo0 + 1] = False
    return m


def same_levels(O, G, oracle, fields, what, skip_covered=False):
    from suhmo_amd import level
    for l in range(1, O.nlev):
        for k in range(len(O.boxes[l - 1])):
            for fo, fg in fields:
                a, b = O.box_get(l, k, fo), G.level[l][k].get(fg)
                if skip_covered:
                    m = uncovered(O, l, k)
                    a, b = a[m], b[m]
                eq(a, b, (what, l, k, fo))
    eq(O.coarse.get(oracle.F_PHI), G.coarse.get(level.F_PHI), (what, "base phi"))


@pytest.mark.parametrize("bc,ph", [(BC_NP, sy.CFG3_PHYS), (BC_V, MASKPH)], ids=["cfg5-bc", "values-mask"])
def test_hier_pieces_bitwise(oracle, bc, ph):
    from suhmo_amd.level import F_PHI, F_RES, F_BX, F_BY
    O, G, fs = pair(oracle, UNION, bc, ph)
    for l in (1, 2, 3):
        O.cf_interp_phi(l); O.exchange(l, oracle.F_PHI)
        G.cf_interp(l); G.exchange(l, F_PHI)
        for k in range(len(UNION[l - 1])):
            a, b = O.box_get(l, k, oracle.F_PHI, ghosted=True), G.level[l][k].get(F_PHI, ghosted=True)
            eq(a[1:-1, :], b[1:-1, :], ("ghosts x", l, k)); eq(a[:, 1:-1], b[:, 1:-1], ("ghosts y", l, k))
    for l in (1, 2, 3):
        O.update_operator(l); G.update_operator(l)
    same_levels(O, G, oracle, ((oracle.F_BX, F_BX), (oracle.F_BY, F_BY)), "update_operator")
    O.coarse.update_operator(); G.coarse.update_operator()
    ro, rg = O.residual(), G.residual()
    assert ro == rg, (ro, rg)
    same_levels(O, G, oracle, ((oracle.F_RES, F_RES),), "residual", skip_covered=True)
    for l in (1, 2, 3):
        O.gsrb(l, 2); G.gsrb(l, 2)
    same_levels(O, G, oracle, ((oracle.F_PHI, F_PHI),), "gsrb")
    # sweep counts that are not a multiple of two launches of two: one launch of two passes (1 sweep), two launches + one of two passes (5 sweeps),
    # an odd number of launches (the head comes back from the second canvas by a copy)
    for n in (1, 5, 3):
        for l in (1, 2, 3):
            O.cf_interp_phi(l); G.cf_interp(l)
            O.gsrb(l, n); G.gsrb(l, n)
        same_levels(O, G, oracle, ((oracle.F_PHI, F_PHI),), "gsrb %d sweeps" % n)
    assert G.get_option("fused_relax_launches") > 0
    O.close(); G.close()


@pytest.mark.parametrize("name,boxes,bc,ph", [("union-4lev", UNION, BC_NP, sy.CFG3_PHYS), ("union-4lev-values-mask", UNION, BC_V, MASKPH),
                                              ("cut-periodic", CUT, BC, sy.CFG3_PHYS), ("wrap-periodic", WRAP, BC, sy.CFG3_PHYS),
                                              ("wrap-periodic-a-launch-per-colour-pass", WRAP, BC, sy.CFG3_PHYS),
                                              ("union-4lev-three-sweeps", UNION, BC_NP, sy.CFG3_PHYS), ("union-4lev-values-mask-two-sweeps", UNION, BC_V, MASKPH),
                                              ("union-4lev-six-sweeps", UNION, BC_NP, sy.CFG3_PHYS), ("union-4lev-exchange-per-pass", UNION, BC_NP, sy.CFG3_PHYS),
                                              ("union-4lev-whole-level-residuals", UNION, BC_NP, sy.CFG3_PHYS),
                                              ("union-4lev-a-launch-per-colour-pass", UNION, BC_V, MASKPH),
                                              ("union-4lev-a-launch-per-ghost-kind", UNION, BC_NP, sy.CFG3_PHYS),
                                              ("union-4lev-base-on-the-streaming-kernel", UNION, BC_NP, sy.CFG3_PHYS),
                                              ("union-4lev-values-mask-base-on-the-streaming-kernel", UNION, BC_V, MASKPH),
                                              ("union-4lev-base-on-the-streaming-kernel-own-residual-pass", UNION, BC_NP, sy.CFG3_PHYS)],
                         ids=lambda v: v if isinstance(v, str) else "")
def test_hier_vcycle_and_solve_bitwise(oracle, name, boxes, bc, ph, monkeypatch):
    from suhmo_amd.level import F_PHI, F_RES, F_BX
    # base-on-the-streaming-kernel: level 0 relaxes with the streaming kernel (what a 4096^2 base does), whose last launch of level 0's own
    # V-cycle leaves L(phi) and TRUE rhs - L(phi) behind for the solve loop's residual evaluation (-own-residual-pass: switched off)
    streams = "base-on-the-streaming-kernel" in name
    if streams:
        monkeypatch.setenv("SUHMO_FUSED_MIN_CELLS", "1")
        monkeypatch.setenv("SUHMO_GSRB_TILE", "0")
        monkeypatch.setenv("SUHMO_RESID_IN_RELAX", "0" if name.endswith("own-residual-pass") else "1")
    # exchange-per-pass: an exchange launch before every colour pass instead of the pushed side cells (creation option of the hierarchy)
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=6, imin=30)
    # -N-sweeps: num_smooth other than the reference's 4 (one launch of 8 passes per smoothing): 3 -> one launch of 6 passes, 2 -> one of 4 (the two-sweep
    # form of the kernel), 6 -> 8 + 4 passes; an odd number of launches leaves the head on its second canvas between pre- and post-smoothing
    for word, n in (("three", 3), ("two", 2), ("six", 6)):
        if name.endswith("-%s-sweeps" % word):
            sp["num_smooth"] = n
    # whole-level-residuals: every composite residual and coarse gradient over all of level 0 (default: the solve loop's evaluation is
    # reused by the next cycle except where level 1 was averaged down; the coarse gradient only where the interpolation reads it)
    # default: two sweeps per launch on the box levels (suhmo_gsrb.hip:k_gsrb_box_m) and AMRProlongS_2 of a box in one workgroup;
    # a-launch-per-colour-pass: the paths they replace (a launch per colour pass that pushes its side cells; gather, BC and prolongation as three launches;
    # merged_launches=0: a launch for either kind of ghost cell, a norm and a read-back per level, the closing ghost fill on its own)
    opts = {"exchange-per-pass": "push_ghosts=0,fused_relax=0", "whole-level-residuals": "incremental_residual=0", "a-launch-per-colour-pass": "fused_relax=0,fused_prolong=0,merged_launches=0",
            "a-launch-per-ghost-kind": "merged_launches=0,box_sweeps=2"}
    options = next((v for k, v in opts.items() if name.endswith(k)), None)
    O, G, fs = pair(oracle, boxes, bc, ph, options=options)
    O.vcycle(sp); G.vcycle(sp)
    same_levels(O, G, oracle, ((oracle.F_PHI, F_PHI), (oracle.F_BX, F_BX)), "vcycle")
    no, ho = O.solve(sp)
    ng, hg = G.solve(sp)
    assert no == ng and np.array_equal(ho, hg), (ho, hg)
    same_levels(O, G, oracle, ((oracle.F_PHI, F_PHI),), "solve")
    same_levels(O, G, oracle, ((oracle.F_RES, F_RES),), "solve residual", skip_covered=True)
    if streams:
        assert (G.coarse.get_option("residual_in_relax_launches") > 0) == (not name.endswith("own-residual-pass"))
    assert (G.get_option("fused_relax_launches") > 0) == (options is None or "fused_relax=0" not in options)
    O.close(); G.close()


def test_cutting_a_level_into_boxes_changes_no_bit_on_the_device(oracle):
    from suhmo_amd import level
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
    res = []
    for boxes in (ONE, CUT):
        fs = sy.amrm_fields(64, 16, boxes)
        G = level.HipHier(64, 16, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, boxes, max_box=32)
        G.set_inputs(fs)
        n, h = G.solve(sp)
        res.append((n, h, [G.level_array(l, level.F_PHI) for l in (1, 2)], G.coarse.get(level.F_PHI)))
        G.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(res[0][3], res[1][3])


def test_one_box_per_level_equals_the_nested_patch_code(oracle):
    from suhmo_amd import level
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=4, imin=30)
    fs = sy.amrm_fields(64, 16, ONE)
    H = level.HipHier(64, 16, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, ONE, max_box=32)
    H.set_inputs(fs)
    A = level.HipAmr(64, 16, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, ((8, 4, 23, 11), (22, 11, 37, 20)), max_box=32)
    A.levels[0].set_inputs(fs[0]); A.levels[0].build_mg_coefficients()
    for l in (1, 2):
        A.levels[l].set_inputs(fs[l][0])
    assert H.residual() == A.residual()
    nh, hh = H.solve(sp)
    na, ha = A.solve(sp)
    assert nh == na and np.array_equal(hh, ha)
    for l in (1, 2):
        assert np.array_equal(H.level[l][0].get(level.F_PHI), A.levels[l].get(level.F_PHI))
    assert np.array_equal(H.coarse.get(level.F_PHI), A.levels[0].get(level.F_PHI))
    H.close(); A.close()


def test_hier_refuses_layouts_the_reference_could_not_have():
    """suhmo_hier_create checks what BRMeshRefine guarantees: coarse-aligned boxes inside the refined domain, disjoint, properly nested
    (coarsen(box) grown by the stencils lies in the level below); and a level 0 cut into rank strips refuses to run without its
    all-gather.  Every refusal is an error code with a message, never a wrong result."""
    from suhmo_amd import capi, level as lv
    mk = lambda boxes, **kw: lv.HipHier(64, 32, 1.0, 1.0, sy.A3_BC, sy.A3_PHYS, boxes, max_box=16, **kw)
    bad = {"odd lower corner": [[(33, 16, 63, 47)]],
           "outside the refined domain": [[(32, 16, 129, 47)]],
           "overlapping boxes": [[(32, 16, 63, 47), (48, 32, 79, 63)]],
           "level 2 not inside level 1": [[(32, 16, 63, 47)], [(56, 24, 135, 71)]],
           "level 2 touching the edge of level 1 (no room for the coarse-fine stencil)": [[(32, 16, 63, 47)], [(64, 32, 95, 63)]]}
    for what, boxes in bad.items():
        with pytest.raises(capi.SuhmoError) as e:
            mk(boxes)
        assert "hier" in str(e.value) or "box" in str(e.value), (what, str(e.value))
    ok = mk([[(32, 16, 63, 47)], [(72, 40, 119, 87)]])
    ok.close()
    # a strip of level 0 without the all-gather attached: the first plan that reads level 0 says so
    H = mk([[(32, 16, 63, 47)]], j0=0, ny_global=64)
    H.level[0][0].set_inputs(sy.amrm_fields(64, 32, [[(32, 16, 63, 47)]])[0])
    with pytest.raises(capi.SuhmoError) as e:
        H.cf_interp(1)
    assert "all-gather" in str(e.value)
    H.close()
