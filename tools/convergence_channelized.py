#!/usr/bin/env python3
"""exec/0_convergence_channelized of the reference, single-level runs 1lev ... 7lev (32x8 ... 2048x512 on 64 m x 16 m, y-periodic,
bed slope 0.02, one moulin of 30 m3/s ramped up with suhmo.ramp, G = 0.05, diffusion + implicit gap-height solve, dt = 1 h): every
run goes its main.maxStep steps with the ramp, then the post-processing restart adds steps without it (input.hydro_pp; the plot
files compared are plot003200 for 1lev-5lev, plot001600 for 6lev, plot001100 for 7lev, CONV_ANA/scripts/launch_comparaison.py).
L2 self-convergence errors between successive resolutions as ChomboCompare computes them; the reference's table is
exec/0_convergence_channelized/CONV_ANA/results/convergence_data_singleLevel.dat.
usage: convergence_channelized.py [max_level 2..7]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suhmo_amd import synthetic as sy

LX, LY = 64.0, 16.0
BC = dict(type=[[0, 0], [1, 1]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])          # x-lo Dirichlet 0, x-hi Neumann 0, y periodic
PHYS = dict(sy.A3_PHYS, A=2.5e-25)
MODEL = dict(sy.A3_MODEL, G=0.05, ct=7.5e-8, diffFactor=1.0, use_impl_diff=1, distributed_input=1.0e-11, dt=3600.0, use_moulin_source=1)
MOULIN = (np.array([[16.015625, 8.015625]]), np.array([1.0]), np.array([30.0]))
# (main steps with the ramp, total steps of the compared plot file)
STEPS = {1: (3000, 3200), 2: (3000, 3200), 3: (3000, 3200), 4: (3000, 3200), 5: (3000, 3200), 6: (1500, 1600), 7: (1000, 1100)}


def basic_state(nx, ny, slope=0.02, ice_height=500.0, gap_init=0.01):
    """HydroIBC::initializeData (src/HydroIBC.cpp:189-277): zb = slope x, b = GapInit, Pi = rho_i g H, head = Pi / (2 rho_w g) + zb"""
    dx, dy = LX / nx, LY / ny
    i = np.arange(-1, nx + 1, dtype=np.float64)
    X = np.tile((i + 0.5) * dx, (ny + 2, 1))
    zb = slope * X
    Pi = np.full_like(X, sy.RHO_I * sy.GRAV * ice_height)
    head = (Pi * 0.5) * (1.0 / (sy.RHO_W * sy.GRAV)) + zb
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, head=head, B=np.full_like(X, gap_init), Pi=Pi, zb=zb, mask=np.ones_like(X))


def ramp(t):
    """src/AmrHydro.cpp:2448-2467 with the committed inputs (ramp_up 0.5, relax 0.1 months, floor 0.001 .. 1)"""
    month = 2635200.0
    return 0.001 + (1.0 - 0.001) * 0.5 * (np.tanh((t - 0.5 * month) / (0.1 * month)) + 1.0)


def run_oracle(level, phys):
    from oracle import pyoracle as po
    nx, ny = 32 << (level - 1), 8 << (level - 1)
    st = basic_state(nx, ny)
    m = dict(MODEL)
    M = po.OracleModel(nx, ny, st["dx"], st["dy"], BC, phys, m, max_box=min(64, ny), nthreads=min(8, os.cpu_count() or 1))
    M.set_state(st)
    M.field(po.OM_MR)[:] = m["G"] / m["L"]
    src, _ = po.moulin_source(nx, ny, st["dx"], st["dy"], MOULIN[0], MOULIN[1], MOULIN[2], 1.0)
    M.field(po.OM_MSRC)[1:-1, 1:-1] = src
    main, total = STEPS[level]
    import ctypes as C
    for k in range(total):
        M._mp.ramp = float(ramp(k * m["dt"])) if k < main else 1.0
        po.lib().or_model_set_ramp(M.h, C.c_double(M._mp.ramp))
        M.timestep(m["dt"])
    v = lambda fid: np.array(M.field(fid))[1:-1, 1:-1]
    out = dict(head=v(po.OM_H), B=v(po.OM_B), Pw=v(po.OM_PW), Re=v(po.OM_RE), msrc=v(po.OM_MSRC), dterm=np.zeros((ny, nx)), cd=v(po.OM_CD))
    M.close()
    return out, st["dx"]


def run(level, which="hip", phys=None):
    phys = PHYS if phys is None else phys
    if which == "oracle":
        return run_oracle(level, phys)
    from suhmo_amd import model, capi
    nx, ny = 32 << (level - 1), 8 << (level - 1)
    st = basic_state(nx, ny)
    m = dict(MODEL)
    M = model.HipModel(nx, ny, st["dx"], st["dy"], BC, phys, m, max_box=min(64, ny))
    M.set_state(st)
    M.level.set(model.lv.F_MR, np.full((ny, nx), m["G"] / m["L"]))       # thismeltRate = G / L
    M.moulin_source(*MOULIN, 1.0)
    main, total = STEPS[level]
    for k in range(total):
        M._mp.ramp = float(ramp(k * m["dt"])) if k < main else 1.0       # input.hydro_pp: suhmo.ramp = false
        M.timestep(m["dt"])
    out = dict(head=M.get("head"), B=M.get("B"), Pw=M.get("Pw"), Re=M.get("Re"), msrc=M.get("msrc"), dterm=M.level.get(model.lv.F_DTERM),
               cd=M.get("cd"))
    M.close()
    return out, st["dx"]


def moulin_table(which="hip", max_level=7):
    """the RHS_moulin column alone: the moulin source term needs no time step (time factor 1), so its self-convergence errors pin
    Calc_moulin_integral / Calc_moulin_source_term_distributed directly"""
    src, dxs = {}, {}
    for lev in range(1, max_level + 1):
        nx, ny = 32 << (lev - 1), 8 << (lev - 1)
        dx, dy = LX / nx, LY / ny
        if which == "hip":
            from suhmo_amd import model
            st = basic_state(nx, ny)
            M = model.HipModel(nx, ny, dx, dy, BC, PHYS, MODEL, max_box=min(64, ny))
            M.moulin_source(*MOULIN, 1.0)
            src[lev] = M.get("msrc")
            M.close()
        else:
            from oracle import pyoracle as po
            src[lev], _ = po.moulin_source(nx, ny, dx, dy, MOULIN[0], MOULIN[1], MOULIN[2], 1.0)
        dxs[lev] = dx
    res = {}
    for lev in range(1, max_level):
        f = src[lev + 1]
        e = src[lev] - 0.25 * (f[0::2, 0::2] + f[0::2, 1::2] + f[1::2, 0::2] + f[1::2, 1::2])
        res[32 << (lev - 1)] = float(np.sqrt(np.sum(e * e) * dxs[lev] * dxs[lev]))
    return res


def amr_grids():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "convergence_channelized_amr_grids.json")))


def amr_boxes(nx0, rects):
    """physical rectangles (x0, x1, y0, y1), one per AMR level -> boxes[l-1] = [(lo0, lo1, hi0, hi1)] in the index space of level l"""
    out = []
    for l, (x0, x1, y0, y1) in enumerate(rects, start=1):
        dx = LX / (nx0 << l)
        out.append([(int(round(x0 / dx)), int(round(y0 / dx)), int(round(x1 / dx)) - 1, int(round(y1 / dx)) - 1)])
    return out


def amr_moulin_error(nx0, rects, which="oracle"):
    """RHS_moulin entry of CONV_ANA/results/convergence_data_{2,3}Levels.dat: composite L2 difference (ChomboCompare: cells under a
    finer level do not count, the exact solution averaged to each level) between the moulin source term of the hierarchy
    (base nx0 x nx0/4 + len(rects) AMR levels) and the single level len(rects) + 1 refinements finer"""
    boxes = amr_boxes(nx0, rects)
    nlev = 1 + len(boxes)
    nxe = nx0 << (nlev)
    if which == "oracle":
        from oracle import pyoracle as po
        A = po.OracleAmrMModel(nx0, nx0 // 4, LX / nx0, LY / (nx0 // 4), BC, PHYS, MODEL, boxes, max_box=16, nthreads=2)
        A.moulin_source(*MOULIN, 1.0)
        get = lambda l, k: np.array(A.field(l, k, po.OM_MSRC))[1:-1, 1:-1]
        ex, _ = po.moulin_source(nxe, nxe // 4, LX / nxe, LY / (nxe // 4), MOULIN[0], MOULIN[1], MOULIN[2], 1.0)
    else:
        from suhmo_amd import model
        A = model.HipHierModel(nx0, nx0 // 4, LX / nx0, LY / (nx0 // 4), BC, PHYS, MODEL, boxes, max_box=16)
        A.moulin_source(*MOULIN, 1.0)
        get = lambda l, k: A.get(l, k, "msrc")
        E = model.HipModel(nxe, nxe // 4, LX / nxe, LY / (nxe // 4), BC, PHYS, MODEL, max_box=min(64, nxe // 4))
        E.moulin_source(*MOULIN, 1.0)
        ex = E.get("msrc")
        E.close()
    allb = [[(0, 0, nx0 - 1, nx0 // 4 - 1)]] + boxes
    tot = 0.0
    for l in range(nlev):
        nx = nx0 << l
        r = nxe // nx
        avg = ex.reshape(ex.shape[0] // r, r, ex.shape[1] // r, r).mean(axis=(1, 3))
        dx = LX / nx
        for k, (lo0, lo1, hi0, hi1) in enumerate(allb[l]):
            e = get(l, k) - avg[lo1:hi1 + 1, lo0:hi0 + 1]
            cov = np.zeros(e.shape, bool)
            if l + 1 < nlev:
                for (f0, f1, g0, g1) in allb[l + 1]:
                    a0, a1, c0, c1 = max(f0 // 2, lo0), min(g0 // 2, hi0), max(f1 // 2, lo1), min(g1 // 2, hi1)
                    if a0 <= a1 and c0 <= c1:
                        cov[c0 - lo1:c1 - lo1 + 1, a0 - lo0:a1 - lo0 + 1] = True
            tot += float(np.sum(e[~cov] ** 2)) * dx * dx
    A.close()
    return float(np.sqrt(tot))


def amr_run(nx0, rects, main_steps=3000, total_steps=7200):
    """{k}lev_base / {k}lev_base2levs on the device with the inferred (fixed) grids: the reference restarts the AMR run from the
    single-level checkpoint after the ramp (3000 steps) and regrids every 250 steps; here the hierarchy exists from the start --
    both end in the same steady channel.  Returns per level and box the fields ChomboCompare reads."""
    from suhmo_amd import model
    boxes = amr_boxes(nx0, rects)
    ny0 = nx0 // 4
    m = dict(MODEL)
    H = model.HipHierModel(nx0, ny0, LX / nx0, LY / ny0, BC, PHYS, m, boxes, max_box=min(64, ny0))
    allb = [[(0, 0, nx0 - 1, ny0 - 1)]] + boxes
    for l, bl in enumerate(allb):
        for k, (lo0, lo1, hi0, hi1) in enumerate(bl):
            nx = nx0 << l
            full = basic_state(nx, nx // 4)
            st = {key: (np.ascontiguousarray(v[lo1:hi1 + 3, lo0:hi0 + 3]) if isinstance(v, np.ndarray) else v) for key, v in full.items()}
            H.set_state(l, k, st)
            H.level[l][k].set(model.lv.F_MR, np.full((hi1 - lo1 + 1, hi0 - lo0 + 1), m["G"] / m["L"]))
    H.moulin_source(*MOULIN, 1.0)
    for k in range(total_steps):
        H._mp.ramp = float(ramp(k * m["dt"])) if k < main_steps else 1.0
        H.timestep(m["dt"])
    out = [[{nm: H.get(l, k, nm) for nm in ("head", "B", "Pw", "Re")} for k in range(len(bl))] for l, bl in enumerate(allb)]
    H.close()
    return out, allb


def amr_errors(nx0, rects, exact):
    """composite L2 errors (head, gapHeight, Pw, Re) of the AMR run against the single-level solution `exact` (dict of arrays)"""
    sol, allb = amr_run(nx0, rects)
    nxe = exact["head"].shape[1]
    res = {}
    for nm in ("head", "B", "Pw", "Re"):
        tot = 0.0
        for l, bl in enumerate(allb):
            nx = nx0 << l
            r = nxe // nx
            ex = exact[nm]
            avg = ex.reshape(ex.shape[0] // r, r, ex.shape[1] // r, r).mean(axis=(1, 3))
            dx = LX / nx
            for k, (lo0, lo1, hi0, hi1) in enumerate(bl):
                e = sol[l][k][nm] - avg[lo1:hi1 + 1, lo0:hi0 + 1]
                cov = np.zeros(e.shape, bool)
                if l + 1 < len(allb):
                    for (f0, f1, g0, g1) in allb[l + 1]:
                        a0, a1, c0, c1 = max(f0 // 2, lo0), min(g0 // 2, hi0), max(f1 // 2, lo1), min(g1 // 2, hi1)
                        if a0 <= a1 and c0 <= c1:
                            cov[c0 - lo1:c1 - lo1 + 1, a0 - lo0:a1 - lo0 + 1] = True
                tot += float(np.sum(e[~cov] ** 2)) * dx * dx
        res[nm] = float(np.sqrt(tot))
    return res


def table(max_level=7, log=None, which="hip", phys=None):
    sol, dxs = {}, {}
    for lev in range(1, max_level + 1):
        t0 = time.time()
        sol[lev], dxs[lev] = run(lev, which, phys)
        if log:
            log("%dlev (%d x %d): %d steps in %.1f s" % (lev, 32 << (lev - 1), 8 << (lev - 1), STEPS[lev][1], time.time() - t0))
    res = {}
    for lev in range(1, max_level):
        row = []
        for k in ("head", "B", "Pw", "Re", "msrc", "dterm", "cd"):
            f = sol[lev + 1][k]
            avg = 0.25 * (f[0::2, 0::2] + f[0::2, 1::2] + f[1::2, 0::2] + f[1::2, 1::2])
            e = sol[lev][k] - avg
            row.append(float(np.sqrt(np.nansum(e * e) * dxs[lev] * dxs[lev])))
        res[32 << (lev - 1)] = tuple(row)
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "amr":                       # convergence_channelized.py amr [2Levels|3Levels] [case ...]
        name = args[1] if len(args) > 1 else "2Levels"
        grids = amr_grids()[name]
        ref = {int(float(r[0])): r[1:] for r in np.loadtxt(os.path.join(ROOT, "tests", "golden", "convergence_channelized_%s_reference.dat" % name))}
        for case in (args[2:] or sorted(grids, key=int)):
            nx0 = int(case)
            nlev = 1 + len(grids[case])
            lev_exact = int(np.log2(nx0 // 32)) + 1 + nlev
            t0 = time.time()
            exact, _ = run(lev_exact, "hip")
            e = amr_errors(nx0, grids[case], exact)
            print("%s %d (exact = %dlev): head %.5g gapHeight %.5g Pw %.5g Re %.5g   [%.0f s]" % (name, nx0, lev_exact, e["head"], e["B"], e["Pw"], e["Re"], time.time() - t0))
            print("   ref  head %.5g gapHeight %.5g Pw %.5g Re %.5g   ratio" % tuple(ref[nx0][:4]), [round(a / b, 4) for a, b in zip((e["head"], e["B"], e["Pw"], e["Re"]), ref[nx0][:4])], flush=True)
        sys.exit(0)
    which, mg = "hip", 0
    if "--oracle" in args:
        which = "oracle"; args.remove("--oracle")
    if "--mask-gradients" in args:
        k = args.index("--mask-gradients"); mg = int(args[k + 1]); del args[k:k + 2]
    if "--head-melt-coef" in args:
        k = args.index("--head-melt-coef"); os.environ["SUHMO_ORACLE_HEAD_MELT_COEF"] = args[k + 1]; del args[k:k + 2]
    ml = int(args[0]) if args else 7
    ref = {int(float(r[0])): r[1:] for r in np.loadtxt(os.path.join(ROOT, "tests", "golden", "convergence_channelized_singleLevel_reference.dat"))}
    res = table(ml, log=lambda s_: print(s_, flush=True), which=which, phys=dict(PHYS, use_mask_gradients=mg))
    print("#case  head gapHeight Pw Re RHS_moulin DT CD")
    for nx, row in sorted(res.items()):
        print(nx, " ".join("%.5g" % v for v in row))
        print("   ref", " ".join("%.5g" % v for v in ref[nx]), "  ratio", [round(a / b, 4) for a, b in zip(row, ref[nx])], flush=True)
