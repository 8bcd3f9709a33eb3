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
