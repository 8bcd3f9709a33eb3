#!/usr/bin/env python3
"""Runs a SHMIP suite-F case (exec/F_SHMIP/F<k>/input.hydro: the valley glacier of suite E, gamma = 0.05, A = 2.5e-25, under a
seasonal temperature cycle shifted by deltaT = -6 ... +6 K: COMPUTE_TIMEVARYINGRECHARGE every step, src/AmrHydro.cpp:2854-2863) and
compares the daily time series with the reference's committed result (tests/golden/shmip_F<k>_postproc_reference.dat, a DATA
fixture copied from exec/F_SHMIP/F<k>/results/postproc.dat: T_hrs T_days avgN N_LB N_MB N_HB rech dis).
The reference restarts F<k> from the checkpoint of exec/F_SHMIP/SS_initial_run (8000 steps of 1 h under the background input
7.93e-11, no time variation) with m_time = 0 (amr.restart_time), then takes 21960 steps of 2 h (5 years); the checkpoint is not
shipped, so the spin-up is run here too.
usage: run_shmip_f.py oracle|hip F<k> [years] [out.json] [--head-melt-coef X] [--mask-gradients 0|1] [--freeze-icefree]
                      [--mask-rhs-b 0|1] [--cutoffb 0|1] [--zs surface|thickness]      (run-state knobs of the oracle)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suhmo_amd import synthetic as sy

DELTA_T = dict(F1=-6.0, F2=-3.0, F3=0.0, F4=3.0, F5=6.0)          # exec/F_SHMIP/F<k>/input.hydro:44
BACKGROUND = 7.93e-11
F_MODEL = dict(sy.E_MODEL, distributed_input=0.0, use_mask_rhs_b=0, use_moulin_source=1, ramp=1.0, eps_picard=1.0e-4)
# use_moulin_source with ramp 1 and distributed_input 0: RHS_h takes the source field as it is (n_moulins < 0, :3067)


def opt(name, conv=str, flag=False):
    if name in sys.argv:
        k = sys.argv.index(name)
        if flag:
            del sys.argv[k]
            return True
        v = conv(sys.argv[k + 1])
        del sys.argv[k:k + 2]
        return v
    return False if flag else None


def daily_row(t_end, dx, dy, qwx, src, mR, Pw, Pi, mask, rho_w=1000.0):
    """the two "Time(h - d)" lines of the temporal post-processing (src/AmrHydro.cpp:3760-3810, 4040-4053) at the end of a day"""
    ny, nx = mR.shape
    ice = mask > 0.0
    ok = ice & (Pi > 0.0)
    N = Pi - Pw
    xloc = (np.arange(nx) + 0.5) * dx
    band = lambda lo, hi: N[:, (xloc > lo) & (xloc < hi)][ok[:, (xloc > lo) & (xloc < hi)]]
    ext = np.where(ice, src * dy * dx, 0.0).sum(axis=0)
    mr = np.where(ice, (mR / rho_w) * dy * dx, 0.0).sum(axis=0)
    rech = ext[1:].sum() + mr[1:].sum()                                   # out_recharge_*[1]: cumulative from the upper end
    dis = (qwx[:, 1] * dy).sum()                                          # - out_water_flux_x_tot[1]
    return [t_end / 3600.0, t_end / 86400.0, N[ok].mean(), band(600.0, 900.0).mean(), band(3000.0, 3300.0).mean(), band(5100.0, 5400.0).mean(),
            rech, -dis, ext[1:].sum(), mr[1:].sum()]


def main():
    coef = opt("--head-melt-coef")
    mask_grad = opt("--mask-gradients", int)
    mask_rhs_b = opt("--mask-rhs-b", int)
    cutoffb = opt("--cutoffb", int)
    spin = opt("--spinup-steps", int)
    zs_kind = opt("--zs") or "thickness"
    freeze = opt("--freeze-icefree", flag=True)
    if freeze:
        os.environ["SUHMO_ORACLE_GAP_FREEZE_ICEFREE"] = "1"
    if coef is not None:
        os.environ["SUHMO_ORACLE_HEAD_MELT_COEF"] = coef
    which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
    case = sys.argv[2] if len(sys.argv) > 2 else "F1"
    years = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    out_json = sys.argv[4] if len(sys.argv) > 4 else None
    m = dict(F_MODEL)
    if mask_rhs_b is not None:
        m["use_mask_rhs_b"] = mask_rhs_b
    if which == "hip":                                                     # the device path: model options (suhmo_model_params_t), no environment
        if coef is not None:
            assert float(coef) == 0.0, "the device path has the model option head_melt_off only"
            m["head_melt_off"] = 1
        m["freeze_icefree_gap"] = int(freeze)
    phys = dict(sy.A3_PHYS, A=2.5e-25, use_mask_gradients=mask_grad or 0, cutOffB=cutoffb or 0)
    nx, ny = m["nx"], m["ny"]
    st = sy.valley_initial_state(nx, ny, 0.05, m["lx"], m["ly"])
    X = (np.arange(-1, nx + 1) + 0.5)[None, :] * st["dx"] + np.zeros((ny + 2, 1))
    surf = 100.0 * np.power(X + 200.0, 0.25) + X / 60.0 - np.power(2.0e10, 0.25) + 1.0
    # thisiceHeight: the committed ValleyIBC ends with the ice THICKNESS max(surface - bed, 0) (src/ValleyIBC.cpp:299 "correct for ice
    # height"); the reference's F tables were written with the SURFACE elevation still in that field (--zs surface)
    zs = surf if zs_kind == "surface" else np.maximum(surf - st["zb"], 0.0)
    mask = st["mask"]
    nspin = 8000 if spin is None else spin
    nsteps = int(round(years * 366 * 12))                                # 21960 steps of 2 h in 5 "years" of 366 days (input.hydro:5)
    t0 = time.time()
    if which == "oracle":
        from oracle import pyoracle as po
        M = po.OracleModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64, nthreads=int(os.environ.get("OMP_NUM_THREADS", min(8, os.cpu_count() or 1))))
        M.set_state(st)
        M.field(po.OM_MR)[:] = m["G"] / m["L"]
        set_src = lambda a: M.field(po.OM_MSRC).__setitem__(slice(None), a)
        recharge = lambda T_K: po.time_varying_recharge(zs, T_K, BACKGROUND)
        g = lambda fid: np.array(M.field(fid))
        v = lambda a: a[1:-1, 1:-1]
        fields = lambda: (g(po.OM_QWX), v(g(po.OM_SRC)), v(g(po.OM_MR)), v(g(po.OM_PW)), v(g(po.OM_PI)), v(g(po.OM_MASK)))
    else:
        from suhmo_amd import model
        M = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64)
        M.set_state(st)
        lv = model.lv
        M.level.set(lv.F_MR, np.full((ny, nx), m["G"] / m["L"]))          # thismeltRate = G / L (ValleyIBC::initializeData)
        set_src = lambda a: M.level.set(lv.F_MSRC, a, ghosted=True)
        recharge = None
        fields = None                       # the daily row comes from the device (suhmo_level_postproc_temporal)
    tot_p = tot_v = 0
    # spin-up: the source term of a run without time variation is the background where there is ice (:2866-2876)
    set_src(np.where(mask > 0.0, BACKGROUND, 0.0))
    for k in range(nspin):
        p, v_ = M.timestep(3600.0); tot_p += p; tot_v += v_
        if (k + 1) % 2000 == 0:
            print("spin-up step %d  picard %d  vcycles %d  %.0f s" % (k + 1, tot_p, tot_v, time.time() - t0), flush=True)
    rows = []
    tm, dt = 0.0, 7200.0
    for k in range(nsteps):
        T_K = -16.0 * np.cos(2.0 * np.pi * tm / (365.0 * 24 * 60 * 60.0)) - 5.0 + DELTA_T[case]     # :2855, m_restart_time = 0
        if which == "oracle":
            set_src(recharge(T_K))
        else:
            M.time_varying_recharge(zs, T_K, BACKGROUND)
        p, v_ = M.timestep(dt); tot_p += p; tot_v += v_
        if int(tm + dt) % 86400 == 0:
            if which == "oracle":
                rows.append(daily_row(tm + dt, st["dx"], st["dy"], *fields()))
            else:
                sums = M.postproc_partial()
                rows.append([(tm + dt) / 3600.0, (tm + dt) / 86400.0] + list(M.postproc_temporal()) + [sums[4, 1:].sum(), sums[5, 1:].sum()])
        tm += dt
        if (k + 1) % 2000 == 0:
            print("step %d  picard %d  vcycles %d  %.0f s" % (k + 1, tot_p, tot_v, time.time() - t0), flush=True)
    table = np.array(rows)
    ref = np.loadtxt(os.path.join(ROOT, "tests", "golden", "shmip_%s_postproc_reference.dat" % case))[: len(rows)]
    names = ["T_hrs", "T_days", "avgN", "N_LB", "N_MB", "N_HB", "rech", "dis"]
    cmp_ = {}
    for c in range(2, 8):
        scale = np.max(np.abs(ref[:, c]))
        d = np.abs(table[:, c] - ref[:, c])
        cmp_[names[c]] = {"max_rel_to_scale": float(d.max() / scale), "at_day": float(table[int(d.argmax()), 1]), "first_row_rel": float(d[0] / scale)}
    res = {"which": which, "case": case, "years": years, "spinup_steps": nspin, "head_melt_coef": coef, "mask_gradients": mask_grad, "mask_rhs_b": mask_rhs_b,
           "cutoffb": cutoffb, "zs": zs_kind, "freeze_icefree": bool(os.environ.get("SUHMO_ORACLE_GAP_FREEZE_ICEFREE")), "picard_total": tot_p, "vcycles_total": tot_v,
           "seconds": time.time() - t0, "rows": len(rows), "vs_reference": cmp_}
    print(json.dumps(res, indent=1))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
        np.savetxt(out_json.replace(".json", "_table.dat"), table, fmt="%.10g")


if __name__ == "__main__":
    main()
