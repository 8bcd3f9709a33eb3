#!/usr/bin/env python3
"""Runs a SHMIP suite-A case (exec/A_SHMIP/A<k>/input.hydro: 320 x 64 cells, dt = 1 h, 10000 steps + the 2
post-processing steps of input.hydro_pp; the six cases differ in suhmo.distributed_input only) through the time
loop and compares the cross-section table with the reference's committed result
tests/golden/shmip_A<k>_postproc_reference.dat (a DATA fixture copied from exec/A_SHMIP/A<k>/results/postproc.dat).
usage: run_shmip_a.py oracle|hip A<k> [nsteps] [out.json] [--head-melt-coef X] [--mask-gradients 0|1] [--freeze-icefree]
--mask-gradients 0|1 overrides solver.use_mask_for_gradients of the case.
--head-melt-coef X scales the melt term of RHS_h (src/AmrHydro.cpp:3046); X = 0 reproduces the code state the reference's
committed results were evidently produced with ("end-to-end pin").  The oracle takes any X (environment knob); the
device path knows the model option head_melt_off (suhmo_model_params_t), i.e. X = 0 or nothing.
--freeze-icefree: cells without ice keep their gap height through the implicit gap-height solve (suite E's third run-state
setting; model option freeze_icefree_gap on both sides).
The committed tests/golden/shmip_A<k>_oracle_run{.json,_table.dat} were written by
    python tools/run_shmip_a.py oracle A<k> 10002 tests/golden/shmip_A<k>_oracle_run.json
and tests/golden/shmip_A<k>_oracle_nomelt{.json,_table.dat} by the same command with --head-melt-coef 0."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suhmo_amd import synthetic as sy


def compare(table, ref):
    out = {}
    names = ["x_km", "Ylength", "discharge", "dischargeEFF", "dischargeINEFF", "recharge_ext", "recharge_melt", "N_MPa"]
    for c in range(8):
        a, b = table[:, c], ref[:, c]
        sel = slice(1, None) if c in (2, 3, 4) else slice(None)   # row 0 of the discharge columns: see DESIGN.md
        scale = np.max(np.abs(b[sel]))
        out[names[c]] = {"max_abs_diff": float(np.max(np.abs(a[sel] - b[sel]))), "scale": float(scale),
                         "max_rel_to_scale": float(np.max(np.abs(a[sel] - b[sel])) / scale) if scale > 0 else 0.0}
    return out


def main():
    coef = None
    if "--head-melt-coef" in sys.argv:
        k = sys.argv.index("--head-melt-coef")
        coef = sys.argv[k + 1]
        del sys.argv[k:k + 2]
        os.environ["SUHMO_ORACLE_HEAD_MELT_COEF"] = coef
    mask_grad = None
    if "--mask-gradients" in sys.argv:
        k = sys.argv.index("--mask-gradients")
        mask_grad = int(sys.argv[k + 1])
        del sys.argv[k:k + 2]
    pp_cutoffb = None
    if "--pp-cutoffb" in sys.argv:                      # solver.cut_solve_outside_domain of the 2 post-processing steps (suite E: input.hydro_pp drops the key -> 0)
        k = sys.argv.index("--pp-cutoffb")
        pp_cutoffb = int(sys.argv[k + 1])
        del sys.argv[k:k + 2]
    freeze = False
    if "--freeze-icefree" in sys.argv:
        sys.argv.remove("--freeze-icefree")
        freeze = True
    which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
    case = sys.argv[2] if len(sys.argv) > 2 else "A3"
    binp = None
    if case.startswith("B"):
        binp = json.load(open(os.path.join(ROOT, "tests", "golden", "shmip_B_inputs.json")))[case]
        m = sy.shmip_b_model(case, binp)
    elif case.startswith("E"):
        m = sy.shmip_e_model(case)
    else:
        m = sy.shmip_a_model(case)
    if freeze:
        m = dict(m, freeze_icefree_gap=1)
    if which == "hip" and coef is not None:
        assert float(coef) == 0.0, "the device path has the model option head_melt_off only"
        m = dict(m, head_melt_off=1)
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else m["max_step"] + 2
    out_json = sys.argv[4] if len(sys.argv) > 4 else None
    phys = sy.A3_PHYS
    if case.startswith("E"):
        st = sy.valley_initial_state(m["nx"], m["ny"], sy.E_GAMMA[case], m["lx"], m["ly"])
        phys = dict(sy.E_PHYS, cutOffB=sy.E_CUTOFFB[case])
    else:
        st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
    if mask_grad is not None:
        phys = dict(phys, use_mask_gradients=mask_grad)
    t0 = time.time()
    if which == "oracle":
        from oracle import pyoracle as po
        M = po.OracleModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64,
                           nthreads=min(8, os.cpu_count() or 1))
        M.set_state(st)
        M.field(po.OM_MR)[:] = m["G"] / m["L"]            # thismeltRate = G / L (SqrtIBC / ValleyIBC::initializeData)
        if binp:
            src, _ = po.moulin_source(m["nx"], m["ny"], st["dx"], st["dy"], np.array(binp["positions"]).reshape(-1, 2),
                                      binp["sigma"], binp["flux"], 1.0)
            M.field(po.OM_MSRC)[1:-1, 1:-1] = src
        tot_p = tot_v = 0
        for k in range(nsteps):
            if pp_cutoffb is not None and k == nsteps - 2:
                po.lib().or_model_set_cutoffb(M.h, pp_cutoffb)
            p, v = M.timestep(m["dt"]); tot_p += p; tot_v += v
            if (k + 1) % 1000 == 0:
                print("step %d  picard %d  vcycles %d  %.0f s" % (k + 1, tot_p, tot_v, time.time() - t0), flush=True)
        g = lambda fid: np.array(M.field(fid))
        v = lambda a: a[1:-1, 1:-1]
        table = sy.shmip_postproc_table(st["dx"], st["dy"], g(po.OM_QWX), g(po.OM_CD), v(g(po.OM_SRC)), v(g(po.OM_MR)),
                                        v(g(po.OM_PW)), v(g(po.OM_PI)), v(g(po.OM_MASK)))
        head, gap = v(g(po.OM_H)), v(g(po.OM_B))
    else:
        from suhmo_amd import model
        from suhmo_amd import level as lv
        M = model.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64)
        M.set_state(st)
        M.level.set(lv.F_MR, np.full((m["ny"], m["nx"]), m["G"] / m["L"]))      # thismeltRate = G / L (SqrtIBC / ValleyIBC::initializeData)
        src = None
        if binp:
            from oracle import pyoracle as po            # the checker's exp(): tests/test_gpu_moulin.py bounds the device's against it
            src, _ = po.moulin_source(m["nx"], m["ny"], st["dx"], st["dy"], np.array(binp["positions"]).reshape(-1, 2),
                                      binp["sigma"], binp["flux"], 1.0)
            M.level.set(lv.F_MSRC, src)
        tot_p = tot_v = 0
        for k in range(nsteps):
            p, v = M.timestep(m["dt"]); tot_p += p; tot_v += v
            if (k + 1) % 1000 == 0:
                print("step %d  picard %d  vcycles %d  %.0f s" % (k + 1, tot_p, tot_v, time.time() - t0), flush=True)
        if binp:
            mask = M.get("mask")
            table = sy.shmip_postproc_table(st["dx"], st["dy"], M.get("qwx"), M.get("cd", ghosted=True),
                                            np.where(mask > 0.0, src * m["ramp"] + m["distributed_input"], 0.0),
                                            M.get("mR"), M.get("Pw"), M.get("Pi"), mask, m["rho_w"])
        else:
            table = M.postproc_table()
        head, gap = M.get("head"), M.get("B")
    ref = np.loadtxt(os.path.join(ROOT, "tests", "golden", "shmip_%s_postproc_reference.dat" % case))
    cmp_ = compare(table, ref)
    res = {"which": which, "case": case, "head_melt_coef": coef, "mask_gradients": mask_grad, "freeze_icefree": freeze, "steps": nsteps, "picard_total": tot_p, "vcycles_total": tot_v, "seconds": time.time() - t0,
           "head_min_max": [float(head.min()), float(head.max())], "gap_min_max": [float(gap.min()), float(gap.max())],
           "vs_reference": cmp_}
    print(json.dumps(res, indent=1))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
        np.savetxt(out_json.replace(".json", "_table.dat"), table, fmt="%.10g")


if __name__ == "__main__":
    main()
