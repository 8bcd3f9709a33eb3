"""The caller of the head solve on the GPU (suhmo_level_timestep: Picard loop, melt rate, RHS assembly,
gap-height update) against the oracle's time loop on the same inputs: BITWISE over several steps, and the
whole SHMIP A3 run (10002 steps) against the reference's committed result table."""
import os

import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hipmodel():
    from suhmo_amd import capi, model
    assert capi.lib().suhmo_device_count() > 0, "no GPU visible: the product path has no fallback"
    return model


def perturbed_state(nx, ny, seed, mask_holes=False, ly=2.0e4):
    st = sy.shmip_initial_state(nx, ny, ly=ly)
    rng = np.random.default_rng(seed)
    st["B"] = st["B"] * rng.uniform(0.5, 12.0, size=st["B"].shape)     # both sides of br = 0.1
    st["head"] = st["head"] + rng.uniform(0.0, 30.0, size=st["head"].shape)
    st["zb"] = st["zb"] + rng.uniform(0.0, 2.0, size=st["zb"].shape)
    if mask_holes:
        st["mask"][rng.uniform(size=st["mask"].shape) < 0.05] = -1.0
    return st


CASES = [
    ("a3-2048x512-fused-kernels", 2048, 512, sy.A3_BC, sy.A3_PHYS, dict(), False, 1),    # depth 0 on the fused GSRB / bCoef kernels
    ("a3-32x16", 32, 16, sy.A3_BC, sy.A3_PHYS, dict(), False, 3),
    ("a3-128x32-perturbed", 128, 32, sy.A3_BC, sy.A3_PHYS, dict(), False, 3),
    ("yperiodic-mask", 64, 32, sy.CONV_BC, dict(sy.A3_PHYS, use_mask_gradients=1, cutOffbr=0.02, maxOffbr=0.08, cutOffB=1),
     dict(use_mask_rhs_b=1, G=0.05), True, 2),
    ("dirichlet-values", 48, 16, dict(type=[[0, 0], [1, 1]], value=[[2.0, 900.0], [0.0, 0.0]], periodic=[0, 0]),
     sy.A3_PHYS, dict(basal_friction=0), False, 2),
    ("implicit-gap-changing-dt", 64, 32, sy.A3_BC, sy.A3_PHYS, dict(diffFactor=1.0, use_impl_diff=1), False, 3),   # the gap operator takes a new beta per step size
]


@pytest.mark.parametrize("name,nx,ny,bc,ph,mpo,holes,nsteps", CASES, ids=[c[0] for c in CASES])
def test_timestep_bitwise(oracle, hipmodel, name, nx, ny, bc, ph, mpo, holes, nsteps):
    m = dict(sy.A3_MODEL, **mpo)
    st = sy.shmip_initial_state(nx, ny) if name == "a3-32x16" else perturbed_state(nx, ny, 11, holes)
    mb = 64 if nx >= 1024 else 16
    O = oracle.OracleModel(nx, ny, st["dx"], st["dy"], bc, ph, m, max_box=mb, nthreads=8 if nx >= 1024 else 2)
    G = hipmodel.HipModel(nx, ny, st["dx"], st["dy"], bc, ph, m, max_box=mb)
    O.set_state(st)
    G.set_state(st)
    v = lambda a: np.array(a)[1:-1, 1:-1]
    for k in range(nsteps):
        dt = m["dt"] * (0.5 if k == 1 else 1.0)          # (a changing step size)
        po, vo = O.timestep(dt)
        pg, vg = G.timestep(dt)
        assert (po, vo) == (pg, vg), (k, po, vo, pg, vg)
        for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("mR", oracle.OM_MR), ("Pw", oracle.OM_PW),
                        ("cd", oracle.OM_CD), ("rhs_h", oracle.OM_RHSH)):
            a, b = v(O.field(fid)), G.get(nm)
            assert np.array_equal(a, b, equal_nan=True), (name, k, nm, float(np.nanmax(np.abs(a - b))))   # cd = 0/0 where nothing opens the gap
        for nm, fid in (("qwx", oracle.OM_QWX), ("qwy", oracle.OM_QWY)):
            a, b = np.array(O.field(fid)), G.get(nm)
            assert np.array_equal(a, b, equal_nan=True), (name, k, nm, float(np.nanmax(np.abs(a - b))))   # cd = 0/0 where nothing opens the gap
        # ghosts of the gap height after the step (CopyGhostCells): edges, corners are never read
        a, b = np.array(O.field(oracle.OM_B)), G.get("B", ghosted=True)
        assert np.array_equal(a[1:-1, :], b[1:-1, :]) and np.array_equal(a[:, 1:-1], b[:, 1:-1])
        assert np.all(np.isfinite(b))
    O.close()
    G.close()


def test_timestep_refuses_what_is_not_built(hipmodel):
    from suhmo_amd import capi
    st = sy.shmip_initial_state(32, 16)
    for bad in (dict(use_impl_diff=1, diffFactor=0.0),          # implicit diffusion without diffusion
                dict(use_moulin_source=1)):                     # moulin source that was never computed
        G = hipmodel.HipModel(32, 16, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, dict(sy.A3_MODEL, **bad))
        G.set_state(st)
        with pytest.raises(capi.SuhmoError):
            G.timestep(3600.0)
        G.close()


@pytest.mark.parametrize("case", ["A1", "A2", "A3", "A4", "A5", "A6"])
def test_shmip_a_full_run(hipmodel, case):
    """exec/A_SHMIP/A<k>: 320 x 64, dt = 1 h, 10000 + 2 steps, all on the device.  Gates:
    (1) same Picard-iteration and V-cycle totals as the oracle's committed run, and table == the oracle's
        (tests/golden/shmip_A<k>_oracle_run_table.dat, written with 10 significant digits) to 1e-9 relative:
        the GPU path stays on the oracle's trajectory for 10002 steps;
    (2) table vs the REFERENCE's committed result (exec/A_SHMIP/A<k>/results/postproc.dat): the tolerances of
        test_oracle_timeloop.ASIS_TOL (the tables were written without the melt term of RHS_h; the oracle with
        that one term off matches them to print precision -- test_oracle_pinned_by_reference_results)."""
    import json
    from test_oracle_timeloop import check_against_reference
    m = sy.shmip_a_model(case)
    st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
    G = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=64)
    G.set_state(st)
    tot_p = tot_v = 0
    for k in range(m["max_step"] + 2):
        p, nv = G.timestep(m["dt"])
        tot_p += p
        tot_v += nv
    table = G.postproc_table()
    tdev = G.postproc_table_device()                      # the same table reduced on the device
    assert np.all(np.abs(tdev - table) <= 1e-12 * np.max(np.abs(table), axis=0))
    G.close()
    orc = np.loadtxt(os.path.join(GOLD, "shmip_%s_oracle_run_table.dat" % case))
    run = json.load(open(os.path.join(GOLD, "shmip_%s_oracle_run.json" % case)))
    assert (tot_p, tot_v) == (run["picard_total"], run["vcycles_total"])
    scale = np.max(np.abs(orc), axis=0)
    assert np.all(np.abs(table - orc) <= 1e-9 * scale), np.max(np.abs(table - orc) / scale, axis=0)
    check_against_reference(table, case, "run")


@pytest.mark.parametrize("case", ["B1", "B5"])
def test_shmip_b_full_run(hipmodel, oracle, case):
    """exec/B_SHMIP/B<k>: suite A physics + 1 / 100 moulins, diffFactor = 1, implicit gap-height solve; 10000 + 2 steps on
    the device.  The moulin source array is the oracle's (its exp() differs from the device's in the last bits,
    tests/test_gpu_moulin.py bounds that), so the run must stay on the oracle's trajectory: same Picard / V-cycle
    totals, table equal to 1e-9."""
    import json
    from suhmo_amd import level as lv
    from test_oracle_timeloop import check_against_reference
    binp = json.load(open(os.path.join(GOLD, "shmip_B_inputs.json")))[case]
    m = sy.shmip_b_model(case, binp)
    st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
    src, _ = oracle.moulin_source(m["nx"], m["ny"], st["dx"], st["dy"], np.array(binp["positions"]).reshape(-1, 2),
                                  binp["sigma"], binp["flux"], 1.0)
    G = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=64)
    G.set_state(st)
    G.level.set(lv.F_MSRC, src)
    tot_p = tot_v = 0
    for k in range(m["max_step"] + 2):
        p, nv = G.timestep(m["dt"])
        tot_p += p
        tot_v += nv
    mask = G.get("mask")
    table = sy.shmip_postproc_table(st["dx"], st["dy"], G.get("qwx"), G.get("cd", ghosted=True),
                                    np.where(mask > 0.0, src * m["ramp"] + m["distributed_input"], 0.0),
                                    G.get("mR"), G.get("Pw"), G.get("Pi"), mask, m["rho_w"])
    G.close()
    orc = np.loadtxt(os.path.join(GOLD, "shmip_%s_oracle_run_table.dat" % case))
    run = json.load(open(os.path.join(GOLD, "shmip_%s_oracle_run.json" % case)))
    assert (tot_p, tot_v) == (run["picard_total"], run["vcycles_total"])
    scale = np.max(np.abs(orc), axis=0)
    assert np.all(np.abs(table - orc) <= 1e-9 * scale), np.max(np.abs(table - orc) / scale, axis=0)
    check_against_reference(table, case, "run")


def test_convergence_distributed_reference_table(hipmodel):
    """exec/1_convergence_distributed: five runs of 5000 steps (64 x 16 ... 1024 x 256, y-periodic sqrt ice sheet, diffusion and
    implicit gap-height solve, dt = 2 h) and the L2 self-convergence errors of head, gap height, water pressure and Reynolds
    number between successive resolutions -- the table the reference commits (CONV_ANA/results/convergence_data.dat, copied as
    a data fixture).  The device path reproduces all 16 numbers to the 5 digits they are printed with, with the source and the
    inputs as they are: a direct pin of the HIP path against the reference's own output."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "tools"))
    import convergence_distributed as cd
    ref = {int(r[0]): r[1:] for r in np.loadtxt(os.path.join(GOLD, "convergence_distributed_reference.dat"))}
    got = cd.table("hip", 5000, 1024)
    assert sorted(got) == sorted(ref) == [64, 128, 256, 512]
    for nx in ref:
        for k, (a, b) in enumerate(zip(got[nx], ref[nx])):
            assert abs(a - b) <= 6e-5 * abs(b), (nx, ("head", "gapHeight", "Pw", "Re")[k], a, b)      # 5 significant digits


@pytest.mark.parametrize("case", ["F1", "F5"])
def test_shmip_f_five_year_series_on_the_device(hipmodel, case):
    """exec/F_SHMIP/F1 (deltaT = -6 K) and F5 (+6 K: the strongest forcing) on the device: 8000 spin-up steps of 1 h, then five years of 2 h steps under the seasonal temperature cycle
    (suhmo_level_time_varying_recharge before every step), the daily series of tools/run_shmip_f.py.  The source and the solver
    inputs as committed (melt term in RHS_h, no masked gradients, gap height of ice-free cells evolving); the height field the lapse
    rate reads holds the surface elevation, as in the run that wrote the reference's table (tests/test_oracle_timeloop.py).  Gates:
    (1) the oracle's committed series of the same configuration to 1e-9 and its Picard / V-cycle totals: 29960 steps on the oracle's
    trajectory through a transient whose discharge swings over three orders of magnitude; (2) the reference's series within what the
    run-state settings of the pin change: the winter discharge by the melt share (4 %), the effective pressures below 1 %."""
    import json, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "f1.json")
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "run_shmip_f.py"), "hip", case, "5", out, "--zs", "surface"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        got, res = np.loadtxt(out.replace(".json", "_table.dat")), json.load(open(out))
    orc = np.loadtxt(os.path.join(GOLD, "shmip_%s_oracle_run_table.dat" % case))
    run = json.load(open(os.path.join(GOLD, "shmip_%s_oracle_run.json" % case)))
    assert (res["picard_total"], res["vcycles_total"]) == (run["picard_total"], run["vcycles_total"])
    scale = np.max(np.abs(orc), axis=0)
    assert got.shape == orc.shape == (1830, 10)
    assert np.all(np.abs(got - orc) <= 1e-9 * scale), np.max(np.abs(got - orc) / scale, axis=0)
    ref = np.loadtxt(os.path.join(GOLD, "shmip_%s_postproc_reference.dat" % case))
    for c, tol in F_ASIS_TOL[case].items():
        sc = np.max(np.abs(ref[:, c])) if c != 7 else np.abs(ref[:, c])
        assert np.max(np.abs(got[:, c] - ref[:, c]) / sc) <= tol, (c, float(np.max(np.abs(got[:, c] - ref[:, c]) / sc)))


# avgN, N_LB, N_MB, N_HB, recharge: of the column's scale; discharge: of each day's own value.  Measured: F1 3.9e-3, 4.2e-5, 3.1e-3,
# 6.6e-3, 5.7e-5, 4.3e-2; F5 6.3e-3, 1.6e-3, 6.5e-3, 8.4e-3, 2.3e-4, 5.0e-2 (the melt share in the winter discharge, the leak of gap
# height across the ice margin)
F_ASIS_TOL = {"F1": {2: 6e-3, 3: 1e-4, 4: 5e-3, 5: 1e-2, 6: 1e-4, 7: 6e-2}, "F5": {2: 9e-3, 3: 3e-3, 4: 9e-3, 5: 1.2e-2, 6: 4e-4, 7: 7e-2}}


@pytest.mark.parametrize("case", ["E1", "E4"])
def test_suite_e_masked_margin_on_the_device(oracle, hipmodel, case):
    """SHMIP suite E (valley glacier, oblique ice margin: ice mask, cut_solve_outside_domain, masked gradients, use_mask_rhs_b,
    diffusion + implicit gap-height solve, exec/E_SHMIP/E<k>/input.hydro) -- the configuration whose tables pin the masked
    branches of the oracle (tests/test_oracle_timeloop.py): the device steps it bit for bit with the oracle (source as it is on
    both sides), 40 steps from the reference's initial state, ice-free cells and margin faces included."""
    m = sy.shmip_e_model(case)
    st = sy.valley_initial_state(m["nx"], m["ny"], sy.E_GAMMA[case], m["lx"], m["ly"])
    phys = dict(sy.E_PHYS, cutOffB=sy.E_CUTOFFB[case])
    O = oracle.OracleModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64, nthreads=8)
    G = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64)
    O.set_state(st); G.set_state(st)
    from suhmo_amd import level as lv
    O.field(oracle.OM_MR)[:] = m["G"] / m["L"]
    G.level.set(lv.F_MR, np.full((m["ny"], m["nx"]), m["G"] / m["L"]))
    v = lambda a: np.array(a)[1:-1, 1:-1]
    mask = v(O.field(oracle.OM_MASK))
    assert (mask < 0).sum() > 1000 and (mask > 0).sum() > 1000
    for k in range(40):
        co, cg = O.timestep(m["dt"]), G.timestep(m["dt"])
        assert co == cg, (k, co, cg)
    for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("mR", oracle.OM_MR), ("Pw", oracle.OM_PW), ("Re", oracle.OM_RE)):
        a, b = v(O.field(fid)), G.get(nm)
        assert np.array_equal(a, b, equal_nan=True), (case, nm, float(np.nanmax(np.abs(a - b))))
    for nm, fid in (("qwx", oracle.OM_QWX), ("qwy", oracle.OM_QWY)):
        assert np.array_equal(np.array(O.field(fid)), G.get(nm), equal_nan=True), (case, nm)
    O.close(); G.close()


# ---- the DEVICE path against the reference's committed result tables, with the run-state settings those tables were written
# with as model options of the C-ABI (suhmo_model_params_t.head_melt_off / freeze_icefree_gap, phys.use_mask_gradients; DESIGN.md
# section 4).  Tolerances = the oracle pin's (tests/test_oracle_timeloop.PIN_TOL, tests/golden/PIN_REPORT.txt), not ASIS_TOL.
PIN_RUNS = {"A3": dict(mask_gradients=1), "B5": dict(mask_gradients=1), "E1": dict(freeze=1)}


@pytest.mark.parametrize("case", ["A3", "B5", "E1"])
def test_device_reproduces_the_reference_table(hipmodel, oracle, case):
    """exec/A_SHMIP/A3, exec/B_SHMIP/B5, exec/E_SHMIP/E1 results/postproc.dat (320 / 256 rows x 8 columns) from a run of the HIP
    path alone: 10002 (E1: 5002) steps on the device, every column and row to the print precision of the table (5e-6 / 5e-7 /
    1.2e-5 of the column scales), and on the trajectory of the oracle's committed pin run (same Picard / V-cycle totals, table to 1e-9)."""
    import json
    from suhmo_amd import level as lv
    from test_oracle_timeloop import check_against_reference
    binp = None
    if case[0] == "B":
        binp = json.load(open(os.path.join(GOLD, "shmip_B_inputs.json")))[case]
        m = sy.shmip_b_model(case, binp)
    elif case[0] == "E":
        m = sy.shmip_e_model(case)
    else:
        m = sy.shmip_a_model(case)
    m = dict(m, head_melt_off=1, freeze_icefree_gap=PIN_RUNS[case].get("freeze", 0))
    if case[0] == "E":
        st = sy.valley_initial_state(m["nx"], m["ny"], sy.E_GAMMA[case], m["lx"], m["ly"])
        phys = dict(sy.E_PHYS, cutOffB=sy.E_CUTOFFB[case])
    else:
        st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
        phys = dict(sy.A3_PHYS, use_mask_gradients=1)                 # the tables' run had it on (inputs of suites A / B say off)
    G = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64)
    G.set_state(st)
    G.level.set(lv.F_MR, np.full((m["ny"], m["nx"]), m["G"] / m["L"]))
    src = None
    if binp:
        src, _ = oracle.moulin_source(m["nx"], m["ny"], st["dx"], st["dy"], np.array(binp["positions"]).reshape(-1, 2), binp["sigma"], binp["flux"], 1.0)
        G.level.set(lv.F_MSRC, src)
    tot_p = tot_v = 0
    for k in range(m["max_step"] + 2):
        p, nv = G.timestep(m["dt"])
        tot_p += p
        tot_v += nv
    if binp:
        mask = G.get("mask")
        table = sy.shmip_postproc_table(st["dx"], st["dy"], G.get("qwx"), G.get("cd", ghosted=True), np.where(mask > 0.0, src * m["ramp"] + m["distributed_input"], 0.0),
                                        G.get("mR"), G.get("Pw"), G.get("Pi"), mask, m["rho_w"])
    else:
        table = G.postproc_table()
    G.close()
    check_against_reference(table, case, "pin")                                  # the reference's own numbers
    orc = np.loadtxt(os.path.join(GOLD, "shmip_%s_oracle_pin_table.dat" % case))
    run = json.load(open(os.path.join(GOLD, "shmip_%s_oracle_pin.json" % case)))
    assert (tot_p, tot_v) == (run["picard_total"], run["vcycles_total"])
    scale = np.max(np.abs(orc), axis=0)
    assert np.all(np.abs(table - orc) <= 1e-9 * scale), np.max(np.abs(table - orc) / scale, axis=0)


# F1: largest deviation of the oracle's pin run from the reference per column (PIN_REPORT.txt): 2.3e-7, 2.8e-7, 2.1e-7, 1.9e-7, 4.3e-7, 1.3e-5
F_PIN_TOL = {2: 1e-6, 3: 1e-6, 4: 1e-6, 5: 1e-6, 6: 1e-6, 7: 3e-5}


def test_device_reproduces_the_reference_five_year_series(hipmodel):
    """exec/F_SHMIP/F1/results/postproc.dat, 1830 daily rows of a five-year seasonal cycle, from a run of the HIP path alone with the
    run-state settings of the table as model options / inputs / data (no melt term in RHS_h, masked gradients and masked gap-height
    right-hand side, ice-free gap height frozen, surface elevation as the lapse-rate field): every day and column to print precision."""
    import json, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "f1.json")
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "run_shmip_f.py"), "hip", "F1", "5", out, "--zs", "surface", "--head-melt-coef", "0",
                            "--mask-gradients", "1", "--mask-rhs-b", "1", "--freeze-icefree"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        got, res = np.loadtxt(out.replace(".json", "_table.dat")), json.load(open(out))
    ref = np.loadtxt(os.path.join(GOLD, "shmip_F1_postproc_reference.dat"))
    assert got.shape[0] == ref.shape[0] == 1830
    for c, tol in F_PIN_TOL.items():
        sc = np.max(np.abs(ref[:, c]))
        assert np.max(np.abs(got[:, c] - ref[:, c])) <= tol * sc, (c, float(np.max(np.abs(got[:, c] - ref[:, c])) / sc))
    orc = np.loadtxt(os.path.join(GOLD, "shmip_F1_oracle_pin_table.dat"))
    run = json.load(open(os.path.join(GOLD, "shmip_F1_oracle_pin.json")))
    assert (res["picard_total"], res["vcycles_total"]) == (run["picard_total"], run["vcycles_total"])
    scale = np.max(np.abs(orc), axis=0)
    assert np.all(np.abs(got - orc) <= 1e-9 * scale), np.max(np.abs(got - orc) / scale, axis=0)


def test_run_state_options_bitwise(oracle, hipmodel):
    """head_melt_off and freeze_icefree_gap on both sides (model options, no environment): 30 steps of suite E1 (ice margin, implicit gap
    solve) bit for bit; the ice-free cells keep their gap height, and without the option they do not."""
    case = "E1"
    m = dict(sy.shmip_e_model(case), head_melt_off=1, freeze_icefree_gap=1)
    st = sy.valley_initial_state(m["nx"], m["ny"], sy.E_GAMMA[case], m["lx"], m["ly"])
    phys = dict(sy.E_PHYS, cutOffB=sy.E_CUTOFFB[case])
    from suhmo_amd import level as lv
    O = oracle.OracleModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64, nthreads=8)
    G = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, m, max_box=64)
    G2 = hipmodel.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, phys, dict(m, freeze_icefree_gap=0), max_box=64)
    for M in (O, G, G2):
        M.set_state(st)
    O.field(oracle.OM_MR)[:] = m["G"] / m["L"]
    for M in (G, G2):
        M.level.set(lv.F_MR, np.full((m["ny"], m["nx"]), m["G"] / m["L"]))
    v = lambda a: np.array(a)[1:-1, 1:-1]
    for k in range(30):
        co, cg = O.timestep(m["dt"]), G.timestep(m["dt"])
        G2.timestep(m["dt"])
        assert co == cg, (k, co, cg)
    for nm, fid in (("head", oracle.OM_H), ("B", oracle.OM_B), ("mR", oracle.OM_MR), ("rhs_h", oracle.OM_RHSH)):
        a, b = v(O.field(fid)), G.get(nm)
        assert np.array_equal(a, b, equal_nan=True), (nm, float(np.nanmax(np.abs(a - b))))
    icefree = st["mask"][1:-1, 1:-1] < 0
    assert np.array_equal(G.get("B")[icefree], st["B"][1:-1, 1:-1][icefree])
    assert not np.array_equal(G2.get("B")[icefree], st["B"][1:-1, 1:-1][icefree])
    for M in (O, G, G2):
        M.close()


def test_iteration_pattern_of_the_tutorial_run_on_the_device(hipmodel, oracle):
    """the tutorial run of docs/GettingStarted.md (exec/0_convergence_channelized/1lev: 32 x 8, 3000 steps, ramped moulin) on the device:
    the Picard / FAS iteration counts of every step equal the oracle's, and meet what the reference says about them as far as the
    reconstructed cycle does (tests/test_oracle_timeloop.check_tutorial_pattern)"""
    from test_oracle_timeloop import tutorial_iteration_counts, check_tutorial_pattern
    pg, po_ = tutorial_iteration_counts("hip"), tutorial_iteration_counts("oracle")
    assert np.array_equal(pg, po_), np.where(np.any(pg != po_, axis=1))[0][:5]
    check_tutorial_pattern(pg)
