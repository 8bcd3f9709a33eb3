#!/usr/bin/env python3
"""Host + device latency of a SHMIP-sized time step (320 x 64, A3) and of a small 3-level hierarchy step: ms per step, Picard
iterations and V-cycles per step, and -- with --timers -- the named scopes (host wall time).  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/step_latency.py --steps 50` the kernel list of a step.
usage: step_latency.py [--steps N] [--warm N] [--timers] [--amr]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suhmo_amd import capi, model, synthetic as sy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warm", type=int, default=60)
    ap.add_argument("--timers", action="store_true")
    ap.add_argument("--amr", action="store_true")
    ap.add_argument("--vcycle", action="store_true", help="V-cycles and residual norms alone on the 320 x 64 level")
    a = ap.parse_args()
    m = sy.A3_MODEL
    if a.vcycle:
        from suhmo_amd import level
        f = sy.shmip_fields(m["nx"], m["ny"], lx=m["lx"], ly=m["ly"])
        G = level.HipLevel(m["nx"], m["ny"], f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64)
        G.set_inputs(f); G.build_mg_coefficients()
        sp = dict(sy.SOLVER_DEFAULT)
        for _ in range(20):
            G.vcycle(sp)
        G.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            G.vcycle(sp)
        G.synchronize()
        tv = (time.perf_counter() - t0) / a.steps
        t0 = time.perf_counter()
        for _ in range(a.steps):
            G.residual(); G.norm(level.F_RES, 0)
        tn = (time.perf_counter() - t0) / a.steps
        print("320x64: %.1f us per V-cycle (graph replay, %d depths), %.1f us per residual + max norm (read back)" % (1e6 * tv, G.ndepth, 1e6 * tn))
        G.close()
        return
    if not a.amr:
        st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
        M = model.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=64)
        M.set_state(st)
        step, sync = (lambda: M.timestep(m["dt"])), M.level.synchronize
    else:
        nx0, ny0, patches = 64, 32, ((16, 8, 47, 23), (40, 22, 79, 41))
        sts = sy.shmip_amr_states(nx0, ny0, patches)
        M = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=16)
        for l, s in enumerate(sts):
            M.set_state(l, s)
        step, sync = (lambda: M.timestep(m["dt"])), M.levels[0].synchronize
    for _ in range(a.warm):
        step()
    sync()
    if a.timers:
        capi.lib().suhmo_timers_enable(1)
        capi.lib().suhmo_timers_reset()
    t0 = time.perf_counter()
    npi = nv = 0
    for _ in range(a.steps):
        p, v = step()
        npi += p; nv += v
    sync()
    dt = (time.perf_counter() - t0) / a.steps
    print("%s: %.4f ms/step, %.2f Picard iterations/step, %.2f V-cycles/step" % ("amr3 64x32" if a.amr else "A3 320x64", 1e3 * dt, npi / a.steps, nv / a.steps))
    if a.timers:
        print(capi.timers_report())
    M.close()


if __name__ == "__main__":
    main()
