#!/usr/bin/env python3
"""cfg5 (exec/AMR_multiMoulins physics) time step on base + 3 AMR levels of box unions: ms per step.
    python tools/hier_bench.py [base cells per side] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suhmo_amd import model, synthetic as sy
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bc, ph, mm, mo = sy.multimoulins_setup()
boxes = sy.boxes_around(mo["positions"], nb, nb, 4, 1.0e5, 1.0e5)
sts = sy.mountain_amrm_states(nb, nb, boxes)
t0 = time.perf_counter()
H = model.HipHierModel(nb, nb, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, mm, boxes, max_box=64)
H.set_states(sts)
H.moulin_source(**mo)
print("setup %.2f s, boxes per level %s, cells %s" % (time.perf_counter() - t0, [len(b) for b in boxes],
      [sum((b[2] - b[0] + 1) * (b[3] - b[1] + 1) for b in bl) for bl in boxes]), flush=True)
for _ in range(3):
    H.timestep(mm["dt"])
H.level[0][0].synchronize()
if os.environ.get("SUHMO_TIMERS"):
    from suhmo_amd import capi
    capi.lib().suhmo_timers_reset()
t0 = time.perf_counter()
c = [H.timestep(mm["dt"]) for _ in range(nstep)]
H.level[0][0].synchronize()
dt = (time.perf_counter() - t0) / nstep
print("base %d^2: %.2f ms per step, Picard %.1f, V-cycles %.1f per step" % (nb, 1e3 * dt, sum(a for a, _ in c) / nstep, sum(b for _, b in c) / nstep))
if os.environ.get("SUHMO_TIMERS"):            # named scopes (mode 2: device time per scope; serialises)
    from suhmo_amd import capi
    print(capi.timers_report())
H.close()
