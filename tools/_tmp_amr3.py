import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from suhmo_amd import model, synthetic as sy
nx0, ny0, patches = 1024, 256, ((256, 64, 767, 191), (768, 192, 1279, 319))
ma = dict(sy.A3_MODEL, use_moulin_source=1, distributed_input=7.93e-11)
sts = sy.shmip_amr_states(nx0, ny0, patches)
A = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, ma, patches, max_box=64)
for l, st_ in enumerate(sts): A.set_state(l, st_)
rng = np.random.default_rng(7)
pos = np.stack([rng.uniform(3.0e4, 7.0e4, 63), rng.uniform(6.0e3, 1.4e4, 63)], axis=1)
A.moulin_source(pos, np.full(63, 200.0), np.full(63, 90.0 / 63), 1.0)
for _ in range(10): A.timestep(ma["dt"])
A.levels[0].synchronize()
t0 = time.perf_counter(); nv = 0; npi = 0
for _ in range(20):
    r = A.timestep(ma["dt"]); nv += r[1]; npi += r[0]
A.levels[0].synchronize()
dt = (time.perf_counter() - t0) / 20
print("ms/step %.3f  vcycles/step %.2f picard/step %.2f" % (1e3 * dt, nv / 20, npi / 20))
