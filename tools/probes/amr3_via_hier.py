#!/usr/bin/env python3
"""bench.py's amr3_timestep_1024x256_base_63_moulins (two nested patches, suhmo_amr_* entry points) and the SAME hierarchy through the box-union
entry points (suhmo_hier_*: one box per level): ms per step of each, and whether the heads agree bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from suhmo_amd import model, synthetic as sy
nx0, ny0, patches = 1024, 256, ((256, 64, 767, 191), (768, 192, 1279, 319))
ma = dict(sy.A3_MODEL, use_moulin_source=1, distributed_input=7.93e-11)
sts = sy.shmip_amr_states(nx0, ny0, patches)
rng = np.random.default_rng(7)
pos = np.stack([rng.uniform(3.0e4, 7.0e4, 63), rng.uniform(6.0e3, 1.4e4, 63)], axis=1)
res = {}
for which in ("amr", "hier"):
    if which == "amr":
        A = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, ma, patches, max_box=64)
        for l, st_ in enumerate(sts):
            A.set_state(l, st_)
        sync = A.levels[0].synchronize
    else:
        A = model.HipHierModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, ma, [[[2 * p[0], 2 * p[1], 2 * p[2] + 1, 2 * p[3] + 1]] for p in patches], max_box=64)      # (patches are given in the coarser level's indices)
        A.set_states([[st_] for st_ in sts])
        sync = A.level[0][0].synchronize
    A.moulin_source(pos, np.full(63, 200.0), np.full(63, 90.0 / 63), 1.0)
    for _ in range(10):
        A.timestep(ma["dt"])
    sync()
    t0 = time.perf_counter()
    nv = 0
    for _ in range(20):
        nv += A.timestep(ma["dt"])[1]
    sync()
    dt = (time.perf_counter() - t0) / 20
    print("%s: %.3f ms per step, %.2f AMR V-cycles per step" % (which, 1e3 * dt, nv / 20.0), flush=True)
    res[which] = [A.get(l, "head") if which == "amr" else A.get(l, 0, "head") for l in range(3)]
    A.close()
print("heads bit for bit:", all(np.array_equal(a, b) for a, b in zip(res["amr"], res["hier"])))
