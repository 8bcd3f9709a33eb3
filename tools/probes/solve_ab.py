#!/usr/bin/env python3
"""A/B of the solve loop's residual evaluation: riding on the launch that ends each V-cycle (level option resid_in_relax = 1) against its own
pass (0).  usage: solve_ab.py [n] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from suhmo_amd import level, synthetic as sy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sy.shmip_fields(n, n, ly=1.0e5)
sp = dict(sy.SOLVER_DEFAULT, eps=1e-30, norm_thresh=1e-30, hang=-1.0, max_iter=iters, imin=iters, iter_min=iters)
for on in (1, 0, 1, 0):
    G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64)
    G.set_inputs(f)
    G.set_option("resid_in_relax", on)
    G.build_mg_coefficients()
    G.solve(dict(sp, max_iter=2, imin=2, iter_min=2))
    G.synchronize()
    t0 = time.perf_counter()
    k, hist = G.solve(sp)
    G.synchronize()
    dt = time.perf_counter() - t0
    print("%dx%d resid_in_relax=%d: %d iterations, %.3f ms per iteration (V-cycle + residual + norm), last norm %.6e, fused launches %d"
          % (n, n, on, k, 1e3 * dt / k, hist[-1], G.get_option("residual_in_relax_launches")), flush=True)
    G.close()
