#!/usr/bin/env python3
"""What the solve loop adds to a V-cycle (residual pass, max norm, read-back) at a given size.  usage: solve_vs_vcycle.py [n] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from suhmo_amd import level, synthetic as sy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = sy.shmip_fields(n, n, ly=1.0e5)
sp = dict(sy.SOLVER_DEFAULT, eps=1e-30, norm_thresh=1e-30, hang=-1.0, max_iter=iters, imin=iters, iter_min=iters)
G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, 64)
G.set_inputs(f)
G.build_mg_coefficients()
for _ in range(5):
    G.vcycle(sp)
G.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    G.vcycle(sp)
G.synchronize()
tv = (time.perf_counter() - t0) / iters
G.solve(dict(sp, max_iter=3, imin=3, iter_min=3))
G.synchronize()
t0 = time.perf_counter()
k, hist = G.solve(sp)
G.synchronize()
ts = (time.perf_counter() - t0) / k
print("%dx%d: V-cycle %.1f us, solve iteration %.1f us (+%.1f us = %.1f %%), residual left by the last launch: %d of %d"
      % (n, n, 1e6 * tv, 1e6 * ts, 1e6 * (ts - tv), 100 * (ts - tv) / tv, G.get_option("residual_in_relax_launches"), k + 3), flush=True)
G.close()
