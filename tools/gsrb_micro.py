#!/usr/bin/env python3
"""Bare relaxation micro-benchmark: N GSRB sweeps on an n x n SHMIP-A level.
usage: gsrb_micro.py [n] [sweeps] [reps]   (env SUHMO_GSRB_VARIANT / SUHMO_FUSED_HC select the kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suhmo_amd import level, synthetic as sy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = sy.shmip_fields(n, n)
G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS)
G.set_inputs(f)
G.update_operator()
G.gsrb(sweeps); G.synchronize()
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); G.gsrb(sweeps); G.synchronize(); dt = time.perf_counter() - t0
    best = min(best, dt / sweeps)
print("n=%d variant=%s hc=%s: %.4f ms/sweep  %.1f Gcell/s  %.0f GB/s algorithmic (72 B/cell)  frac %.3f" % (
    n, os.environ.get("SUHMO_GSRB_VARIANT", "auto"), os.environ.get("SUHMO_FUSED_HC", "auto"),
    best * 1e3, n * n / best / 1e9, 72 * n * n / best / 1e9, 72 * n * n / best / 8e12))
