#!/usr/bin/env python3
"""UpdateOperator (WFlx_level) micro-benchmark at n x n: usage bcoef_micro.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suhmo_amd import level, synthetic as sy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sy.shmip_fields(n, n)
G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS)
G.set_inputs(f)
for _ in range(3):
    G.update_operator()
G.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    G.update_operator()
G.synchronize()
dt = (time.perf_counter() - t0) / reps
print("n=%d update_operator %.4f ms  %.0f GB/s algorithmic (40 B/cell)" % (n, dt * 1e3, 40 * n * n / dt / 1e9))
