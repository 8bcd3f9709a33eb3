#!/usr/bin/env python3
"""Depth 1 of the bench V-cycle (2048^2 on the streaming kernel, 300 MB of working set against 256 MiB of memory-side cache): are the
1.59 x of its measured "HBM" bytes over the compulsory ones served by the cache?  The same launches back to back, and with the cache
flushed between them by a pass over a 4096^2 level (reads 268 MB, writes 134 MB).  usage: depth1_mall.py [n] [hc ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from suhmo_amd import level, synthetic as sy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hcs = [int(a) for a in sys.argv[2:]] or [0]
f = sy.shmip_fields(n, n)
G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS)
G.set_inputs(f); G.update_operator()
g = sy.shmip_fields(4096, 4096)
F = level.HipLevel(4096, 4096, g["dx"], g["dy"], sy.A3_BC, sy.A3_PHYS)
F.set_inputs(g)
for hc in hcs:
    G.set_option("fused_hc", hc)
    for flush in (0, 1):
        G.gsrb(2); G.synchronize()
        G.profile(True)
        for _ in range(24):
            if flush:
                F.axby(level.F_RES, level.F_PHI, level.F_RHS, 1.0, 1.0)
            G.gsrb(2)
        G.synchronize()
        ms, nl, nc = G.profile_read()
        G.profile(False)
        print("n=%d hc=%s %-26s %.1f us per 2-sweep launch (%d launches)" % (n, hc or "auto", "cache flushed in between" if flush else "back to back", 1e3 * ms / max(nl, 1), nl), flush=True)
