#!/usr/bin/env python3
"""Cost of the halo exchanges of a V-cycle: an n x n periodic level whole, and the same level as ONE rank strip that is its own
neighbour over the native RCCL transport (ncclSend / ncclRecv to itself on the kernels' stream).  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/strip_probe.py strip` the pack / transport / unpack kernels of the 14 exchanges.
SUHMO_TRANSPORT=ipc: the strip's halo rows go peer-direct (suhmo_ipc.hip: the pack kernel stores into the receive slots -- its own, being
its own neighbour -- and the unpack kernel waits on a flag word), RCCL keeps the reductions.
usage: strip_probe.py [whole|strip|both] [n] [cycles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: the library binds to the HIP runtime PyTorch ships)
from suhmo_amd import capi, level, multigpu, synthetic as sy

which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 10
f = sy.wrap_ghosts(sy.shmip_fields(n, n), sy.CONV_BC)
sp = dict(sy.SOLVER_DEFAULT)
for name in ("whole", "strip"):
    if which not in (name, "both"):
        continue
    if name == "whole":
        L = level.HipLevel(n, n, f["dx"], f["dy"], sy.CONV_BC, sy.A3_PHYS, 0.0, -1.0, 64)
        L.set_inputs(f)
    else:
        L = level.HipLevel(n, n, f["dx"], f["dy"], sy.CONV_BC, sy.A3_PHYS, 0.0, -1.0, 64, j0=0, ny_global=2 * n,
                           halo_rows=int(os.environ.get("SUHMO_HALO_ROWS", "24")))
        L.set_inputs(f)
        multigpu.attach_rccl(L, 0, 1, periodic_y=True)
        if os.environ.get("SUHMO_TRANSPORT") == "ipc":
            multigpu.ipc_attach(L, 0, 1, True, [multigpu.ipc_export(L)])
    L.build_mg_coefficients()
    for _ in range(3):
        L.vcycle(sp)
    L.synchronize()
    e0 = L.rccl_exchanges() if name == "strip" else 0
    t0 = time.perf_counter()
    for _ in range(cycles):
        L.vcycle(sp)
    L.synchronize()
    dt = (time.perf_counter() - t0) / cycles
    ex = (L.rccl_exchanges() - e0) / cycles if name == "strip" else 0
    print("%s %dx%d%s: %.3f ms per V-cycle, %.1f exchanges per V-cycle" % (name, n, n, " [%s]" % getattr(L, "_transport", "rccl") if name == "strip" else "", dt * 1e3, ex), flush=True)
    L.close()
