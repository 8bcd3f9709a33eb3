"""diagnostic: which librccl can the native transport use in this process?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "torch"
from suhmo_amd import capi, level, multigpu, synthetic as sy
if which == "torch":          # torch first: this library binds to the HIP runtime PyTorch ships
    import torch
f = sy.shmip_fields(64, 32)
S = level.HipLevel(64, 32, f["dx"], f["dy"], sy.CONV_BC, sy.A3_PHYS, 0.0, -1.0, 32, j0=0, ny_global=64, halo_rows=4)
S.set_inputs(f)
multigpu.attach_rccl(S, 0, 1, periodic_y=True)
print("attached", flush=True)
os.system("grep -E 'hip|rccl' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
S.build_mg_coefficients(); S.vcycle(dict(sy.SOLVER_DEFAULT)); S.synchronize()
print("vcycle ok, exchanges", capi.lib().suhmo_level_rccl_exchanges(S.h))

# cost of the exchanges: 4096^2 whole periodic level vs the same level as a self-neighbour strip
import time
wrap_ghosts = sy.wrap_ghosts
n = int(os.environ.get("PROBE_N", "4096"))
f = wrap_ghosts(sy.shmip_fields(n, n), sy.CONV_BC)
W = level.HipLevel(n, n, f["dx"], f["dy"], sy.CONV_BC, sy.A3_PHYS, 0.0, -1.0, 64)
W.set_inputs(f)
S2 = level.HipLevel(n, n, f["dx"], f["dy"], sy.CONV_BC, sy.A3_PHYS, 0.0, -1.0, 64, j0=0, ny_global=2 * n, halo_rows=int(os.environ.get("SUHMO_HALO_ROWS", "24")))
S2.set_inputs(f)
multigpu.attach_rccl(S2, 0, 1, periodic_y=True)
sp = dict(sy.SOLVER_DEFAULT)
only = os.environ.get("PROBE_ONLY")
for name, L in (("whole", W), ("strip+rccl-self", S2)):
    if only and only not in name:
        continue
    L.build_mg_coefficients()
    for _ in range(3):
        L.vcycle(sp)
    L.synchronize()
    e0 = capi.lib().suhmo_level_rccl_exchanges(L.h)
    t0 = time.perf_counter()
    for _ in range(10):
        L.vcycle(sp)
    L.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s: %.3f ms / V-cycle, exchanges per V-cycle %s" % (name, dt * 1e3, (capi.lib().suhmo_level_rccl_exchanges(L.h) - e0) / 10))
