#!/usr/bin/env python3
"""AMR hierarchy (base + nested patches) cut into rank strips, one process per rank (torch.distributed.run).
Every rank holds the rows of its own slab of every level; suhmo_amd.multigpu.attach_amr gives every level its own
communicator.  Runs a few AMR V-cycles and checks the composite residual norm against the single-process hierarchy
(rank 0 recomputes it).  SUHMO_DIST_BACKEND=gloo rehearses on one GPU."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
from suhmo_amd import capi, level, multigpu, synthetic as sy

BC = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])


def strip_of(f, j0, ny):
    s = dict(nx=f["nx"], ny=ny, dx=f["dx"], dy=f["dy"])
    for k in ("phi", "rhs", "aCoef"):
        s[k] = f[k][j0:j0 + ny]
    for k in ("B", "Pi", "zb", "mask"):
        s[k] = f[k][j0:j0 + ny + 2]
    return s


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
    dist.init_process_group(os.environ.get("SUHMO_DIST_BACKEND", "nccl"))
    nx0, ny0, MB = 64, 64, 16
    patches = ((8, 12, 39, 51), (24, 40, 59, 87)) if world <= 2 else ((8, 20, 39, 43),)
    if os.environ.get("AMR_CASE") == "full":
        patches = ((8, 12, 39, 51),)
    fs = sy.amr_fields(nx0, ny0, patches, lx=64.0, ly=64.0, moulin=(24.0, 32.0, 2.0, 30.0))
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
    nlev, n0 = 1 + len(patches), ny0 // world
    rng = [(0, ny0)] + [(2 * p[1], 2 * p[3] + 2) for p in patches]
    def own(r, l):
        lo, hi = max(rng[l][0], r * n0 * 2 ** l), min(rng[l][1], (r + 1) * n0 * 2 ** l)
        return (lo, hi - lo) if hi > lo else None
    ranges = [[r for r in range(world) if own(r, l)] for l in range(nlev)]
    lv = []
    nxg, nyg, dx, dy = nx0, ny0, fs[0]["dx"], fs[0]["dy"]
    dev = torch.cuda.current_device()
    for l in range(nlev):
        if l > 0:
            nxg, nyg, dx, dy = 2 * nxg, 2 * nyg, dx / 2.0, dy / 2.0
        o = own(rank, l)
        if not o:
            lv.append(None)
            continue
        j0, ny = o
        if l == 0:
            G = level.HipLevel(nx0, ny, dx, dy, BC, sy.CFG3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=ny0, halo_rows=4, device=dev)
            G.set_inputs(strip_of(fs[0], j0, ny))
        else:
            ci0, cj0, ci1, cj1 = patches[l - 1]
            G = level.HipLevel(2 * (ci1 - ci0 + 1), ny, dx, dy, BC, sy.CFG3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=nyg, i0=2 * ci0,
                               nx_global=nxg, halo_rows=2, patch_j0=rng[l][0], patch_ny=rng[l][1] - rng[l][0], device=dev)
            G.set_inputs(strip_of(fs[l], j0 - rng[l][0], ny))
        lv.append(G)
    keep = multigpu.attach_amr(lv, ranges, dist, rank, world)
    lv[0].build_mg_coefficients()
    arr = (C.c_void_p * nlev)(*[(g.h if g else None) for g in lv])
    s = level.solver_params(sp)
    norms = []
    for k in range(3):
        r = C.c_double()
        capi.check(capi.lib().suhmo_amr_residual(arr, nlev, C.cast(C.pointer(r), C.POINTER(C.c_double)), None))
        norms.append(r.value)
        if k < 2:
            capi.check(capi.lib().suhmo_amr_vcycle(arr, nlev, C.byref(s), None))
    mine = [(g.get(level.F_PHI) if g else None) for g in lv]
    allphi = [None] * world
    dist.all_gather_object(allphi, mine)
    ok = True
    if rank == 0:
        G = level.HipAmr(nx0, ny0, fs[0]["dx"], fs[0]["dy"], BC, sy.CFG3_PHYS, patches, max_box=MB, device=dev)
        G.levels[0].set_inputs(fs[0]); G.levels[0].build_mg_coefficients()
        for l in range(1, nlev):
            G.levels[l].set_inputs(fs[l])
        ref = []
        for k in range(3):
            ref.append(G.residual())
            if k < 2:
                G.vcycle(sp)
        ok = ref == norms
        for l in range(nlev):
            got = np.vstack([allphi[r][l] for r in range(world) if allphi[r][l] is not None])
            d = np.abs(got - G.levels[l].get(level.F_PHI))
            rows = np.where(d.max(axis=1) > 0)[0]
            print("  level %d head: max |diff| %.3g, rows with a difference: %s" % (l, d.max(), (int(rows.min()), int(rows.max()), len(rows)) if len(rows) else None), flush=True)
        print("AMR strips on %d ranks (%d levels): composite residual norms %s; single process %s -> %s"
              % (world, nlev, norms, ref, "BITWISE EQUAL" if ok else "MISMATCH"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
