#!/usr/bin/env python3
"""cfg4 / cfg5 shape: the SHMIP suite-B time loop (moulins, diffusion, implicit gap-height solve) on a 3-level AMR
hierarchy cut into rank strips, one process per GPU:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/amr_shmip_dist.py --case B5 --steps 20
Every rank holds, of every level, the rows of its own slab; suhmo_amd.multigpu.attach_amr gives every level its own
communicator (native RCCL on the "nccl" backend; SUHMO_DIST_BACKEND=gloo rehearses several ranks on one GPU).
--check: rank 0 also runs the whole hierarchy alone and compares bit for bit."""
import argparse
import ctypes as C
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SUHMO_DUMP_AFTER"):          # tests: a rank still running after that many seconds prints where every thread waits
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ["SUHMO_DUMP_AFTER"]), exit=False)
import numpy as np
import torch
import torch.distributed as dist
from suhmo_amd import capi, level as lv, model, multigpu, synthetic as sy

NAMES = ("head", "B", "mR")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="B5")
    ap.add_argument("--scale", type=int, default=1, help="refinement of the reference's 320 x 64 base grid")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29534")):
        os.environ.setdefault(k, v)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get("SUHMO_DIST_BACKEND", "nccl"))
    b = json.load(open(os.path.join(ROOT, "tests", "golden", "shmip_B_inputs.json")))[a.case]
    m = sy.shmip_b_model(a.case, b)
    pos, sg, fl = np.array(b["positions"]).reshape(-1, 2), np.array(b["sigma"], dtype=np.float64), np.array(b["flux"], dtype=np.float64)
    nx0, ny0 = m["nx"] * a.scale, m["ny"] * a.scale
    q = a.scale
    patches = ((64 * q, 8 * q, 255 * q, 55 * q), (160 * q, 36 * q, 479 * q, 91 * q))          # nested boxes around the moulin band
    sts = sy.shmip_amr_states(nx0, ny0, patches, m["lx"], m["ly"])
    nlev, MB = 3, 16
    assert ny0 % world == 0 and (ny0 // world) % MB == 0, "rows must split into whole boxes per rank"
    n0 = ny0 // world
    rng = [(0, ny0)] + [(2 * p[1], 2 * p[3] + 2) for p in patches]
    def own(r, l):
        lo, hi = max(rng[l][0], r * n0 * 2 ** l), min(rng[l][1], (r + 1) * n0 * 2 ** l)
        return (lo, hi - lo) if hi > lo else None
    ranges = [[r for r in range(world) if own(r, l)] for l in range(nlev)]
    levels = []
    nxg, nyg, dx, dy = nx0, ny0, sts[0]["dx"], sts[0]["dy"]
    for l in range(nlev):
        if l > 0:
            nxg, nyg, dx, dy = 2 * nxg, 2 * nyg, dx / 2.0, dy / 2.0
        o = own(rank, l)
        if not o:
            levels.append(None)
            continue
        j0, ny = o
        if l == 0:
            G = lv.HipLevel(nx0, ny, dx, dy, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=ny0, halo_rows=4 if world > 1 else 1, device=dev)
        else:
            p = patches[l - 1]
            G = lv.HipLevel(2 * (p[2] - p[0] + 1), ny, dx, dy, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=nyg, i0=2 * p[0], nx_global=nxg,
                            halo_rows=2, patch_j0=rng[l][0], patch_ny=rng[l][1] - rng[l][0], device=dev)
        r0 = j0 - rng[l][0]
        st = sts[l]
        G.set(lv.F_PHI, st["head"][1:-1, 1:-1][r0:r0 + ny])
        G.set(lv.F_ACOEF, np.zeros((ny, G.nx)))
        for k, fid in (("B", lv.F_B), ("Pi", lv.F_PI), ("zb", lv.F_ZB), ("mask", lv.F_MASK)):
            G.set(fid, st[k][r0:r0 + ny + 2], ghosted=True)
        levels.append(G)
    keep = multigpu.attach_amr(levels, ranges, dist, rank, world) if world > 1 else []
    arr = (C.c_void_p * nlev)(*[(g.h if g else None) for g in levels])
    boxes = (C.c_int * 8)(*[v for p in patches for v in p])
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    integ = np.zeros(sg.size)
    posf = np.ascontiguousarray(pos.reshape(-1))
    capi.check(capi.lib().suhmo_amr_moulin_source(arr, nlev, boxes, sg.size, dp(posf), dp(sg), dp(fl), 1.0, dp(integ), None))
    mp = model.model_params(m)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = []
    for k in range(a.steps):
        pi, nv = C.c_int(), C.c_int()
        capi.check(capi.lib().suhmo_amr_timestep(arr, nlev, C.byref(mp), float(m["dt"]), k + 1, C.byref(pi), C.byref(nv), None))
        counts.append((pi.value, nv.value))
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    if a.check:
        mine = [({nm: g.get(model.HipModel.FIELDS[nm]) for nm in NAMES} if g else None) for g in levels]
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        if rank == 0:
            A = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=MB, device=dev)
            for l, st in enumerate(sts):
                A.set_state(l, st)
            iref = A.moulin_source(pos, sg, fl, 1.0)
            cw = [A.timestep(m["dt"]) for _ in range(a.steps)]
            ok = cw == counts and np.array_equal(iref, integ)
            for l in range(nlev):
                for nm in NAMES:
                    got = np.vstack([allv[r][l][nm] for r in range(world) if allv[r][l] is not None])
                    eq = np.array_equal(got, A.get(l, nm), equal_nan=True)
                    ok = ok and eq
                    print("  level %d %-4s %s" % (l, nm, "bitwise equal" if eq else "DIFFERS (max %.3e)" % np.nanmax(np.abs(got - A.get(l, nm)))), flush=True)
            A.close()
    if rank == 0:
        cells = [int(s_["nx"] * s_["ny"]) for s_ in sts]
        print("SHMIP %s on a 3-level hierarchy (%s cells per level), %d rank(s): %d steps in %.2f s (%.1f steps/s), %d Picard iterations, %d AMR V-cycles%s"
              % (a.case, cells, world, a.steps, dt, a.steps / dt, sum(c[0] for c in counts), sum(c[1] for c in counts),
                 (" -> " + ("BITWISE EQUAL to the single-process hierarchy" if ok else "MISMATCH")) if a.check else ""), flush=True)
    for g in reversed(levels):
        if g:
            g.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
