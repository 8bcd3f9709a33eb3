#!/usr/bin/env python3
"""exec/1_convergence_distributed of the reference: the sqrt ice sheet on 80 km x 20 km (y-periodic, diffFactor 1, implicit
gap-height solve, dt = 2 h), 5000 steps at 64x16 ... 1024x256, then the L2 self-convergence errors between successive
resolutions the reference commits in exec/1_convergence_distributed/CONV_ANA/results/convergence_data.dat (ChomboCompare:
error = computed - average(finer), L2 = sqrt(sum e^2 dx^2)).  usage: convergence_distributed.py [hip|oracle] [nsteps] [max_nx]
                                                                       [--mask-gradients 0|1]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suhmo_amd import synthetic as sy

REF = {64: (6466.2, 3.4136, 63368000.0, 280930.0), 128: (1618.1, 0.85712, 15858000.0, 76185.0),
       256: (404.65, 0.21451, 3965600.0, 19408.0), 512: (101.17, 0.053646, 991460.0, 4874.4)}
MODEL = dict(sy.A3_MODEL, G=0.0, ct=7.5e-8, diffFactor=1.0, use_impl_diff=1, distributed_input=5.79e-9, dt=7200.0)
PHYS = dict(sy.A3_PHYS, A=2.5e-25)


def run(which, nx, ny, nsteps, phys):
    st = sy.shmip_initial_state(nx, ny, 8.0e4, 2.0e4)
    for k in ("head", "B", "Pi", "zb", "mask"):                 # y-periodic: ghost rows are the periodic images
        st[k][0, :], st[k][-1, :] = st[k][-2, :].copy(), st[k][1, :].copy()
    mb = min(64, ny)
    if which == "hip":
        from suhmo_amd import model
        M = model.HipModel(nx, ny, st["dx"], st["dy"], sy.CONV_BC, phys, MODEL, max_box=mb)
        M.set_state(st)
        for k in range(nsteps):
            M.timestep(MODEL["dt"])
        out = dict(head=M.get("head"), B=M.get("B"), Pw=M.get("Pw"), Re=M.get("Re"))
        M.close()
    else:
        from oracle import pyoracle as po
        M = po.OracleModel(nx, ny, st["dx"], st["dy"], sy.CONV_BC, phys, MODEL, max_box=mb, nthreads=min(8, os.cpu_count() or 1))
        M.set_state(st)
        for k in range(nsteps):
            M.timestep(MODEL["dt"])
        v = lambda fid: np.array(M.field(fid))[1:-1, 1:-1]
        out = dict(head=v(po.OM_H), B=v(po.OM_B), Pw=v(po.OM_PW), Re=v(po.OM_RE))
        M.close()
    return out, st["dx"]


def table(which, nsteps=5000, max_nx=1024, phys=None, log=None):
    """{nx: (head, gapHeight, Pw, Re) L2 errors vs the next finer run} for nx = 64 ... max_nx / 2"""
    phys = PHYS if phys is None else phys
    sol, dxs = {}, {}
    nx = 64
    while nx <= max_nx:
        t0 = time.time()
        sol[nx], dxs[nx] = run(which, nx, nx // 4, nsteps, phys)
        if log:
            log("%dx%d: %d steps in %.1f s" % (nx, nx // 4, nsteps, time.time() - t0))
        nx *= 2
    res = {}
    for nx in sorted(sol):
        if 2 * nx not in sol:
            continue
        row = []
        for k in ("head", "B", "Pw", "Re"):
            f = sol[2 * nx][k]
            avg = 0.25 * (f[0::2, 0::2] + f[0::2, 1::2] + f[1::2, 0::2] + f[1::2, 1::2])     # ChomboCompare: conservative average of the finer run
            e = sol[nx][k] - avg
            row.append(float(np.sqrt(np.sum(e * e) * dxs[nx] * dxs[nx])))                    # computeNorm p = 2: sqrt(sum e^2 dx^D)
        res[nx] = tuple(row)
    return res


def main():
    args = [a for a in sys.argv[1:]]
    mg = 0
    if "--mask-gradients" in args:
        k = args.index("--mask-gradients"); mg = int(args[k + 1]); del args[k:k + 2]
    which = args[0] if args else "hip"
    nsteps = int(args[1]) if len(args) > 1 else 5000
    max_nx = int(args[2]) if len(args) > 2 else 1024
    phys = dict(PHYS, use_mask_gradients=mg)
    res = table(which, nsteps, max_nx, phys, log=lambda m: print(m, flush=True))
    print("#case     head    gapHeight  Pw          Re        (reference: exec/1_convergence_distributed/CONV_ANA/results/convergence_data.dat)")
    for nx, row in sorted(res.items()):
        ref = REF.get(nx)
        print("%-6d %s   ref %s   ratio %s" % (nx, " ".join("%.5g" % v for v in row), ref, [round(a / b, 4) for a, b in zip(row, ref)] if ref else None), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({str(k): v for k, v in res.items()}, open(os.path.join(ROOT, "gpurun_out", "convergence_distributed_%s.json" % which), "w"))


if __name__ == "__main__":
    main()
