#!/usr/bin/env python3
"""SHMIP suite-A/B time loop on a level cut into rank strips, one process per GPU:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/shmip_dist.py --case A3 --scale 4 --steps 50
Every rank steps its strip with suhmo_level_timestep (halo exchanges and the Picard all-reduce go through the native
RCCL hooks on the "nccl" backend; SUHMO_DIST_BACKEND=gloo rehearses several ranks on one GPU), the SHMIP table is
assembled from the strips' column sums.  --check: rank 0 also runs the whole level alone and compares bit for bit."""
import argparse
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
from suhmo_amd import model, multigpu, synthetic as sy

NAMES = ("head", "B", "mR", "qwx")


def build(case, scale):
    if case.startswith("B"):
        import json
        b = json.load(open(os.path.join(ROOT, "tests", "golden", "shmip_B_inputs.json")))[case]
        m = dict(sy.shmip_b_model(case, b), moulin_position=np.array(b["positions"]).reshape(-1, 2),
                 moulin_sigma=np.array(b["sigma"], dtype=np.float64), moulin_flux=np.array(b["flux"], dtype=np.float64))
    else:
        m = sy.shmip_a_model(case)
    nx, ny = m["nx"] * scale, m["ny"] * scale
    st = sy.shmip_initial_state(nx, ny, m["lx"], m["ly"])
    return m, nx, ny, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="A3")
    ap.add_argument("--scale", type=int, default=1, help="refinement of the reference's 320 x 64 grid")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--halo", type=int, default=4)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
        os.environ.setdefault(k, v)                           # plain `python tools/shmip_dist.py` = one rank
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get("SUHMO_DIST_BACKEND", "nccl"))
    m, nx, ny, st = build(a.case, a.scale)
    assert ny % world == 0 and (ny // world) % 2 == 0, "rows must split evenly over the ranks"
    nyl, j0 = ny // world, rank * (ny // world)
    mb = min(64, nyl)          # the same boxes for the partitioned and the whole level: the MG depth follows the box size
    G = model.HipModel(nx, nyl, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=mb, device=dev,
                       j0=j0, ny_global=ny, halo_rows=a.halo if world > 1 else 1)
    G.set_state({k: (v[j0:j0 + nyl + 2] if isinstance(v, np.ndarray) else v) for k, v in st.items()})
    if world > 1:
        multigpu.attach(G.level, dist, rank, world, periodic_y=False)
        if rank == 0 and getattr(G.level, "_ipc_probe", None):
            print("transport probe: %s -> %s" % (G.level._ipc_probe, getattr(G.level, "_transport", None) or "the transport it had"), flush=True)
    if m.get("use_moulin_source"):
        G.moulin_source(m["moulin_position"], m["moulin_sigma"], m["moulin_flux"], 1.0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = [G.timestep(m["dt"]) for _ in range(a.steps)]
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    sums = torch.from_numpy(G.postproc_partial()).to(torch.device("cuda", dev))
    dist.all_reduce(sums)
    table = G.postproc_finish(sums.cpu().numpy())
    ok = True
    if a.check:
        mine = {k: torch.from_numpy(G.get(k)).to(torch.device("cuda", dev)) for k in NAMES}
        parts = {}
        for k in NAMES:
            lst = [torch.empty_like(mine[k]) for _ in range(world)]
            dist.all_gather(lst, mine[k])
            parts[k] = np.vstack([t.cpu().numpy() for t in lst])
        if rank == 0:
            W = model.HipModel(nx, ny, st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=mb, device=dev)
            W.set_state(st)
            if m.get("use_moulin_source"):
                W.moulin_source(m["moulin_position"], m["moulin_sigma"], m["moulin_flux"], 1.0)
            cw = [W.timestep(m["dt"]) for _ in range(a.steps)]
            ok = cw == counts
            for k in NAMES:
                eq = np.array_equal(W.get(k), parts[k], equal_nan=True)
                print("  %-5s %s" % (k, "bitwise equal" if eq else "DIFFERS (max %.3e)" % np.nanmax(np.abs(W.get(k) - parts[k]))))
                ok = ok and eq
            tw = W.postproc_table_device()
            okf = np.isfinite(tw)
            terr = float(np.max(np.where(okf, np.abs(tw - table), 0.0) / np.max(np.where(okf, np.abs(tw), 0.0), axis=0).clip(1e-300)))
            print("  table max rel diff %.2e" % terr)
            ok = ok and terr < 1e-11
            W.close()
    if rank == 0:
        tot_p, tot_v = sum(c[0] for c in counts), sum(c[1] for c in counts)
        print("SHMIP %s %dx%d on %d rank(s): %d steps in %.2f s (%.1f steps/s), %d Picard iterations, %d V-cycles%s"
              % (a.case, nx, ny, world, a.steps, dt, a.steps / dt, tot_p, tot_v,
                 (" -> " + ("BITWISE EQUAL to the single-process run" if ok else "MISMATCH")) if a.check else ""), flush=True)
    G.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
