#!/usr/bin/env python3
"""cfg5 shape on several GPUs: the time step (moulins, diffusion, implicit gap-height solve) on a hierarchy whose levels are unions of
boxes, level 0 cut into rank strips, one process per GPU:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/hier_dist.py --base 256 --steps 5
Every rank holds its rows of level 0 and all boxes of the finer levels (or, from --partition-min-cells on, the boxes it owns and mirrors of their neighbours) (suhmo_amd.multigpu.attach_hier: halo rows and the all-gather
of the coarse cells level 1 reads, native RCCL on the "nccl" backend; SUHMO_DIST_BACKEND=gloo rehearses several ranks on one GPU).
--check: rank 0 also runs the whole hierarchy alone and compares bit for bit."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SUHMO_DUMP_AFTER"):          # tests: a rank still running after that many seconds prints where every thread waits
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ["SUHMO_DUMP_AFTER"]), exit=False)
import numpy as np
import torch
import torch.distributed as dist
from suhmo_amd import model, multigpu, synthetic as sy

NAMES = ("head", "B", "mR")


t_start = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=256, help="cells per side of level 0")
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--partition-min-cells", type=int, default=0,
                    help="> 0: hierarchy option partition_min_cells (levels of boxes with at least that many cells per rank are relaxed by their owners; "
                         "default: the library's 350000 for the largest level, i.e. cfg5's small levels stay replicated)")
    ap.add_argument("--dense", action="store_true",
                    help="level 1 = the middle half of the domain tiled with 64^2 boxes (a LARGE level of boxes: what the owner-computes partition is for), "
                         "the finer levels around the moulins inside it")
    a = ap.parse_args()
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29536")):
        os.environ.setdefault(k, v)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get("SUHMO_DIST_BACKEND", "nccl"))
    nb, MB = a.base, 64
    assert nb % world == 0 and (nb // world) % MB == 0, "rows must split into whole boxes per rank"
    n0 = nb // world
    bc, ph, mm, mo = sy.multimoulins_setup()
    boxes = sy.boxes_around(mo["positions"], nb, nb, a.levels, 1.0e5, 1.0e5)
    if a.dense:                                     # level-1 cells [nb/2, 3 nb/2)^2 in boxes of 64^2; deeper levels: the moulins' boxes that lie properly inside
        lo, hi = nb // 2, 3 * nb // 2
        dense = [(i, j, i + 63, j + 63) for j in range(lo, hi, 64) for i in range(lo, hi, 64)]
        keep = [dense]
        for l, bl in enumerate(boxes[1:], start=2):
            f = 1 << (l - 1)
            keep.append([b for b in bl if b[0] >= (lo + 4) * f and b[1] >= (lo + 4) * f and b[2] < (hi - 4) * f and b[3] < (hi - 4) * f])
            if not keep[-1]:
                keep.pop()
                break
        boxes = keep
    sts = sy.mountain_amrm_states(nb, nb, boxes)
    dx, dy = sts[0][0]["dx"], sts[0][0]["dy"]
    H = model.HipHierModel(nb, n0, dx, dy, bc, ph, mm, boxes, max_box=MB, device=dev, j0=rank * n0, ny_global=nb, halo_rows=4 if world > 1 else 1,
                           options=("partition_min_cells=%d" % a.partition_min_cells) if a.partition_min_cells > 0 else None)
    s0 = {k: (v[rank * n0:rank * n0 + n0 + 2] if isinstance(v, np.ndarray) else v) for k, v in sts[0][0].items()}
    H.set_state(0, 0, s0)
    for l in range(1, len(sts)):
        for k, st in enumerate(sts[l]):
            H.set_state(l, k, st)
    if world > 1:
        multigpu.attach_hier(H.hier, dist, rank, world)
    if os.environ.get("SUHMO_DUMP_AFTER"):          # tests: how far has this rank come (twice, 20 s apart: hung or slow?)
        import threading

        def progress():
            for k in range(2):
                time.sleep(max(1, int(os.environ["SUHMO_DUMP_AFTER"]) - 30 + 20 * k))
                print("rank %d after %.0f s: %d all-gathers of coarse cells, %d of owners' boxes, %d halo message groups, %d agglomeration gathers"
                      % (rank, time.perf_counter() - t_start, H.hier.gathers(), H.hier.get_option("partition_gathers"),
                         H.level[0][0].rccl_exchanges() if hasattr(H.level[0][0], "rccl_exchanges") else -1, H.level[0][0].get_option("agg_gathers")), flush=True)
        threading.Thread(target=progress, daemon=True).start()
    integ = H.moulin_source(**mo)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = [H.timestep(mm["dt"]) for _ in range(a.steps)]
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    owned = [H.hier.get_option("own_boxes_level_%d" % l) for l in range(1, len(sts))]
    parted = [H.hier.get_option("partitioned_level_%d" % l) for l in range(1, len(sts))]
    owned_all = [None] * world
    dist.all_gather_object(owned_all, owned)
    # owner computes: what a rank sends per colour-pass ghost exchange against 4 sides x 8 B of its boxes, the canvases it keeps against the level's
    stats = [{k: H.hier.get_option("%s_level_%d" % (k, l)) for k in ("ghost_exchange_bytes", "ghost_exchange_bound_bytes", "held_boxes", "owned_cells", "canvas_bytes")}
             for l in range(1, len(sts))]
    stats_all = [None] * world
    dist.all_gather_object(stats_all, stats)
    if rank == 0 and any(parted):
        for l in range(1, len(sts)):
            tot = sum(q[l - 1]["canvas_bytes"] for q in stats_all)
            print("  level %d dealt to the ranks: per rank, bytes sent per colour-pass ghost exchange %s (bound 4 x side x 8 B of the owned boxes: %s); boxes held %s of %d; "
                  "canvas bytes %s (shares %s of what all ranks hold)"
                  % (l, [q[l - 1]["ghost_exchange_bytes"] for q in stats_all], [q[l - 1]["ghost_exchange_bound_bytes"] for q in stats_all],
                     [q[l - 1]["held_boxes"] for q in stats_all], len(sts[l]), [q[l - 1]["canvas_bytes"] for q in stats_all],
                     ["%.2f" % (q[l - 1]["canvas_bytes"] / max(tot, 1)) for q in stats_all]), flush=True)
    if a.check:
        mine = [[{nm: H.get(l, k, nm) for nm in NAMES} for k in range(len(H.level[l]))] for l in range(len(sts))]
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        if rank == 0:
            A = model.HipHierModel(nb, nb, dx, dy, bc, ph, mm, boxes, max_box=MB, device=dev)
            A.set_states(sts)
            iref = A.moulin_source(**mo)
            cw = [A.timestep(mm["dt"]) for _ in range(a.steps)]
            ok = cw == counts and np.array_equal(iref, integ)
            for nm in NAMES:
                eq = np.array_equal(np.vstack([allv[r][0][0][nm] for r in range(world)]), A.get(0, 0, nm), equal_nan=True)
                ok = ok and eq
                print("  level 0 %-4s %s" % (nm, "bitwise equal" if eq else "DIFFERS"), flush=True)
            for l in range(1, len(sts)):
                # a level dealt to the ranks: the owner of a box answers for it (the other ranks hold no storage for it, or a mirror)
                eq = all(any(allv[r][l][k][nm] is not None for r in range(world)) and
                         all(np.array_equal(allv[r][l][k][nm], A.get(l, k, nm), equal_nan=True) for r in range(world) if allv[r][l][k][nm] is not None)
                         for k in range(len(sts[l])) for nm in NAMES)
                ok = ok and eq
                print("  level %d (%d boxes, %s) %s" % (l, len(sts[l]), "boxes relaxed per rank %s" % [o[l - 1] for o in owned_all] if parted[l - 1] else "relaxed on every rank",
                                                      "bitwise equal" if eq else "DIFFERS"), flush=True)
            A.close()
    if rank == 0:
        print("cfg5 physics on base %d^2 + %d levels of boxes %s, %d rank(s): %d steps in %.2f s (%.2f steps/s), %d Picard iterations, %d AMR V-cycles, %d all-gathers of coarse cells + %d of owners' boxes%s"
              % (nb, len(boxes), [len(b) for b in boxes], world, a.steps, dt, a.steps / dt, sum(c[0] for c in counts), sum(c[1] for c in counts), H.hier.gathers(),
                 H.hier.get_option("partition_gathers"),
                 (" -> " + ("BITWISE EQUAL to the single-process hierarchy" if ok else "MISMATCH")) if a.check else ""), flush=True)
    H.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
