#!/usr/bin/env python3
"""bench.py -- head-solve hot path on MI355X.

One "step" = one FAS multigrid V-cycle of the hydraulic-head solve (4 pre + 4 post
nonlinear GSRB sweeps per depth, bottom relaxes, restrict/prolong, on-the-fly bCoef
update) on a 4096 x 4096 single level with 64^2 boxes, SHMIP-A synthetic inputs
(SURVEY.md 8d "K-bench").  N > 1: weak scaling, every rank holds a 4096 x 4096 strip of
a 4096 x (4096 N) level, strips coupled by halo exchange over RCCL.

Prints ONE JSON line (rank 0).  `value` = V-cycles/s of the whole job; the GSRB
cell-update rate and the roofline of the dominant kernel (the GSRB sweep at depth 0,
72 algorithmic bytes per cell per sweep) ride along, plus the CPU baseline (the oracle's
un-fused box-by-box restatement of the reference path, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_CELL_SWEEP = 72.0    # phi r+w, rhs, bx, by, B, Pi, zb, mask
PMC_FILE = "r04_pmc_traffic_gsrb.json"   # HBM bytes per launch of the depth-0 kernel, from this round's rocprofv3 --pmc passes
PMC_VCYCLE_FILE = "r04_pmc_traffic_vcycle.json"   # HBM bytes of ONE whole V-cycle, every kernel (tools/pmc_vcycle.sh + tools/make_traffic_json.py --vcycle)
FLOOR_FILE = "r04_other_configs_floor.json"       # ms of the smaller configurations this build is held to (regression guard)
LX = 1.0e5                     # width of the synthetic domain in metres (SHMIP-A: 100 km)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", dest="n", type=int, default=4096, help="cells per side of one rank strip")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--sweeps-only", type=int, default=0, help="also time this many bare GSRB sweeps")
    ap.add_argument("--no-side", action="store_true", help="skip the side figures of the smaller BASELINE configurations")
    ap.add_argument("--no-guard", action="store_true", help="report a configuration slower than its committed floor without failing")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank holds cells x cells; strong: the level of cells x cells is cut into --gpus row strips")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                     # plain `python bench.py --gpus N`: this process starts the N ranks itself
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE = %d but --gpus %d" % (world, args.gpus))

    import numpy as np
    from suhmo_amd import capi, level, synthetic as sy

    n = args.n
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)          # rehearsal: several ranks may share one GPU (gloo only)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("SUHMO_DIST_BACKEND", "nccl"))
    assert capi.lib().suhmo_device_count() > 0, "no GPU visible: the product path has no CPU fallback"

    strong = args.scaling == "strong"
    if strong:
        assert n % world == 0 and (n // world) % 64 == 0, "strong scaling: --cells must split into 64-row boxes per rank"
    rows = n // world if strong else n                  # rows of this rank's strip
    ny_global = n if strong else n * world
    # square cells (100 km across n columns, as many metres per row): on them the cycle being timed is a contractive solver
    # (SHMIP-A's own 20 km width over n rows would make the cells 5 : 1, where point relaxation stalls: same arithmetic per
    # cycle, but not a solve anybody would run)
    f = sy.shmip_fields(n, rows, ly=LX * ny_global / n, j0=rank * rows, ny_total=ny_global)
    G = level.HipLevel(n, rows, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64, j0=rank * rows,
                       ny_global=ny_global, device=local_rank, halo_rows=int(os.environ.get("SUHMO_HALO_ROWS", "24")) if world > 1 else 1)
    G.set_inputs(f)
    if world > 1:
        from suhmo_amd import multigpu
        multigpu.attach(G, dist, rank, world)
    G.build_mg_coefficients()
    sp = dict(sy.SOLVER_DEFAULT)

    def sync():
        G.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    G.residual(); res_before = G.norm(level.F_RES, 0)     # max norm over all ranks (outside the timed region)
    for _ in range(args.warmup):
        G.vcycle(sp)
    sync()
    G.profile(True)
    msgs0 = G.rccl_exchanges() if world > 1 else 0
    agg0 = G.get_option("agg_gathers") if world > 1 else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        G.vcycle(sp)
    sync()
    t1 = time.perf_counter()
    msgs = (G.rccl_exchanges() - msgs0) / args.steps if world > 1 else 0.0
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    G.residual(); res_after = G.norm(level.F_RES, 0)
    gsrb_ms, gsrb_launches, gsrb_cells = G.profile_read()
    rst_ms, rst_launches, rst_cells = G.profile_read(restricting=True)
    G.profile(False)

    sweeps_depth0 = 2 * sp["num_smooth"]
    cells = n * rows                                    # cells of this rank's strip
    ms_per_step = 1e3 * elapsed / args.steps
    vps = args.steps / elapsed
    # GSRB sweeps at every depth: 8 per depth + bottom, geometric in cells
    ndepth = G.ndepth
    updates_per_vcycle = sum((cells >> (2 * d)) * (sp["num_bottom"] if d == ndepth - 1 else sweeps_depth0)
                             for d in range(ndepth))
    # profile_read: device time over all depth-0 relax launches in the timed region and the
    # cell-updates they performed (a K-sweep fused launch counts K sweeps)
    sweeps_timed = gsrb_cells / cells
    sweep_ms = gsrb_ms / max(sweeps_timed, 1)
    achieved = BYTES_PER_CELL_SWEEP * cells / (sweep_ms * 1e-3) / 1e9 if gsrb_launches else 0.0

    # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc cannot run inside this process:
    # collected with tools/pmc_any.sh on the same workload, corrected as MI355X_MICROARCH.md prescribes, committed)
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", PMC_FILE)
    if gsrb_launches and os.path.exists(pmc) and world == 1:     # the PMC passes were taken on the single-GPU workload only
        pj = json.load(open(pmc))
        if pj.get("cells") == cells and abs(sweeps_timed / gsrb_launches - pj.get("sweeps_per_launch", 0)) < 1e-9:
            traffic, traffic_src = pj["hbm_bytes_per_launch"], "profiles/" + PMC_FILE
    vtraffic, vtraffic_src = None, None
    pmcv = os.path.join(ROOT, "profiles", PMC_VCYCLE_FILE)
    if os.path.exists(pmcv) and world == 1:
        pv = json.load(open(pmcv))
        if pv.get("cells") == cells:
            vtraffic, vtraffic_src = pv["hbm_bytes_per_vcycle"], "profiles/" + PMC_VCYCLE_FILE
    launch_ms = gsrb_ms / max(gsrb_launches, 1)
    alg_bytes_launch = BYTES_PER_CELL_SWEEP * cells * sweeps_timed / max(gsrb_launches, 1)

    extra = {}
    if rst_launches:
        # the launch that ends the pre-smoothing also restricts: 2 sweeps (144 B/cell) + RESTRICTRESVCNL2D/RESTRICTVCNL (76 B/cell)
        rl = rst_ms / rst_launches
        rbytes = (BYTES_PER_CELL_SWEEP * rst_cells / rst_launches) + 76.0 * cells
        extra["gsrb_plus_restrict_launch"] = {"kernel": "k_gsrb_fused<2, false, 64, 1> (2 sweeps + restriction in one pass)", "avg_launch_ms": rl,
                                              "launches_timed": rst_launches, "algorithmic_bytes_per_launch": rbytes,
                                              "algorithmic_GBs": rbytes / (rl * 1e-3) / 1e9,
                                              "note": "algorithmic bytes of 2 sweeps + a separate restriction pass; the fused launch moves far less, "
                                                      "so this is an effective rate, not HBM utilisation"}
    # one iteration of the solve loop the time step runs (AMRFASMultiGrid::solve: V-cycle, residual for the stopping rule, its max norm and
    # the read-back the host decides on) -- outside the timed region, a side figure: the launch that ends the V-cycle leaves the residual behind
    # where the streaming kernel runs depth 0
    if not args.no_side and world == 1:                 # (N > 1: the line carries the timed V-cycles and nothing that could cost it)
        k_it = 10
        sp_it = dict(sp, eps=1e-30, norm_thresh=1e-30, hang=-1.0, max_iter=k_it, imin=k_it, iter_min=k_it)
        sync()
        c0 = G.get_option("residual_in_relax_launches")
        t0 = time.perf_counter()
        G.solve(sp_it)
        sync()
        extra["solve_iteration"] = {"ms": 1e3 * (time.perf_counter() - t0) / k_it, "iterations": k_it,
                                    "what": "V-cycle + residual + max norm + read-back per iteration of suhmo_level_solve",
                                    "residual_left_by_the_last_launch": bool(G.get_option("residual_in_relax_launches") - c0 >= k_it)}
    if not args.no_side and world == 1:
        extra["converged_solve"] = converged_solve(sy, level, G, f, n, rows)
    if args.sweeps_only:
        sync()
        G.profile(True)
        t0 = time.perf_counter()
        G.gsrb(args.sweeps_only)
        sync()
        dt = time.perf_counter() - t0
        ms, nl, nc = G.profile_read()
        G.profile(False)
        extra["bare_gsrb"] = {"sweeps": args.sweeps_only, "wall_ms_per_sweep": 1e3 * dt / args.sweeps_only,
                              "event_ms_per_sweep": ms / max(nc / cells, 1),
                              "cell_updates_per_s": cells * args.sweeps_only / dt}

    # the other configurations of BASELINE.json that fit this run, as side figures (not `value`): configs[1] =
    # SHMIP A3 on 1024^2 single-level (cache-resident: 9 arrays x 8 MB, so it is not an HBM-roofline case),
    # configs[2] = 2-level AMR (64 x 16 base + refined box), and the time step of SHMIP A3 (320 x 64)
    guard = None
    if world == 1 and not args.no_side:
        extra["other_configs"] = side_configs(sy, level, sp, args)
        guard = regression_guard(extra["other_configs"], lambda: side_configs(sy, level, sp, args))
        extra["regression_guard"] = guard

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, phi_cpu, ncyc = cpu_baseline(sy, n if n <= 4096 else 4096, sp)
        if n <= 4096:
            # the level that was just timed, reloaded with the same inputs, against the checker after the same number of cycles:
            # bit for bit at the size the headline is quoted on (outside the timed region)
            G.set_inputs(f); G.build_mg_coefficients()
            for _ in range(ncyc):
                G.vcycle(sp)
            phi_gpu = G.get(level.F_PHI)
            parity = bool(np.array_equal(phi_gpu, phi_cpu))
            if not parity:
                sys.exit("bench.py: the HIP V-cycles differ from the CPU restatement at %dx%d: max |d| = %g" % (n, n, float(np.max(np.abs(phi_gpu - phi_cpu)))))
            del phi_gpu, phi_cpu

    if rank == 0:
        out = {
            "metric": "multigrid V-cycles/s (FAS head solve) + GSRB cell-updates/s; achieved HBM GB/s vs peak",
            # whole-job aggregate: a unit is one V-cycle over one GPU's batch of n x n cells; a step runs one on every GPU
            # (weak scaling: the level is world x as large), so value = world x steps / time
            "value": vps * (1 if strong else world), "unit": "V-cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SHMIP-A head solve, %dx%d cells per GPU (square cells of %.1f m), single AMR level, 64x64 boxes, "
                                   "%d MG depths, 4+4 GSRB sweeps per depth, %d bottom (BASELINE north_star: 4096^2 single-level)"
                                   % (n, rows, LX / n, ndepth, sp["num_bottom"]),
                       "global_cells": [n, ny_global], "partition": "row strips, 1 per GPU" if world > 1 else "none",
                       # N > 1: the halo transport in use and the number of ranks its communicator reports (ncclCommCount), so that N can be verified
                       "transport": transport_name(G) if world > 1 else None,
                       "transport_probe": getattr(G, "_ipc_probe", None) if world > 1 else None,
                       "ranks_in_communicator": int(capi.lib().suhmo_level_rccl_comm_count(G.h)) if world > 1 else None,
                       "unit_of_value": "V-cycles over %dx%d cells (%s; the level of %dx%d cells completes %.4g V-cycles/s)"
                                        % (n, n, "the whole level, cut into strips" if strong else "one per GPU per step", n, ny_global, vps),
                       "halo_message_groups_per_vcycle_per_rank": msgs if world > 1 else None,
                       # rank strips: multigrid depths from this one on run agglomerated on a whole-level copy (0 = none; suhmo_agg.hip),
                       # fed by all-gathers that are counted among the message groups above
                       "agglomerated_from_depth": int(capi.lib().suhmo_level_agglomerated_depth(G.h)) if world > 1 else None,
                       "allgathers_per_vcycle_per_rank": (G.get_option("agg_gathers") - agg0) / args.steps if world > 1 else None},
            "residual_max_norm": {"before_warmup": res_before, "after_timed_cycles": res_after, "cycles": args.warmup + args.steps},
            "gsrb_cell_updates_per_s": updates_per_vcycle * vps * world,   # per-rank strip updates x ranks (strong and weak alike)
            "gsrb_depth0_cell_updates_per_s_kernel": cells / (sweep_ms * 1e-3) * world if gsrb_launches else None,
            "roofline": {"bound": "hbm", "kernel": "GSRB sweep (red+black) at depth 0", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_is": "effective (temporal blocking: K = 2 sweeps per pass over HBM); HBM utilisation = physical_frac",
                         "frac_note": "algorithmic bytes (72 B per cell per sweep) / time / peak, SURVEY 8(d); the kernel does K = 2 sweeps per "
                                      "pass over HBM and skips the ice-mask array when the V-cycle's UpdateOperator found no negative cell, so this "
                                      "is an effective rate that can exceed 1: physical_frac is the HBM utilisation",
                         # HBM bytes the launch really moved (PMC) / time / peak, and the bound of a K-sweep blocked launch (72 B per cell ONCE) / time / peak
                         "physical_frac": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "blocked_bound_frac": (BYTES_PER_CELL_SWEEP * cells / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if gsrb_launches else None,
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                         # the whole cycle: PMC bytes of EVERY kernel of one V-cycle / the measured time of a V-cycle / peak
                         "vcycle_hbm_bytes": vtraffic, "vcycle_hbm_bytes_source": vtraffic_src,
                         "vcycle_physical_frac": (vtraffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if vtraffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": launch_ms,
                         "algorithmic_bytes_per_cell_sweep": BYTES_PER_CELL_SWEEP,
                         "avg_sweep_ms": sweep_ms, "sweeps_timed": sweeps_timed, "launches_timed": gsrb_launches},
            "cpu_baseline": cpu,
            # head after the cpu_baseline's V-cycles: HIP level == CPU restatement, np.array_equal, at this very size (None: no CPU leg)
            "parity_at_bench_size": parity,
        }
        out.update(extra)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if guard is not None and not guard["ok"] and not args.no_guard:
        sys.exit("bench.py: regression guard: %s" % "; ".join(guard["slower_than_floor"]))


def transport_name(G):
    if getattr(G, "_transport", None):
        return G._transport
    ex = getattr(G, "_exchanger", None)
    if ex == "rccl":
        return "rccl (ncclSend / ncclRecv on the kernels' stream)"
    return "torch.distributed P2P (%s)" % type(getattr(ex, "tr", ex)).__name__


def converged_solve(sy, level, G, f, n, rows):
    """Time to solution, not V-cycles per second: suhmo_level_solve from SHMIP-A's initial head to the reference's tolerances of a
    step >= 50 (src/AmrHydro.cpp:737-762: eps 1e-7, hang 0.01, normThresh 1e-7, iterMin 2, <= 100 cycles, 4 + 4 sweeps, 16 bottom) on the
    bench level with the 64^2 boxes of the headline (the cycle bottoms out at 128^2 cells) and with max_box_size = the level (an ordinary
    input of the reference, exec/A_SHMIP/A3/input.hydro:71-72: the cycle goes down to 2^2 cells).  Both bitwise against the oracle at
    1024^2 (tests/test_gpu_parity.py::test_converged_solve_bitwise)."""
    out = {}
    sp = dict(sy.SOLVER_DEFAULT)
    for tag, mb in (("max_box_64", 64), ("max_box_is_the_level", n)):
        # a level of its own for each case (the timed one carries face coefficients of the head it ended with; a fresh operator starts
        # from the state the parity test starts from)
        L = level.HipLevel(n, rows, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=mb)
        L.set_inputs(f); L.build_mg_coefficients()
        L.synchronize()
        t0 = time.perf_counter()
        it, hist = L.solve(sp)
        L.synchronize()
        dt = time.perf_counter() - t0
        out[tag] = {"ms": 1e3 * dt, "vcycles": int(it), "mg_depths": L.ndepth, "residual_first": float(hist[0]), "residual_last": float(hist[-1]),
                    "converged": bool(hist[-1] <= sp["norm_thresh"] or hist[-1] <= sp["eps"] * hist[0])}
        L.close()
    out["tolerances"] = "eps %g, hang %g, normThresh %g, iterMin %d, max %d cycles (src/AmrHydro.cpp:737-762, step >= 50)" % (
        sp["eps"], sp["hang"], sp["norm_thresh"], sp["iter_min"], sp["max_iter"])
    return out


def regression_guard(configs, remeasure):
    """every entry of other_configs against the committed floor (profiles/FLOOR_FILE: ms this build is held to): > 10 % slower fails the run
    (non-zero exit after the JSON line) -- once re-measured, so that one disturbed measurement does not"""
    path = os.path.join(ROOT, "profiles", FLOOR_FILE)
    if not os.path.exists(path):
        return {"ok": True, "floor": None}
    floor = json.load(open(path))["ms"]

    def ms_of(e):
        return e.get("ms_per_vcycle", e.get("ms_per_step"))

    def slow(cfgs):
        return [k for k, v in floor.items() if k in cfgs and ms_of(cfgs[k]) > 1.10 * v]
    bad = slow(configs)
    retried = False
    if bad:
        retried = True
        again = remeasure()
        for k in bad:
            if k in again and ms_of(again[k]) < ms_of(configs[k]):
                configs[k] = again[k]
        bad = slow(configs)
    return {"ok": not bad, "floor": "profiles/" + FLOOR_FILE, "threshold": "1.10 x floor ms", "remeasured": retried,
            "slower_than_floor": ["%s: %.3f ms against a floor of %.3f" % (k, ms_of(configs[k]), floor[k]) for k in bad]}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here (fresh children; this parent never
    touches the GPU and never execs), one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is the output."""
    import socket
    import subprocess
    import random
    port = None
    for _ in range(200):                    # outside the ephemeral range: a port found by bind(0) can be taken by an outgoing connection before the store listens
        cand = random.randint(20000, 29999)
        with socket.socket() as so:
            try:
                so.bind(("127.0.0.1", cand))
            except OSError:
                continue
        port = cand
        break
    if port is None:
        sys.exit("bench.py: no free rendezvous port")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if os.environ.get("SUHMO_DIST_BACKEND") == "gloo":
            env.setdefault("GLOO_SOCKET_IFNAME", "lo")      # single-node rehearsal: the hostname may not resolve
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def side_configs(sy, level, sp, args):
    out = {}
    # configs[1]: SHMIP A3, 1024^2 single level
    n = 1024
    f = sy.shmip_fields(n, n, ly=LX)
    G = level.HipLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64)
    G.set_inputs(f); G.build_mg_coefficients()
    for _ in range(4):
        G.vcycle(sp)
    G.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        G.vcycle(sp)
    G.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["shmip_a3_1024x1024_single_level"] = {"vcycles_per_s": 1.0 / dt, "ms_per_vcycle": 1e3 * dt, "mg_depths": G.ndepth}
    G.close()
    # configs[2]: exec/0_convergence_channelized 2lev_base, base 64 x 16 + refined box (and the same shape x 16)
    for tag, nx0, ny0, patch, kw in (("amr2_cfg3_64x16", 64, 16, sy.CFG3_PATCH, {}),
                                     ("amr2_1024x256_base", 1024, 256, (256, 64, 767, 191), dict(lx=1024.0, ly=256.0))):
        c, fi = sy.amr2_fields(nx0, ny0, patch, **kw)
        bc = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 1])
        A = level.HipAmr2(nx0, ny0, c["dx"], c["dy"], bc, sy.CFG3_PHYS, patch, max_box=32)
        A.coarse.set_inputs(c); A.coarse.build_mg_coefficients(); A.fine.set_inputs(fi)
        for _ in range(4):                       # eager, then one captured graph per ping-pong state, then replays
            A.vcycle(sp)
        A.coarse.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            A.vcycle(sp)
        A.coarse.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out[tag] = {"amr_vcycles_per_s": 1.0 / dt, "ms_per_vcycle": 1e3 * dt,
                    "fine_cells": int(fi["nx"] * fi["ny"]), "base_cells": nx0 * ny0}
        A.close()
    # the caller of the solve: SHMIP A3 time steps (320 x 64, dt = 1 h), 200 steps
    from suhmo_amd import model
    m = sy.A3_MODEL
    st = sy.shmip_initial_state(m["nx"], m["ny"], m["lx"], m["ly"])
    M = model.HipModel(m["nx"], m["ny"], st["dx"], st["dy"], sy.A3_BC, sy.A3_PHYS, m, max_box=64)
    M.set_state(st)
    for _ in range(60):
        M.timestep(m["dt"])
    t0 = time.perf_counter()
    nv = 0
    for _ in range(200):
        nv += M.timestep(m["dt"])[1]
    dt = (time.perf_counter() - t0) / 200
    out["shmip_a3_320x64_timestep"] = {"steps_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "vcycles_per_step": nv / 200.0}
    M.close()
    # cfg5 shape: transient head + gap height on a 3-level hierarchy with moulins (base 1024 x 256 over 100 km x 20 km,
    # two nested patches refined by 2 each), 20 steps after 10
    nx0, ny0, patches = 1024, 256, ((256, 64, 767, 191), (768, 192, 1279, 319))
    ma = dict(sy.A3_MODEL, use_moulin_source=1, distributed_input=7.93e-11)
    sts = sy.shmip_amr_states(nx0, ny0, patches)
    A = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, ma, patches, max_box=64)
    for l, st_ in enumerate(sts):
        A.set_state(l, st_)
    import numpy as np
    rng = np.random.default_rng(7)
    pos = np.stack([rng.uniform(3.0e4, 7.0e4, 63), rng.uniform(6.0e3, 1.4e4, 63)], axis=1)      # 63 moulins as exec/AMR_multiMoulins
    A.moulin_source(pos, np.full(63, 200.0), np.full(63, 90.0 / 63), 1.0)
    for _ in range(10):
        A.timestep(ma["dt"])
    A.levels[0].synchronize()
    t0 = time.perf_counter()
    nv = 0
    for _ in range(20):
        nv += A.timestep(ma["dt"])[1]
    A.levels[0].synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["amr3_timestep_1024x256_base_63_moulins"] = {"steps_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "amr_vcycles_per_step": nv / 20.0,
                                                     "cells_per_level": [int(st_["nx"] * st_["ny"]) for st_ in sts]}
    A.close()
    # cfg5 as the reference grids it (exec/AMR_multiMoulins/run_C_3lev: 63 moulins on 100 km x 100 km, dino bed without its unseeded
    # noise, diffusion + implicit gap-height solve): base + 3 AMR levels, every level a union of boxes around the moulins
    for tag, nb in (("cfg5_multimoulins_256_base_3_amr_levels", 256), ("cfg5_multimoulins_4096_base_3_amr_levels", 4096)):
        if nb > 256 and args.n < 4096:
            continue
        bc, ph, mm, mo = sy.multimoulins_setup()
        boxes = sy.boxes_around(mo["positions"], nb, nb, 4, 1.0e5, 1.0e5)
        sts = sy.mountain_amrm_states(nb, nb, boxes)
        H = model.HipHierModel(nb, nb, sts[0][0]["dx"], sts[0][0]["dy"], bc, ph, mm, boxes, max_box=64)
        H.set_states(sts)
        H.moulin_source(**mo)
        nwarm, nstep = (3, 5) if nb > 256 else (5, 10)
        for _ in range(nwarm):
            H.timestep(mm["dt"])
        H.level[0][0].synchronize()
        t0 = time.perf_counter()
        nv = npi = 0
        for _ in range(nstep):
            a, b = H.timestep(mm["dt"])
            npi += a; nv += b
        H.level[0][0].synchronize()
        dt = (time.perf_counter() - t0) / nstep
        out[tag] = {"steps_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "picard_iterations_per_step": npi / nstep, "amr_vcycles_per_step": nv / nstep,
                    "boxes_per_level": [1] + [len(bl) for bl in boxes],
                    "cells_per_level": [nb * nb] + [int(sum((b[2] - b[0] + 1) * (b[3] - b[1] + 1) for b in bl)) for bl in boxes]}
        H.close()
    return out


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the
    GPU box exposes all host CPUs in the mask but grants a 16-core share per GPU)."""
    n = len(os.sched_getaffinity(0))
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(int(q[0]) / int(q[1]))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return min(n, int(os.environ.get("SUHMO_CPU_CORES", "16")))


def cpu_baseline(sy, n, sp):
    """The oracle's restatement of the reference CPU path (un-fused levelGSRB etc. over 64^2
    boxes, OpenMP over boxes), same workload, bounded sample: 1 warm-up + 2 timed V-cycles."""
    from oracle import pyoracle as po
    cores = host_cores()
    f = sy.shmip_fields(n, n, ly=LX)
    O = po.OracleLevel(n, n, f["dx"], f["dy"], sy.A3_BC, sy.A3_PHYS, max_box=64, nthreads=cores)
    O.set_inputs(f)
    O.build_mg_coefficients()
    O.vcycle(sp)
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        O.vcycle(sp)
    dt = time.perf_counter() - t0
    phi = O.get(po.F_PHI)
    t0 = time.perf_counter()
    O.gsrb(2)
    dts = time.perf_counter() - t0
    O.close()
    return {"value": reps / dt, "unit": "V-cycles/s", "cores": cores, "kind": "port",
            "sample": "%d V-cycles of the same %dx%d workload (after 1 warm-up), CPU restatement of the "
                      "reference path (oracle level shim, 64^2 boxes, OpenMP over boxes)" % (reps, n, n),
            "gsrb_cell_updates_per_s": 2 * n * n / dts}, phi, 1 + reps


if __name__ == "__main__":
    main()
