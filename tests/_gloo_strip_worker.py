"""Worker for tests/test_strips_gloo_cpu.py: one rank of a world_size-N gloo job on CPU.
Exercises suhmo_amd.multigpu.TorchDistTransport (the transport bench.py uses with the nccl
backend) and the strip algorithm (exchange before every colour pass, global colour parity)
on a numpy stand-in of the relaxation."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suhmo_amd import multigpu, synthetic as sy  # noqa: E402
from tests import npref  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    periodic = int(sys.argv[1])
    out = sys.argv[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = multigpu.TorchDistTransport(dist, torch.device("cpu"))
    lo = rank - 1 if rank > 0 else (world - 1 if periodic else None)
    hi = rank + 1 if rank < world - 1 else (0 if periodic else None)

    # 1. transport: neighbours' edge rows land on the right side, also when lo == hi
    n = 7
    slo, shi = torch.full((n,), 10.0 * rank + 1, dtype=torch.float64), torch.full((n,), 10.0 * rank + 2, dtype=torch.float64)
    rlo, rhi = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    tr.sendrecv(rank, lo, hi, slo, shi, rlo, rhi, "t1")
    if lo is not None:
        assert torch.all(rlo == 10.0 * lo + 2), (rank, rlo)      # lo neighbour's TOP rows
    if hi is not None:
        assert torch.all(rhi == 10.0 * hi + 1), (rank, rhi)      # hi neighbour's BOTTOM rows
    assert tr.allreduce_max(rank, float(rank)) == world - 1

    # 2. strip-partitioned GSRB == global GSRB, bit for bit
    nx, ny_tot = 24, 8 * world
    f = sy.random_fields(nx, ny_tot, seed=5)
    bc = sy.CONV_BC if periodic else sy.RANDOM_BC
    ph, alpha, beta = sy.RANDOM_PHYS, 0.3, -1.0
    ny = ny_tot // world
    j0 = rank * ny
    phi = f["phi"][j0:j0 + ny].copy()
    sl, sg = slice(j0, j0 + ny), slice(j0, j0 + ny + 2)
    lm = npref.lam(f["aCoef"][sl], f["bx"][sl], f["by"][j0:j0 + ny + 1], alpha, beta, f["dx"], f["dy"])
    jj, ii = np.meshgrid(np.arange(j0, j0 + ny), np.arange(nx), indexing="ij")
    for sweep in range(3):
        for p in range(2):
            pg = npref.fill_ghosts(phi, dict(bc, periodic=[bc["periodic"][0], 0]), f["dx"], f["dy"])
            # rank boundaries: ghost rows come from the neighbours (exchange before each pass)
            rlo, rhi = torch.zeros(nx, dtype=torch.float64), torch.zeros(nx, dtype=torch.float64)
            tr.sendrecv(rank, lo, hi, torch.from_numpy(phi[0].copy()), torch.from_numpy(phi[-1].copy()), rlo, rhi, None)
            if lo is not None:
                pg[0, 1:-1] = rlo.numpy()
            if hi is not None:
                pg[-1, 1:-1] = rhi.numpy()
            v = slice(1, -1)
            nl, dnl = npref.nl_terms(phi, f["B"][sg][v, v], f["Pi"][sg][v, v], f["zb"][sg][v, v], f["mask"][sg][v, v], ph)
            L = npref.op(pg, f["aCoef"][sl], f["bx"][sl], f["by"][j0:j0 + ny + 1], nl, alpha, beta, f["dx"], f["dy"])
            new = phi + (f["rhs"][sl] - L) / (1.0e-16 + lm + dnl)
            phi = np.where(((ii + jj + p) % 2) == 0, new, phi)
    gathered = [torch.zeros(ny, nx, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(phi))
    if rank == 0:
        ref = f["phi"]
        for _ in range(3):
            ref = npref.gsrb_sweep(ref, f["rhs"], f["aCoef"], f["bx"], f["by"], f["B"], f["Pi"], f["zb"], f["mask"],
                                   ph, bc, alpha, beta, f["dx"], f["dy"])
        got = np.vstack([g.numpy() for g in gathered])
        assert np.array_equal(got, ref), "strip result differs from the global sweep"
        open(out, "w").write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
