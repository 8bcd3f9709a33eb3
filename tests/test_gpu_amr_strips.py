"""AMR hierarchy cut into rank strips (every rank holds, of every level, the rows of its own physical slab): thread-ranks
on one GPU drive suhmo_amr_vcycle / suhmo_amr_solve through the same hooks the multi-GPU run uses, and the gathered
result must equal the single-process hierarchy BIT FOR BIT.  The barrier-based test transport also proves that every
rank of a level's communicator issues the same sequence of exchanges (a mismatch breaks the barrier)."""
import ctypes as C
import threading

import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
BC = dict(type=[[0, 0], [1, 0]], value=[[0.0, 0.0], [0.0, 0.0]], periodic=[0, 0])


def strip_of(f, j0, ny):
    """rows [j0, j0 + ny) of a level's input dict (patch-local row offsets)"""
    s = dict(nx=f["nx"], ny=ny, dx=f["dx"], dy=f["dy"])
    for k in ("phi", "rhs", "aCoef"):
        s[k] = f[k][j0:j0 + ny]
    for k in ("B", "Pi", "zb", "mask"):
        s[k] = f[k][j0:j0 + ny + 2]
    return s


def run_amr_strips(world, nx0, ny0, patches, bc, ph, fs, sp, ncycles, body=None):
    from suhmo_amd import capi, level, multigpu
    nlev = 1 + len(patches)
    n0 = ny0 // world
    # level l: global row range of the patch in level-l rows, and the ranks it reaches
    rng = [(0, ny0)]
    for l, (ci0, cj0, ci1, cj1) in enumerate(patches, start=1):
        rng.append((2 * cj0, 2 * cj1 + 2))
    own = []       # own[r][l] = (j0, ny) or None
    for r in range(world):
        o = []
        for l in range(nlev):
            lo, hi = max(rng[l][0], r * n0 * 2 ** l), min(rng[l][1], (r + 1) * n0 * 2 ** l)
            o.append((lo, hi - lo) if hi > lo else None)
        own.append(o)
    part = [[r for r in range(world) if own[r][l]] for l in range(nlev)]
    trs = [multigpu.ThreadTransport(len(part[l])) for l in range(nlev)]
    out, err = [None] * world, []

    def worker(rank):
        try:
            lv, keep = [], []
            nxg, nyg, dx, dy = nx0, ny0, fs[0]["dx"], fs[0]["dy"]
            for l in range(nlev):
                if l > 0:
                    nxg, nyg, dx, dy = 2 * nxg, 2 * nyg, dx / 2.0, dy / 2.0
                if not own[rank][l]:
                    lv.append(None)
                    continue
                j0, ny = own[rank][l]
                if l == 0:
                    G = level.HipLevel(nx0, ny, dx, dy, bc, ph, 0.0, -1.0, MB, j0=j0, ny_global=ny0, halo_rows=4)
                    G.set_inputs(strip_of(fs[0], j0, ny))
                else:
                    ci0, cj0, ci1, cj1 = patches[l - 1]
                    G = level.HipLevel(2 * (ci1 - ci0 + 1), ny, dx, dy, bc, ph, 0.0, -1.0, MB, j0=j0, ny_global=nyg, i0=2 * ci0,
                                       nx_global=nxg, halo_rows=2, patch_j0=rng[l][0], patch_ny=rng[l][1] - rng[l][0])
                    G.set_inputs(strip_of(fs[l], j0 - rng[l][0], ny))
                sub = part[l].index(rank)
                ex = multigpu.StripExchanger(G, trs[l], sub, len(part[l]), False)
                ex.exchange_static()
                keep.append(ex)
                lv.append(G)
            lv[0].build_mg_coefficients()
            arr = (C.c_void_p * nlev)(*[(g.h if g else None) for g in lv])
            s = level.solver_params(sp)
            if body is not None:
                out[rank] = body(lv, arr, nlev, rank)
                for g in lv:
                    if g:
                        g.synchronize()
                return
            r0 = C.c_double()
            capi.check(capi.lib().suhmo_amr_residual(arr, nlev, C.cast(C.pointer(r0), C.POINTER(C.c_double)), None))
            for _ in range(ncycles):
                capi.check(capi.lib().suhmo_amr_vcycle(arr, nlev, C.byref(s), None))
            r1 = C.c_double()
            capi.check(capi.lib().suhmo_amr_residual(arr, nlev, C.cast(C.pointer(r1), C.POINTER(C.c_double)), None))
            out[rank] = ([(g.get(level.F_PHI) if g else None) for g in lv], r0.value, r1.value)
            for g in lv:
                if g:
                    g.synchronize()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)
            for t in trs:
                t.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    return out, own


# 16 x 16 boxes on every rank count: the box set (hence the multigrid depth, MGnewOp) is the same for 1, 2 and 4 ranks
MB = 16
CASES = [
    ("2-levels-2-ranks", 2, 64, 64, ((8, 12, 39, 51),)),                      # the patch spans both slabs
    ("2-levels-4-ranks", 4, 64, 64, ((8, 12, 39, 51),)),                      # ... all four
    ("3-levels-2-ranks", 2, 64, 64, ((8, 12, 39, 51), (24, 40, 59, 87))),     # level 2 spans both slabs too
    ("2-levels-4-ranks-partial", 4, 64, 64, ((8, 20, 39, 43),)),              # the patch reaches only ranks 1 and 2
]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_amr_strips_bitwise(case):
    from suhmo_amd import level
    _, world, nx0, ny0, patches = case
    ph = sy.CFG3_PHYS
    fs = sy.amr_fields(nx0, ny0, patches, lx=64.0, ly=64.0, moulin=(24.0, 32.0, 2.0, 30.0))
    sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
    # single process
    G = level.HipAmr(nx0, ny0, fs[0]["dx"], fs[0]["dy"], BC, ph, patches, max_box=MB)
    G.levels[0].set_inputs(fs[0]); G.levels[0].build_mg_coefficients()
    for l in range(1, len(fs)):
        G.levels[l].set_inputs(fs[l])
    r0 = G.residual()
    for _ in range(2):
        G.vcycle(sp)
    r1 = G.residual()
    ref = [lv.get(level.F_PHI) for lv in G.levels]
    G.close()
    out, own = run_amr_strips(world, nx0, ny0, patches, BC, ph, fs, sp, 2)
    for r in range(world):
        assert out[r][1] == r0 and out[r][2] == r1, (r, out[r][1], r0, out[r][2], r1)
    for l in range(len(fs)):
        rows = [out[r][0][l] for r in range(world) if own[r][l]]
        got = np.vstack(rows)
        assert np.array_equal(got, ref[l]), (l, float(np.max(np.abs(got - ref[l]))))
