"""The time step on an AMR hierarchy cut into rank strips (every rank holds, of every level, the rows of its own slab;
a patch may reach only some ranks): thread-"ranks" on one GPU drive suhmo_amr_timestep through the hooks the multi-GPU
run uses; head, gap height, melt rate and the iteration counts of every level must equal the single-process hierarchy
BIT FOR BIT, with the explicit and the implicit gap-height update."""
import ctypes as C
import threading

import numpy as np
import pytest

from suhmo_amd import synthetic as sy

pytestmark = pytest.mark.gpu
MB = 16
NAMES = ("head", "B", "mR", "rhs_h", "qwx")


def strip_state(st, r0, ny):
    """rows [r0, r0 + ny) of a level's state (row offsets local to the level's rectangle), ghosted"""
    return {k: (v[r0:r0 + ny + 2] if isinstance(v, np.ndarray) else v) for k, v in st.items()}


def run_strips(world, nx0, ny0, patches, sts, m, nsteps, mou=None):
    from suhmo_amd import capi, level as lv, model, multigpu
    nlev = 1 + len(patches)
    n0 = ny0 // world
    rng = [(0, ny0)] + [(2 * p[1], 2 * p[3] + 2) for p in patches]
    own = [[None] * nlev for _ in range(world)]
    for r in range(world):
        for l in range(nlev):
            lo, hi = max(rng[l][0], r * n0 * 2 ** l), min(rng[l][1], (r + 1) * n0 * 2 ** l)
            own[r][l] = (lo, hi - lo) if hi > lo else None
    part = [[r for r in range(world) if own[r][l]] for l in range(nlev)]
    trs = [multigpu.ThreadTransport(len(part[l])) for l in range(nlev)]
    mp = model.model_params(m)
    out, err = [None] * world, []

    def worker(rank):
        try:
            levels, keep = [], []
            nxg, nyg, dx, dy = nx0, ny0, sts[0]["dx"], sts[0]["dy"]
            for l in range(nlev):
                if l > 0:
                    nxg, nyg, dx, dy = 2 * nxg, 2 * nyg, dx / 2.0, dy / 2.0
                if not own[rank][l]:
                    levels.append(None)
                    continue
                j0, ny = own[rank][l]
                if l == 0:
                    G = lv.HipLevel(nx0, ny, dx, dy, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=ny0, halo_rows=4)
                else:
                    ci0, cj0, ci1, cj1 = patches[l - 1]
                    G = lv.HipLevel(2 * (ci1 - ci0 + 1), ny, dx, dy, sy.A3_BC, sy.A3_PHYS, 0.0, -1.0, MB, j0=j0, ny_global=nyg, i0=2 * ci0,
                                    nx_global=nxg, halo_rows=2, patch_j0=rng[l][0], patch_ny=rng[l][1] - rng[l][0])
                f = strip_state(sts[l], j0 - rng[l][0], ny)
                G.set(lv.F_PHI, f["head"][1:-1, 1:-1])
                G.set(lv.F_ACOEF, np.zeros((ny, G.nx)))
                for k, fid in (("B", lv.F_B), ("Pi", lv.F_PI), ("zb", lv.F_ZB), ("mask", lv.F_MASK)):
                    G.set(fid, f[k], ghosted=True)
                ex = multigpu.StripExchanger(G, trs[l], part[l].index(rank), len(part[l]), False)
                ex.exchange_static()
                keep.append(ex)
                levels.append(G)
            arr = (C.c_void_p * nlev)(*[(g.h if g else None) for g in levels])
            integ = None
            if mou is not None:          # every rank integrates the whole hierarchy itself from the patch boxes
                pos, sg, fl = [np.ascontiguousarray(a, dtype=np.float64) for a in mou]
                integ = np.zeros(sg.size)
                boxes = (C.c_int * (4 * len(patches)))(*[v for p in patches for v in p])
                dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
                capi.check(capi.lib().suhmo_amr_moulin_source(arr, nlev, boxes, sg.size, dp(pos.reshape(-1)), dp(sg), dp(fl), 1.0, dp(integ), None))
            counts = []
            for k in range(nsteps):
                pi, nv = C.c_int(), C.c_int()
                capi.check(capi.lib().suhmo_amr_timestep(arr, nlev, C.byref(mp), float(m["dt"]), k + 1, C.byref(pi), C.byref(nv), None))
                counts.append((pi.value, nv.value))
            res = [({nm: g.get(model.HipModel.FIELDS[nm]) for nm in NAMES} if g else None) for g in levels]
            for g in levels:
                if g:
                    g.synchronize()
            out[rank] = (counts, res, integ, [(g.get(lv.F_MSRC) if (g and mou is not None) else None) for g in levels])
            for g in reversed(levels):
                if g:
                    g.close()
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


CASES = [
    ("2-levels-2-ranks", 2, 64, 64, ((8, 12, 39, 51),), dict(), 2),
    ("2-levels-4-ranks-patch-on-two-of-them", 4, 64, 64, ((8, 20, 39, 43),), dict(), 2),
    ("3-levels-2-ranks", 2, 64, 64, ((8, 12, 39, 51), (24, 40, 59, 87)), dict(), 2),
    ("2-levels-2-ranks-implicit-gap-moulins", 2, 64, 64, ((8, 12, 39, 51),), dict(diffFactor=1.0, use_impl_diff=1, use_moulin_source=1, distributed_input=7.93e-11), 2),
]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_amr_timestep_on_strips_bitwise(case):
    from suhmo_amd import model
    name, world, nx0, ny0, patches, mpo, nsteps = case
    m = dict(sy.A3_MODEL, **mpo)
    sts = sy.shmip_amr_states(nx0, ny0, patches, rough=0.5)
    A = model.HipAmrModel(nx0, ny0, sts[0]["dx"], sts[0]["dy"], sy.A3_BC, sy.A3_PHYS, m, patches, max_box=MB)
    for l, st in enumerate(sts):
        A.set_state(l, st)
    mou = msrc = iref = None
    if m.get("use_moulin_source"):
        from test_gpu_moulin import moulins
        pos, sg, fl = moulins(5, 3)
        mou = (pos, np.full(5, 1500.0), fl)
        iref = A.moulin_source(*mou, 1.0)
        msrc = [A.get(l, "msrc") for l in range(len(sts))]
    ref_counts = [A.timestep(m["dt"]) for _ in range(nsteps)]
    ref = [{nm: A.get(l, nm) for nm in NAMES} for l in range(len(sts))]
    A.close()
    out, own = run_strips(world, nx0, ny0, patches, sts, m, nsteps, mou)
    for r in range(world):
        assert out[r][0] == ref_counts, (r, out[r][0], ref_counts)
        if mou is not None:
            assert np.array_equal(out[r][2], iref)                 # integrals: the single-process bits on every rank
    if mou is not None:
        for l in range(len(sts)):
            got = np.vstack([out[r][3][l] for r in range(world) if own[r][l]])
            assert np.array_equal(got, msrc[l]), (name, l, "msrc")
    for l in range(len(sts)):
        for nm in NAMES:
            got = np.vstack([out[r][1][l][nm] for r in range(world) if own[r][l]])
            assert np.array_equal(got, ref[l][nm], equal_nan=True), (name, l, nm, float(np.nanmax(np.abs(got - ref[l][nm]))))
