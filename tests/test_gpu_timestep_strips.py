"""The time step on rank strips (one strip per GPU): thread-"ranks" on one GPU drive suhmo_level_timestep through
the same hooks the RCCL transport installs; head, gap height, melt rate, fluxes and the iteration counts must equal
the single-process result BIT FOR BIT (which tests/test_gpu_timestep.py ties to the oracle), with the explicit gap
update, with moulins + diffusion + the implicit gap solve (suite B physics), on y-periodic and masked levels."""
import threading

import numpy as np
import pytest

from suhmo_amd import synthetic as sy
from test_gpu_timestep import perturbed_state
from test_gpu_moulin import moulins

pytestmark = pytest.mark.gpu

NAMES = ("head", "B", "mR", "Pw", "cd", "rhs_h", "qwx", "qwy")


def split_state(st, j0, ny):
    return {k: (v[j0:j0 + ny + 2] if isinstance(v, np.ndarray) else v) for k, v in st.items()}


def wrap(st, bc):
    """y-periodic: caller-side ghost rows must be the periodic image"""
    if bc["periodic"][1]:
        for k in ("head", "B", "Pi", "zb", "mask"):
            st[k][0, :], st[k][-1, :] = st[k][-2, :].copy(), st[k][1, :].copy()
    return st


def run(world, nx, ny, st, bc, ph, m, nsteps, halo, mou=None, max_box=16):
    from suhmo_amd import model, multigpu
    if world == 1:
        G = model.HipModel(nx, ny, st["dx"], st["dy"], bc, ph, m, max_box=max_box)
        G.set_state(st)
        integ = G.moulin_source(*mou) if mou else None
        counts = [G.timestep(m["dt"]) for _ in range(nsteps)]
        out = {k: G.get(k) for k in NAMES}
        out["Bg"] = G.get("B", ghosted=True)
        tab = G.postproc_table_device()
        G.close()
        return counts, out, integ, tab
    assert ny % world == 0
    nyl = ny // world
    tr = multigpu.ThreadTransport(world)
    res, err = [None] * world, []

    def worker(rank):
        try:
            j0 = rank * nyl
            G = model.HipModel(nx, nyl, st["dx"], st["dy"], bc, ph, m, max_box=max_box, j0=j0, ny_global=ny, halo_rows=halo)
            G.set_state(split_state(st, j0, nyl))
            ex = multigpu.StripExchanger(G.level, tr, rank, world, bool(bc["periodic"][1]))
            ex.exchange_static()
            import os
            from suhmo_amd import capi
            da = capi.lib().suhmo_level_agglomerated_depth(G.level.h)
            assert (da > 0) == (int(os.environ.get("SUHMO_AGG_MIN_CELLS", "0")) > 0), da      # the variant really runs what its name says
            integ = G.moulin_source(*mou) if mou else None
            counts = [G.timestep(m["dt"]) for _ in range(nsteps)]
            out = {k: G.get(k) for k in NAMES}
            out["Bg"] = G.get("B", ghosted=True)
            res[rank] = (counts, out, integ, G.postproc_partial(), G)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)
            tr.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    assert all(r[0] == res[0][0] for r in res)
    out = {}
    for k in NAMES:
        if k == "qwy":
            out[k] = np.vstack([r[1][k][:-1] for r in res] + [res[-1][1][k][-1:]])
            for a, b in zip(res[:-1], res[1:]):
                assert np.array_equal(a[1][k][-1], b[1][k][0])             # the face two strips share
        else:
            out[k] = np.vstack([r[1][k] for r in res])
    out["Bg"] = np.vstack([res[0][1]["Bg"][:1]] + [r[1]["Bg"][1:-1] for r in res] + [res[-1][1]["Bg"][-1:]])
    for a, b in zip(res[:-1], res[1:]):                                    # halo rows of the gap height = the neighbour's rows
        assert np.array_equal(a[1]["Bg"][-1, 1:-1], b[1]["Bg"][1, 1:-1]) and np.array_equal(b[1]["Bg"][0, 1:-1], a[1]["Bg"][-2, 1:-1])
    G0 = res[0][4]
    tab = G0.postproc_finish(sum(r[3] for r in res))
    [r[4].close() for r in res]
    return res[0][0], out, res[0][2], tab


CASES = [
    # name, nx, ny, bc, phys, model overrides, mask holes, steps, moulins
    ("a3-explicit", 128, 64, sy.A3_BC, sy.A3_PHYS, dict(), False, 3, 0),
    ("yperiodic-mask", 64, 64, sy.CONV_BC, dict(sy.A3_PHYS, use_mask_gradients=1, cutOffbr=0.02, maxOffbr=0.08, cutOffB=1),
     dict(use_mask_rhs_b=1, G=0.05), True, 2, 0),
    ("moulins-diffusion-explicit", 128, 64, sy.A3_BC, sy.A3_PHYS, dict(use_moulin_source=1, diffFactor=1.0, distributed_input=7.93e-11), False, 2, 7),
    ("moulins-diffusion-implicit", 128, 64, sy.A3_BC, sy.A3_PHYS,
     dict(use_moulin_source=1, diffFactor=1.0, use_impl_diff=1, distributed_input=7.93e-11), False, 2, 7),
]


@pytest.mark.parametrize("world,halo,agg", [(2, 4, 0), (4, 4, 0), (2, 1, 0), (4, 16, 0), (2, 4, 1500), (4, 16, 300)])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_timestep_on_strips_bitwise(case, world, halo, agg, monkeypatch):
    """agg > 0: multigrid depths whose strip holds fewer cells run agglomerated on a whole-level copy (suhmo_agg.hip), the head
    solve's and the implicit gap-height solve's alike"""
    monkeypatch.setenv("SUHMO_AGG_MIN_CELLS", str(agg))
    name, nx, ny, bc, ph, mpo, holes, nsteps, nm = case
    m = dict(sy.A3_MODEL, **mpo)
    st = wrap(perturbed_state(nx, ny, 17, holes), bc)
    mou = None
    if nm:
        pos, sg, fl = moulins(nm, 5)
        mou = (pos, sg, fl, 1.0)
    c1, o1, i1, t1 = run(1, nx, ny, st, bc, ph, m, nsteps, halo, mou)
    cn, on, i_n, tn = run(world, nx, ny, st, bc, ph, m, nsteps, halo, mou)
    assert c1 == cn, (c1, cn)
    if nm:
        assert np.array_equal(i1, i_n)                 # whole-level integrals, evaluated redundantly in the same order
    for k in NAMES:
        assert np.array_equal(o1[k], on[k], equal_nan=True), (name, k, float(np.nanmax(np.abs(o1[k] - on[k]))))
    assert np.array_equal(o1["Bg"][1:-1, :], on["Bg"][1:-1, :]) and np.array_equal(o1["Bg"][:, 1:-1], on["Bg"][:, 1:-1])
    ok = np.isfinite(t1)                                   # cd = 0/0 where nothing opens the gap: NaN in both
    assert np.array_equal(ok, np.isfinite(tn))
    scale = np.max(np.where(ok, np.abs(t1), 0.0), axis=0)
    assert np.all(np.where(ok, np.abs(t1 - tn), 0.0) <= 1e-12 * scale)     # column sums added in a different order
