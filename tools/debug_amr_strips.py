import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import numpy as np
from suhmo_amd import synthetic as sy, level, capi
import test_gpu_amr_strips as T
case = T.CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
_, world, nx0, ny0, patches = case
fs = sy.amr_fields(nx0, ny0, patches, lx=64.0, ly=32.0)
sp = dict(sy.SOLVER_DEFAULT, eps=1e-9, norm_thresh=1e-14, max_iter=3, imin=30)
lib = capi.lib()
F_ = level

def steps(lv):
    """two-level sequence of suhmo_amr vcycle, piece by piece; returns snapshots"""
    Cl, Fl = lv[0], lv[1]
    snaps = []
    def snap(tag):
        snaps.append((tag, Fl.get(F_.F_PHI, ghosted=True) if Fl else None, Cl.get(F_.F_PHI),
                      Fl.get(F_.F_BX) if Fl else None, Fl.get(F_.F_RES) if Fl else None, Cl.get(F_.F_RES), Cl.get(F_.F_RHS)))
    ch = capi.check
    if Fl: ch(lib.suhmo_amr2_cf_interp(Cl.h, Fl.h, 0, 0, None))
    else: pass
    snap("cf_interp")
    if Fl:
        ch(lib.suhmo_amr2_fine_update_operator(Cl.h, Fl.h, None))
    else:
        pass
    snap("update_op")
    if Fl: Fl.gsrb(4)
    snap("gsrb")
    if Fl: ch(lib.suhmo_amr2_average(Cl.h, Fl.h, 0, 0, None))
    snap("average")
    if Fl:
        ch(lib.suhmo_amr2_residual(Cl.h, Fl.h, None, None))
        ch(lib.suhmo_amr2_average(Cl.h, Fl.h, F_.F_RES, F_.F_RES, None))
    snap("residual+average")
    # FAS rhs, base-level V-cycle
    rhs0 = Cl.get(F_.F_RHS)
    Cl.axby(F_.F_RHS, F_.F_RES, F_.F_LPHI, 1.0, 1.0)
    ch(lib.suhmo_level_exchange(Cl.h, 0, F_.F_RHS, None))
    snap("fas rhs")
    old = Cl.get(F_.F_PHI)
    Cl.vcycle(sp)
    snap("base vcycle")
    return snaps

G = level.HipAmr(nx0, ny0, fs[0]["dx"], fs[0]["dy"], T.BC, sy.CFG3_PHYS, patches, max_box=32)
G.levels[0].set_inputs(fs[0]); G.levels[0].build_mg_coefficients()
for l in range(1, len(fs)):
    G.levels[l].set_inputs(fs[l])
ref = steps(G.levels)
out, own = T.run_amr_strips(world, nx0, ny0, patches, T.BC, sy.CFG3_PHYS, fs, sp, 0, body=lambda lv, arr, nlev, rank: steps(lv))
for k, (tag, *refv) in enumerate(ref):
    names = ["phi_f(ghosted)", "phi_c", "bx_f", "res_f", "res_c", "rhs_c"]
    for q, nm in enumerate(names):
        fine = q in (0, 2, 3)
        l = 1 if fine else 0
        parts = [out[r][k][1 + q] for r in range(world) if own[r][l] and out[r][k][1 + q] is not None]
        if nm == "phi_f(ghosted)":
            parts = [p[1:-1] for p in parts]; rv = refv[q][1:-1]
        else:
            rv = refv[q]
        got = np.vstack(parts)
        d = np.abs(got - rv)
        rows = np.where(d.max(axis=1) > 0)[0]
        print(tag, nm, "max diff %.3g" % d.max(), "rows", (rows.min(), rows.max(), len(rows)) if len(rows) else None,
              "cols", np.where(d.max(axis=0) > 0)[0][:6] if len(rows) else None)
