"""Host-side handle on the hydrology time loop of one device-resident level: the single-level
subset of AmrHydro::timeStepFAS (src/AmrHydro.cpp:2254-3460) the C-ABI exposes as
suhmo_level_timestep.  Head = F_PHI and gap height = F_B stay in HBM from step to step."""
import ctypes as C

import numpy as np

from . import capi, level as lv
from . import synthetic as sy
from .capi import check


def model_params(m):
    ub = m.get("ub", (0.0, 0.0))
    return capi.ModelParams(m["rho_i"], m["rho_w"], m["gravity"], m["G"], m["L"], m["ct"], m["cw"], ub[0], ub[1],
                            m["br"], m["lr"], m.get("diffFactor", 0.0), m["distributed_input"], m["eps_picard"],
                            int(m.get("basal_friction", 1)), int(m.get("use_mask_rhs_b", 0)),
                            int(m.get("use_moulin_source", 0)), float(m.get("ramp", 1.0)), int(m.get("use_impl_diff", 0)),
                            int(m.get("head_melt_off", 0)), int(m.get("freeze_icefree_gap", 0)))


class HipModel:
    FIELDS = dict(head=lv.F_PHI, B=lv.F_B, Pi=lv.F_PI, zb=lv.F_ZB, mask=lv.F_MASK, mR=lv.F_MR, Pw=lv.F_PW,
                  qwx=lv.F_QWX, qwy=lv.F_QWY, cd=lv.F_CD, rhs_h=lv.F_RHS, Re=lv.F_RE, msrc=lv.F_MSRC)

    def __init__(self, nx, ny, dx, dy, bc, phys, model, max_box=64, device=0, j0=0, ny_global=None, halo_rows=1):
        """j0 / ny_global: this process holds rows j0 .. j0 + ny - 1 of a level of ny_global rows (one strip per GPU;
        couple the strips with suhmo_amd.multigpu.attach(model.level, ...) before the first step)"""
        self.level = lv.HipLevel(nx, ny, dx, dy, bc, phys, alpha=0.0, beta=-1.0, max_box=max_box, device=device,
                                 j0=j0, ny_global=ny_global, halo_rows=halo_rows)
        self.nx, self.ny, self.dx, self.dy = nx, ny, dx, dy
        self.model = dict(model)
        self._mp = model_params(model)
        self.cur_step = 0

    def set_state(self, f):
        """f: dict with ghosted (ny+2, nx+2) arrays head, B, Pi, zb, mask"""
        L = self.level
        L.set(lv.F_PHI, f["head"][1:-1, 1:-1])
        L.set(lv.F_ACOEF, np.zeros((self.ny, self.nx)))
        for k, fid in (("B", lv.F_B), ("Pi", lv.F_PI), ("zb", lv.F_ZB), ("mask", lv.F_MASK)):
            L.set(fid, f[k], ghosted=True)

    def moulin_source(self, positions, sigma, flux, time_factor=1.0):
        """Calc_moulin_integral + Calc_moulin_source_term_distributed (src/AmrHydro.cpp:1866-2066) -> F_MSRC; returns the integrals"""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
        integ = np.zeros(sg.size)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        check(capi.lib().suhmo_level_moulin_source(self.level.h, sg.size, dp(pos), dp(sg), dp(fl), float(time_factor), dp(integ),
                                                   self.level.stream))
        return integ

    def time_varying_recharge(self, zs, T_K, background):
        """suhmo.time_varying_input (src/AmrHydro.cpp:2849-2861): F_MSRC from the ghosted ice surface height zs"""
        self.level.set(lv.F_ZS, zs, ghosted=True)
        check(capi.lib().suhmo_level_time_varying_recharge(self.level.h, float(T_K), float(background), self.level.stream))

    def timestep(self, dt):
        self.cur_step += 1                                         # src/AmrHydro.cpp:2259
        pi, nv = C.c_int(), C.c_int()
        check(capi.lib().suhmo_level_timestep(self.level.h, C.byref(self._mp), float(dt), self.cur_step,
                                              C.byref(pi), C.byref(nv), self.level.stream))
        return pi.value, nv.value

    def get(self, name, ghosted=False):
        return self.level.get(self.FIELDS[name], ghosted=ghosted)

    def postproc_temporal(self):
        """the daily row of AmrHydro.post_proc_shmip_temporal (suhmo_level_postproc_temporal): avgN, N in the three bands, recharge, discharge"""
        out = np.zeros(6)
        check(capi.lib().suhmo_level_postproc_temporal(self.level.h, C.byref(self._mp), out.ctypes.data_as(C.POINTER(C.c_double)), self.level.stream))
        return out

    def postproc_table_device(self):
        """SHMIP cross-section table reduced on the device (suhmo_level_postproc_table)"""
        t = np.zeros((self.nx, 8))
        check(capi.lib().suhmo_level_postproc_table(self.level.h, C.byref(self._mp), t.ctypes.data_as(C.POINTER(C.c_double)),
                                                    self.level.stream))
        return t

    def postproc_partial(self):
        """column sums over this strip's rows (8 x nx); add them over the ranks, then postproc_finish"""
        t = np.zeros((8, self.nx))
        check(capi.lib().suhmo_level_postproc_partial(self.level.h, C.byref(self._mp), t.ctypes.data_as(C.POINTER(C.c_double)),
                                                      self.level.stream))
        return t

    def postproc_finish(self, sums):
        t, a = np.zeros((self.nx, 8)), np.ascontiguousarray(sums, dtype=np.float64)
        check(capi.lib().suhmo_postproc_finish(a.ctypes.data_as(C.POINTER(C.c_double)), self.nx, self.dx,
                                               t.ctypes.data_as(C.POINTER(C.c_double))))
        return t

    def postproc_table(self):
        """SHMIP cross-section table (src/AmrHydro.cpp:3647-4102) from the device-resident state, reduced on the host"""
        mask = self.get("mask")
        src = np.where(mask > 0.0, self.model["distributed_input"], 0.0)
        return sy.shmip_postproc_table(self.dx, self.dy, self.get("qwx"), self.get("cd", ghosted=True), src,
                                       self.get("mR"), self.get("Pw"), self.get("Pi"), mask, self.model["rho_w"])

    def close(self):
        self.level.close()


class HipAmrModel:
    """The time loop on a hierarchy (base level + nested patches, patches[k] = box of level k+1 in the cells of level k):
    suhmo_amr_timestep over the array of level handles; head and gap height of every level stay in HBM."""

    FIELDS = HipModel.FIELDS

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, model, patches, max_box=64, device=0):
        self.amr = lv.HipAmr(nx0, ny0, dx0, dy0, bc, phys, patches, alpha=0.0, beta=-1.0, max_box=max_box, device=device)
        self.levels = self.amr.levels
        self.model = dict(model)
        self._mp = model_params(model)
        self.cur_step = 0

    def set_state(self, l, f):
        """f: dict with ghosted arrays head, B, Pi, zb, mask of level l (coarse-fine ghost cells: anything, they are interpolated)"""
        L = self.levels[l]
        L.set(lv.F_PHI, f["head"][1:-1, 1:-1])
        L.set(lv.F_ACOEF, np.zeros((L.ny, L.nx)))
        for k, fid in (("B", lv.F_B), ("Pi", lv.F_PI), ("zb", lv.F_ZB), ("mask", lv.F_MASK)):
            L.set(fid, f[k], ghosted=True)

    def timestep(self, dt):
        self.cur_step += 1
        pi, nv = C.c_int(), C.c_int()
        check(capi.lib().suhmo_amr_timestep(self.amr._arr, len(self.levels), C.byref(self._mp), float(dt), self.cur_step,
                                            C.byref(pi), C.byref(nv), self.amr.stream))
        return pi.value, nv.value

    def moulin_source(self, positions, sigma, flux, time_factor=1.0):
        """Calc_moulin_integral over the hierarchy + the source term of every level (suhmo_amr_moulin_source)"""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
        integ = np.zeros(sg.size)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        check(capi.lib().suhmo_amr_moulin_source(self.amr._arr, len(self.levels), None, sg.size, dp(pos), dp(sg), dp(fl), float(time_factor),
                                                 dp(integ), self.amr.stream))
        return integ

    def get(self, l, name, ghosted=False):
        return self.levels[l].get(self.FIELDS[name], ghosted=ghosted)

    def close(self):
        self.amr.close()


class HipHierModel:
    """The time loop on a hierarchy whose levels are unions of boxes (boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the index
    space of level l): suhmo_hier_timestep / suhmo_hier_moulin_source; head and gap height of every box stay in HBM."""

    FIELDS = HipModel.FIELDS

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, model, boxes, max_box=64, device=0, j0=0, ny_global=None, halo_rows=1, options=None):
        self.hier = lv.HipHier(nx0, ny0, dx0, dy0, bc, phys, boxes, alpha=0.0, beta=-1.0, max_box=max_box, device=device,
                               j0=j0, ny_global=ny_global, halo_rows=halo_rows, options=options)
        self.level = self.hier.level
        self.model = dict(model)
        self._mp = model_params(model)
        self.cur_step = 0

    def set_state(self, l, k, f):
        """f: dict with ghosted arrays head, B, Pi, zb, mask of box k of level l (fine-fine / coarse-fine ghost cells: anything).  On a level dealt
        to the ranks a box this rank does not hold is skipped (its owner loads it); loading a mirror is harmless"""
        if not self.hier.held[l][k]:
            return
        L = self.level[l][k]
        L.set(lv.F_PHI, f["head"][1:-1, 1:-1])
        L.set(lv.F_ACOEF, np.zeros((L.ny, L.nx)))
        for key, fid in (("B", lv.F_B), ("Pi", lv.F_PI), ("zb", lv.F_ZB), ("mask", lv.F_MASK)):
            L.set(fid, f[key], ghosted=True)

    def set_states(self, sts):
        for l, bl in enumerate(sts):
            for k, st in enumerate(bl):
                self.set_state(l, k, st)

    def timestep(self, dt):
        self.cur_step += 1
        pi, nv = C.c_int(), C.c_int()
        check(capi.lib().suhmo_hier_timestep(self.hier.h, C.byref(self._mp), float(dt), self.cur_step, C.byref(pi), C.byref(nv),
                                             self.hier.stream))
        return pi.value, nv.value

    def moulin_source(self, positions, sigma, flux, time_factor=1.0):
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
        integ = np.zeros(sg.size)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        check(capi.lib().suhmo_hier_moulin_source(self.hier.h, sg.size, dp(pos), dp(sg), dp(fl), float(time_factor), dp(integ), self.hier.stream))
        return integ

    def get(self, l, k, name, ghosted=False):
        """a field of box k of level l; None where another rank owns the box (levels dealt to the ranks: hier.owns(l, k))"""
        if not self.hier.owns(l, k):
            return None
        return self.level[l][k].get(self.FIELDS[name], ghosted=ghosted)

    def postproc_table_device(self):
        """SHMIP cross-section table of a run with AMR levels: the reference evaluates it on LEVEL 0 ("POST PROC -- 1 LEVEL",
        m_amrGrids[0], src/AmrHydro.cpp:3643-3700; the finer levels enter through the averaged-down head, CoarseAverage :3138-3141) --
        suhmo_level_postproc_table on the base handle of the hierarchy (whole level 0; on rank strips: postproc_partial per rank)"""
        base = self.level[0][0]
        t = np.zeros((base.nx, 8))
        check(capi.lib().suhmo_level_postproc_table(base.h, C.byref(self._mp), t.ctypes.data_as(C.POINTER(C.c_double)), self.hier.stream))
        return t

    def close(self):
        self.hier.close()
