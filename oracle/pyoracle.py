"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- pinned end-to-end by the reference's SHMIP A tables, no kernel-level vectors
(see oracle/suhmo_oracle.h).  May be
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package suhmo_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrPhys(C.Structure):
    _fields_ = [("A", C.c_double), ("omega", C.c_double), ("nu", C.c_double),
                ("cutOffbr", C.c_double), ("maxOffbr", C.c_double),
                ("rho_w_g", C.c_double), ("grav", C.c_double),
                ("cutOffB", C.c_int), ("use_NL", C.c_int), ("use_mask_gradients", C.c_int)]


class OrBC(C.Structure):
    _fields_ = [("type", (C.c_int * 2) * 2), ("value", (C.c_double * 2) * 2),
                ("periodic", C.c_int * 2)]


class OrModelParams(C.Structure):
    _fields_ = [("rho_i", C.c_double), ("rho_w", C.c_double), ("gravity", C.c_double), ("G", C.c_double),
                ("L", C.c_double), ("ct", C.c_double), ("cw", C.c_double), ("ub0", C.c_double), ("ub1", C.c_double),
                ("br", C.c_double), ("lr", C.c_double), ("diffFactor", C.c_double),
                ("distributed_input", C.c_double), ("eps_picard", C.c_double),
                ("basal_friction", C.c_int), ("use_mask_rhs_b", C.c_int), ("use_moulin_source", C.c_int),
                ("ramp", C.c_double), ("use_impl_diff", C.c_int),
                ("head_melt_off", C.c_int), ("freeze_icefree_gap", C.c_int)]


class OrSolverParams(C.Structure):
    _fields_ = [("num_smooth", C.c_int), ("num_bottom", C.c_int), ("max_iter", C.c_int),
                ("iter_min", C.c_int), ("imin", C.c_int), ("eps", C.c_double),
                ("hang", C.c_double), ("norm_thresh", C.c_double),
                ("bcoeff_otf", C.c_int), ("max_depth", C.c_int)]


F_PHI, F_RHS, F_ACOEF, F_B, F_PI, F_ZB, F_MASK, F_BX, F_BY, F_LAMBDA, F_RES, F_LPHI, F_NL, F_DNL = range(14)
F_PHIOLD, F_CORR = 14, 15


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.or_level_create.restype = C.c_void_p
        L.or_level_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                      C.POINTER(OrBC), C.POINTER(OrPhys), C.c_double, C.c_double, C.c_int]
        L.or_level_destroy.argtypes = [C.c_void_p]
        L.or_level_num_depths.argtypes = [C.c_void_p]
        L.or_level_num_boxes.argtypes = [C.c_void_p]
        L.or_level_set.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int]
        L.or_level_get.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int]
        for name in ("or_level_exchange",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_level_bc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        for name in ("or_level_reset_lambda", "or_level_nonlinear", "or_level_residual",
                     "or_level_restrict_residual", "or_level_restrict_r",
                     "or_level_update_operator", "or_level_average_operator"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.or_level_gsrb.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_level_apply_op.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_level_prolong_increment.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_level_build_mg_coefficients.argtypes = [C.c_void_p]
        L.or_level_norm.restype = C.c_double
        L.or_level_norm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.or_level_vcycle.argtypes = [C.c_void_p, C.POINTER(OrSolverParams)]
        L.or_level_solve.restype = C.c_int
        L.or_level_solve.argtypes = [C.c_void_p, C.POINTER(OrSolverParams), dp]
        L.or_model_create.restype = C.c_void_p
        L.or_model_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC),
                                      C.POINTER(OrPhys), C.POINTER(OrModelParams)]
        L.or_model_destroy.argtypes = [C.c_void_p]
        L.or_model_gap_solver_layout.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_model_field.restype = dp
        L.or_model_field.argtypes = [C.c_void_p, C.c_int]
        L.or_model_timestep.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.or_model_step_index.argtypes = [C.c_void_p]
        L.or_model_set_ramp.argtypes = [C.c_void_p, C.c_double]
        L.or_time_varying_recharge.argtypes = [C.c_int, dp, C.c_double, C.c_double, dp]
        L.or_amr_model_create.restype = C.c_void_p
        L.or_amr_model_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC), C.POINTER(OrPhys),
                                          C.POINTER(OrModelParams), C.c_int, C.POINTER(C.c_int)]
        L.or_amr_model_destroy.argtypes = [C.c_void_p]
        L.or_amr_model_gap_solver_layout.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_amr_model_level.restype = C.c_void_p
        L.or_amr_model_level.argtypes = [C.c_void_p, C.c_int]
        L.or_pwl_fill.argtypes = [C.c_void_p, C.c_void_p, dp, dp]
        L.or_quadcf_fill.argtypes = [C.c_void_p, C.c_void_p, dp, dp]
        L.or_amr_model_field.restype = dp
        L.or_amr_model_field.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_amr_model_timestep.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.or_amr_model_moulin_source.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_double, dp]
        L.or_moulin_source.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp, dp, C.c_double, dp, dp]
        L.or_amr2_create.restype = C.c_void_p
        L.or_amr2_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC), C.POINTER(OrPhys),
                                     C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        L.or_amr2_destroy.argtypes = [C.c_void_p]
        L.or_amr2_fine_io.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, C.c_int]
        for name in ("or_amr2_cf_interp_phi", "or_amr2_fine_residual", "or_amr2_fine_update_operator"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.or_amr2_fine_gsrb.argtypes = [C.c_void_p, C.c_int]
        L.or_amr2_fine_apply_op.argtypes = [C.c_void_p, C.c_int]
        L.or_amr2_residual.restype = C.c_double
        L.or_amr2_residual.argtypes = [C.c_void_p]
        L.or_amr2_vcycle.argtypes = [C.c_void_p, C.POINTER(OrSolverParams)]
        L.or_amr2_solve.restype = C.c_int
        L.or_amr2_solve.argtypes = [C.c_void_p, C.POINTER(OrSolverParams), dp]
        L.or_amr_create.restype = C.c_void_p
        L.or_amr_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC), C.POINTER(OrPhys),
                                    C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.or_amr_destroy.argtypes = [C.c_void_p]
        L.or_amr_patch_io.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.c_int, C.c_int]
        L.or_amr_residual.restype = C.c_double
        L.or_amr_residual.argtypes = [C.c_void_p]
        L.or_amr_vcycle.argtypes = [C.c_void_p, C.POINTER(OrSolverParams)]
        L.or_amr_solve.restype = C.c_int
        L.or_amr_solve.argtypes = [C.c_void_p, C.POINTER(OrSolverParams), dp]
        L.or_model_set_cutoffb.argtypes = [C.c_void_p, C.c_int]
        L.or_amrm_model_create.restype = C.c_void_p
        L.or_amrm_model_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC), C.POINTER(OrPhys),
                                           C.POINTER(OrModelParams), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.or_amrm_model_destroy.argtypes = [C.c_void_p]
        L.or_amrm_model_gap_solver_layout.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_amrm_model_field.restype = dp
        L.or_amrm_model_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.or_amrm_model_timestep.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.or_amrm_model_moulin_source.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_double, dp]
        ip = C.POINTER(C.c_int)
        L.or_amrm_create.restype = C.c_void_p
        L.or_amrm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(OrBC), C.POINTER(OrPhys),
                                     C.c_double, C.c_double, C.c_int, ip, ip]
        L.or_amrm_destroy.argtypes = [C.c_void_p]
        L.or_amrm_box_io.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_int]
        L.or_amrm_residual.restype = C.c_double
        L.or_amrm_residual.argtypes = [C.c_void_p]
        L.or_amrm_vcycle.argtypes = [C.c_void_p, C.POINTER(OrSolverParams)]
        L.or_amrm_solve.restype = C.c_int
        L.or_amrm_solve.argtypes = [C.c_void_p, C.POINTER(OrSolverParams), dp]
        L.or_amrm_cf_interp_phi.argtypes = [C.c_void_p, C.c_int]
        L.or_amrm_exchange.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.or_amrm_gsrb.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_amrm_level_residual.argtypes = [C.c_void_p, C.c_int]
        L.or_amrm_update_operator.argtypes = [C.c_void_p, C.c_int]
        L.or_amrm_average_down.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_prolong2_global.argtypes = [dp, dp, C.c_int, C.c_int]
        L.or_divergence_global.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.or_difterm_global.argtypes = [dp, dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.or_getflux_global.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        _LIB = L
    return _LIB


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_phys(p):
    return OrPhys(p["A"], p["omega"], p["nu"], p["cutOffbr"], p["maxOffbr"],
                  p.get("rho_w_g", 9800.0), p.get("grav", 9.8),
                  int(p.get("cutOffB", 0)), int(p.get("use_NL", 1)), int(p.get("use_mask_gradients", 0)))


def make_bc(bc):
    b = OrBC()
    for d in range(2):
        for s in range(2):
            b.type[d][s] = int(bc["type"][d][s])
            b.value[d][s] = float(bc["value"][d][s])
        b.periodic[d] = int(bc["periodic"][d])
    return b


def make_solver_params(sp):
    return OrSolverParams(sp.get("num_smooth", 4), sp.get("num_bottom", 16), sp.get("max_iter", 100),
                          sp.get("iter_min", 2), sp.get("imin", 5), sp.get("eps", 1e-7),
                          sp.get("hang", 0.01), sp.get("norm_thresh", 1e-7),
                          int(sp.get("bcoeff_otf", 1)), sp.get("max_depth", -1))


class OracleLevel:
    """One AMR level of nx x ny cells split into boxes of at most max_box^2 cells."""

    def __init__(self, nx, ny, dx, dy, bc, phys, alpha=0.0, beta=-1.0, max_box=64, nthreads=1):
        self.nx, self.ny, self.dx, self.dy = nx, ny, dx, dy
        self._bc, self._ph = make_bc(bc), make_phys(phys)
        self.h = lib().or_level_create(nx, ny, dx, dy, max_box, C.byref(self._bc), C.byref(self._ph),
                                       alpha, beta, nthreads)
        self.ndepth = lib().or_level_num_depths(self.h)
        self.nbox = lib().or_level_num_boxes(self.h)

    def close(self):
        if self.h:
            lib().or_level_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def shape(self, field, depth=0, ghosted=False):
        nx, ny = self.nx >> depth, self.ny >> depth
        if field == F_BX:
            return (ny, nx + 1)
        if field == F_BY:
            return (ny + 1, nx)
        return (ny + 2, nx + 2) if ghosted else (ny, nx)

    def set(self, field, arr, depth=0, ghosted=False):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.shape(field, depth, ghosted), (a.shape, self.shape(field, depth, ghosted))
        lib().or_level_set(self.h, depth, field, _dp(a), int(ghosted))

    def get(self, field, depth=0, ghosted=False):
        out = np.zeros(self.shape(field, depth, ghosted), dtype=np.float64)
        lib().or_level_get(self.h, depth, field, _dp(out), int(ghosted))
        return out

    def set_inputs(self, f):
        """f: dict from suhmo_amd.synthetic (phi, rhs, aCoef valid; B, Pi, zb, mask ghosted)."""
        self.set(F_PHI, f["phi"])
        self.set(F_RHS, f["rhs"])
        self.set(F_ACOEF, f["aCoef"])
        for k, fid in (("B", F_B), ("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK)):
            self.set(fid, f[k], ghosted=True)
        if "bx" in f:
            self.set(F_BX, f["bx"])
            self.set(F_BY, f["by"])

    # restated methods
    def exchange(self, field, depth=0): lib().or_level_exchange(self.h, depth, field)
    def bc(self, field, homogeneous, depth=0): lib().or_level_bc(self.h, depth, field, int(homogeneous))
    def reset_lambda(self, depth=0): lib().or_level_reset_lambda(self.h, depth)
    def nonlinear(self, depth=0): lib().or_level_nonlinear(self.h, depth)
    def gsrb(self, sweeps=1, depth=0): lib().or_level_gsrb(self.h, depth, sweeps)
    def apply_op(self, homogeneous=False, depth=0): lib().or_level_apply_op(self.h, depth, int(homogeneous))
    def residual(self, depth=0): lib().or_level_residual(self.h, depth)
    def restrict_residual(self, depth=0): lib().or_level_restrict_residual(self.h, depth)
    def restrict_r(self, depth=0): lib().or_level_restrict_r(self.h, depth)
    def update_operator(self, depth=0): lib().or_level_update_operator(self.h, depth)
    def average_operator(self, depth): lib().or_level_average_operator(self.h, depth)
    def build_mg_coefficients(self): lib().or_level_build_mg_coefficients(self.h)
    def norm(self, field, ord=0, depth=0): return lib().or_level_norm(self.h, depth, field, ord)

    def prolong_increment(self, coarse_corr, depth=0):
        a = np.ascontiguousarray(coarse_corr, dtype=np.float64)
        lib().or_level_prolong_increment(self.h, depth, _dp(a))

    def vcycle(self, sp):
        s = make_solver_params(sp)
        lib().or_level_vcycle(self.h, C.byref(s))

    def solve(self, sp):
        s = make_solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = lib().or_level_solve(self.h, C.byref(s), _dp(hist))
        return n, hist[: n + 1]


def prolong2(fine, coarse_ghosted):
    f = np.ascontiguousarray(fine, dtype=np.float64).copy()
    c = np.ascontiguousarray(coarse_ghosted, dtype=np.float64)
    ny, nx = f.shape
    assert c.shape == (ny // 2 + 2, nx // 2 + 2)
    lib().or_prolong2_global(_dp(f), _dp(c), nx, ny)
    return f


def divergence(ux, uy, dx, dy, div0=None):
    ny, nx = uy.shape[0] - 1, ux.shape[1] - 1
    d = np.zeros((ny, nx)) if div0 is None else np.ascontiguousarray(div0, dtype=np.float64).copy()
    lib().or_divergence_global(_dp(np.ascontiguousarray(ux)), _dp(np.ascontiguousarray(uy)), _dp(d), nx, ny, dx, dy)
    return d


def difterm(phi_ghosted, dxf, dyf, dx, dy):
    ny, nx = phi_ghosted.shape[0] - 2, phi_ghosted.shape[1] - 2
    out = np.zeros((ny, nx))
    lib().or_difterm_global(_dp(np.ascontiguousarray(phi_ghosted)), _dp(np.ascontiguousarray(dxf)),
                            _dp(np.ascontiguousarray(dyf)), _dp(out), nx, ny, dx, dy)
    return out


def getflux(phi_ghosted, bface, direction, beta, dx_dir, ref=1):
    ny, nx = phi_ghosted.shape[0] - 2, phi_ghosted.shape[1] - 2
    out = np.zeros_like(bface)
    lib().or_getflux_global(_dp(np.ascontiguousarray(phi_ghosted)), _dp(np.ascontiguousarray(bface)), _dp(out),
                            nx, ny, direction, beta, dx_dir, ref)
    return out


# ---- time loop ("next rows"): one AmrHydro::timeStepFAS per call
OM_H, OM_B, OM_BOLD, OM_PI, OM_ZB, OM_MASK, OM_MR, OM_PW, OM_SRC, OM_RHSH, OM_CD, OM_GRADX, OM_GRADY, OM_RE, OM_HLAG, OM_MSRC = range(16)
OM_QWX, OM_QWY = 100, 101


def make_model_params(m):
    return OrModelParams(m["rho_i"], m["rho_w"], m["gravity"], m["G"], m["L"], m["ct"], m["cw"], m["ub"][0], m["ub"][1],
                         m["br"], m["lr"], m["diffFactor"], m["distributed_input"], m["eps_picard"],
                         int(m["basal_friction"]), int(m.get("use_mask_rhs_b", 0)), int(m.get("use_moulin_source", 0)),
                         float(m.get("ramp", 1.0)), int(m.get("use_impl_diff", 0)),
                         int(m.get("head_melt_off", 0)), int(m.get("freeze_icefree_gap", 0)))


class OracleModel:
    """Hydrology time loop on one level: head (Picard + FAS solve) and gap height (forward Euler)."""

    def __init__(self, nx, ny, dx, dy, bc, phys, model, max_box=64, nthreads=1):
        self.level = OracleLevel(nx, ny, dx, dy, bc, phys, 0.0, -1.0, max_box, nthreads)
        self.nx, self.ny = nx, ny
        self._mp = make_model_params(model)
        self.h = lib().or_model_create(self.level.h, nx, ny, dx, dy, C.byref(self.level._bc), C.byref(self.level._ph),
                                       C.byref(self._mp))
        lib().or_model_gap_solver_layout(self.h, max_box, nthreads)

    def field(self, fid):
        """numpy VIEW of a model array (ghosted cells (ny+2, nx+2); QWX (ny, nx+1); QWY (ny+1, nx))"""
        p = lib().or_model_field(self.h, fid)
        shape = {OM_QWX: (self.ny, self.nx + 1), OM_QWY: (self.ny + 1, self.nx)}.get(fid, (self.ny + 2, self.nx + 2))
        return np.ctypeslib.as_array(p, shape=shape)

    def set_state(self, f):
        """f: dict with ghosted arrays head, B, Pi, zb, mask"""
        for k, fid in (("head", OM_H), ("B", OM_B), ("Pi", OM_PI), ("zb", OM_ZB), ("mask", OM_MASK)):
            self.field(fid)[:] = f[k]
        L = self.level
        L.set(F_ACOEF, np.zeros((self.ny, self.nx)))
        for k, fid in (("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK), ("B", F_B)):
            L.set(fid, f[k], ghosted=True)

    def timestep(self, dt):
        pi, nv = C.c_int(), C.c_int()
        rc = lib().or_model_timestep(self.h, dt, C.byref(pi), C.byref(nv))
        if rc:
            raise RuntimeError("Picard loop did not converge (> 100 iterations)")
        return pi.value, nv.value

    def close(self):
        if self.h:
            lib().or_model_destroy(self.h)
            self.h = None
            self.level.close()


def time_varying_recharge(zs, T_K, background):
    """oracle/time_loop.c:or_time_varying_recharge on an array of ice surface heights"""
    a = np.ascontiguousarray(zs, dtype=np.float64)
    out = np.zeros_like(a)
    lib().or_time_varying_recharge(a.size, _dp(a.reshape(-1)), float(T_K), float(background), _dp(out.reshape(-1)))
    return out


class OracleAmrModel:
    """Hydrology time loop on a hierarchy (base level + nested patches, patches[k] = box of level k+1 in level-k cells):
    oracle/amr_step.c"""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, model, patches, max_box=64, nthreads=1):
        self.level = OracleLevel(nx0, ny0, dx0, dy0, bc, phys, 0.0, -1.0, max_box, nthreads)
        self.patches = [tuple(int(v) for v in p) for p in patches]
        self.nlev = 1 + len(self.patches)
        self.dims = [(nx0, ny0)] + [(2 * (p[2] - p[0] + 1), 2 * (p[3] - p[1] + 1)) for p in self.patches]
        self._mp = make_model_params(model)
        flat = (C.c_int * max(4 * len(self.patches), 1))(*[v for p in self.patches for v in p])
        self.h = lib().or_amr_model_create(self.level.h, nx0, ny0, dx0, dy0, C.byref(self.level._bc), C.byref(self.level._ph),
                                           C.byref(self._mp), self.nlev, flat)
        self.level.set(F_ACOEF, np.zeros((ny0, nx0)))
        lib().or_amr_model_gap_solver_layout(self.h, max_box, nthreads)

    def field(self, l, fid):
        nx, ny = self.dims[l]
        p = lib().or_amr_model_field(self.h, l, fid)
        shape = {OM_QWX: (ny, nx + 1), OM_QWY: (ny + 1, nx)}.get(fid, (ny + 2, nx + 2))
        return np.ctypeslib.as_array(p, shape=shape)

    def set_state(self, l, f):
        for k, fid in (("head", OM_H), ("B", OM_B), ("Pi", OM_PI), ("zb", OM_ZB), ("mask", OM_MASK)):
            self.field(l, fid)[:] = f[k]

    def fill_ghosts(self, l, fid, kind="pwl"):
        """coarse-fine ghost cells of field fid of level l from level l-1: PiecewiseLinearFillPatch or QuadCFInterp"""
        f = lib().or_pwl_fill if kind == "pwl" else lib().or_quadcf_fill
        f(lib().or_amr_model_level(self.h, l), lib().or_amr_model_level(self.h, l - 1), lib().or_amr_model_field(self.h, l, fid),
          lib().or_amr_model_field(self.h, l - 1, fid))

    def moulin_source(self, positions, sigma, flux, time_factor=1.0):
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
        integ = np.zeros(sg.size)
        lib().or_amr_model_moulin_source(self.h, sg.size, _dp(pos), _dp(sg), _dp(fl), float(time_factor), _dp(integ))
        return integ

    def timestep(self, dt):
        pi, nv = C.c_int(), C.c_int()
        rc = lib().or_amr_model_timestep(self.h, dt, C.byref(pi), C.byref(nv))
        if rc:
            raise RuntimeError("AMR time step failed (rc %d)" % rc)
        return pi.value, nv.value

    def close(self):
        if self.h:
            lib().or_amr_model_destroy(self.h)
            self.h = None
            self.level.close()


class OracleAmr2:
    """Base level + one fine patch (coarse index box ci0..ci1 x cj0..cj1, refined by 2): oracle/amr2.c"""

    def __init__(self, nxc, nyc, dxc, dyc, bc, phys, patch, alpha=0.0, beta=-1.0, max_box=64, nthreads=1):
        self.coarse = OracleLevel(nxc, nyc, dxc, dyc, bc, phys, alpha, beta, max_box, nthreads)
        self.patch = tuple(int(v) for v in patch)
        ci0, cj0, ci1, cj1 = self.patch
        self.nxp, self.nyp = 2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1)
        self.h = lib().or_amr2_create(self.coarse.h, nxc, nyc, dxc, dyc, C.byref(self.coarse._bc), C.byref(self.coarse._ph),
                                      alpha, beta, ci0, cj0, ci1, cj1)

    def fine_shape(self, field, ghosted=False):
        if field == F_BX:
            return (self.nyp, self.nxp + 1)
        if field == F_BY:
            return (self.nyp + 1, self.nxp)
        return (self.nyp + 2, self.nxp + 2) if ghosted else (self.nyp, self.nxp)

    def fine_set(self, field, arr, ghosted=False):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.fine_shape(field, ghosted), (a.shape, self.fine_shape(field, ghosted))
        lib().or_amr2_fine_io(self.h, field, _dp(a), int(ghosted), 1)

    def fine_get(self, field, ghosted=False):
        out = np.zeros(self.fine_shape(field, ghosted))
        lib().or_amr2_fine_io(self.h, field, _dp(out), int(ghosted), 0)
        return out

    def set_fine_inputs(self, f):
        self.fine_set(F_PHI, f["phi"])
        self.fine_set(F_RHS, f["rhs"])
        self.fine_set(F_ACOEF, f["aCoef"])
        for k, fid in (("B", F_B), ("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK)):
            self.fine_set(fid, f[k], ghosted=True)

    def cf_interp(self): lib().or_amr2_cf_interp_phi(self.h)
    def fine_gsrb(self, sweeps): lib().or_amr2_fine_gsrb(self.h, sweeps)
    def fine_residual(self): lib().or_amr2_fine_residual(self.h)
    def fine_apply_op(self, homogeneous=False): lib().or_amr2_fine_apply_op(self.h, int(homogeneous))
    def fine_update_operator(self): lib().or_amr2_fine_update_operator(self.h)
    def residual(self): return lib().or_amr2_residual(self.h)

    def vcycle(self, sp):
        s = make_solver_params(sp)
        lib().or_amr2_vcycle(self.h, C.byref(s))

    def solve(self, sp):
        s = make_solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = lib().or_amr2_solve(self.h, C.byref(s), _dp(hist))
        return n, hist[: n + 1]

    def close(self):
        if self.h:
            lib().or_amr2_destroy(self.h)
            self.h = None
            self.coarse.close()


def moulin_source(nx, ny, dx, dy, positions, sigma, flux, time_factor=1.0):
    """(src (ny, nx), integrals (n,)): oracle/time_loop.c:or_moulin_source"""
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
    sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
    nm = sg.size
    integ, src = np.zeros(nm), np.zeros((ny, nx))
    lib().or_moulin_source(nx, ny, dx, dy, nm, _dp(pos), _dp(sg), _dp(fl), float(time_factor), _dp(integ), _dp(src))
    return src, integ


class OracleAmr:
    """Base level + nested patches (patches[k] = box of level k+1 in the index space of level k): oracle/amrn.c"""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, patches, alpha=0.0, beta=-1.0, max_box=64, nthreads=1):
        self.coarse = OracleLevel(nx0, ny0, dx0, dy0, bc, phys, alpha, beta, max_box, nthreads)
        self.patches = [tuple(int(v) for v in p) for p in patches]
        self.nlev = 1 + len(self.patches)
        flat = (C.c_int * (4 * len(self.patches)))(*[v for p in self.patches for v in p])
        self.h = lib().or_amr_create(self.coarse.h, nx0, ny0, dx0, dy0, C.byref(self.coarse._bc), C.byref(self.coarse._ph),
                                     alpha, beta, self.nlev, flat)

    def patch_shape(self, l, field, ghosted=False):
        ci0, cj0, ci1, cj1 = self.patches[l - 1]
        nxp, nyp = 2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1)
        if field == F_BX:
            return (nyp, nxp + 1)
        if field == F_BY:
            return (nyp + 1, nxp)
        return (nyp + 2, nxp + 2) if ghosted else (nyp, nxp)

    def patch_set(self, l, field, arr, ghosted=False):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.patch_shape(l, field, ghosted), (a.shape, self.patch_shape(l, field, ghosted))
        lib().or_amr_patch_io(self.h, l, field, _dp(a), int(ghosted), 1)

    def patch_get(self, l, field, ghosted=False):
        out = np.zeros(self.patch_shape(l, field, ghosted))
        lib().or_amr_patch_io(self.h, l, field, _dp(out), int(ghosted), 0)
        return out

    def set_patch_inputs(self, l, f):
        self.patch_set(l, F_PHI, f["phi"]); self.patch_set(l, F_RHS, f["rhs"]); self.patch_set(l, F_ACOEF, f["aCoef"])
        for k, fid in (("B", F_B), ("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK)):
            self.patch_set(l, fid, f[k], ghosted=True)

    def residual(self): return lib().or_amr_residual(self.h)

    def vcycle(self, sp):
        s = make_solver_params(sp)
        lib().or_amr_vcycle(self.h, C.byref(s))

    def solve(self, sp):
        s = make_solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = lib().or_amr_solve(self.h, C.byref(s), _dp(hist))
        return n, hist[: n + 1]

    def close(self):
        if self.h:
            lib().or_amr_destroy(self.h)
            self.h = None
            self.coarse.close()


class OracleAmrM:
    """Base level + levels that are unions of boxes (boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the index space of level l):
    oracle/amrm.c"""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, boxes, alpha=0.0, beta=-1.0, max_box=64, nthreads=1):
        self.coarse = OracleLevel(nx0, ny0, dx0, dy0, bc, phys, alpha, beta, max_box, nthreads)
        self.boxes = [[tuple(int(v) for v in b) for b in bl] for bl in boxes]
        self.nlev = 1 + len(self.boxes)
        nbox = (C.c_int * self.nlev)(0, *[len(bl) for bl in self.boxes])
        flat = [v for bl in self.boxes for b in bl for v in b]
        arr = (C.c_int * max(len(flat), 1))(*flat)
        self.h = lib().or_amrm_create(self.coarse.h, nx0, ny0, dx0, dy0, C.byref(self.coarse._bc), C.byref(self.coarse._ph),
                                      alpha, beta, self.nlev, nbox, arr)
        if not self.h:
            self.coarse.close()
            raise ValueError("boxes misaligned, overlapping or not properly nested")

    def box_shape(self, l, k, field, ghosted=False):
        lo0, lo1, hi0, hi1 = self.boxes[l - 1][k]
        nxp, nyp = hi0 - lo0 + 1, hi1 - lo1 + 1
        if field == F_BX:
            return (nyp, nxp + 1)
        if field == F_BY:
            return (nyp + 1, nxp)
        return (nyp + 2, nxp + 2) if ghosted else (nyp, nxp)

    def box_set(self, l, k, field, arr, ghosted=False):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.box_shape(l, k, field, ghosted), (a.shape, self.box_shape(l, k, field, ghosted))
        lib().or_amrm_box_io(self.h, l, k, field, _dp(a), int(ghosted), 1)

    def box_get(self, l, k, field, ghosted=False):
        out = np.zeros(self.box_shape(l, k, field, ghosted))
        lib().or_amrm_box_io(self.h, l, k, field, _dp(out), int(ghosted), 0)
        return out

    def set_box_inputs(self, l, k, f):
        self.box_set(l, k, F_PHI, f["phi"]); self.box_set(l, k, F_RHS, f["rhs"]); self.box_set(l, k, F_ACOEF, f["aCoef"])
        for key, fid in (("B", F_B), ("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK)):
            self.box_set(l, k, fid, f[key], ghosted=True)

    def set_inputs(self, fs):
        """fs as suhmo_amd.synthetic.amrm_fields returns it"""
        self.coarse.set_inputs(fs[0])
        self.coarse.build_mg_coefficients()
        for l in range(1, self.nlev):
            for k, f in enumerate(fs[l]):
                self.set_box_inputs(l, k, f)

    def level_array(self, l, field):
        """valid cells of level l >= 1 over its whole domain (NaN where the level has no box)"""
        nx, ny = self.coarse.nx << l, self.coarse.ny << l
        out = np.full((ny, nx), np.nan)
        for k, (lo0, lo1, hi0, hi1) in enumerate(self.boxes[l - 1]):
            out[lo1:hi1 + 1, lo0:hi0 + 1] = self.box_get(l, k, field)
        return out

    def residual(self): return lib().or_amrm_residual(self.h)
    def cf_interp_phi(self, l): lib().or_amrm_cf_interp_phi(self.h, l)
    def exchange(self, l, field, corners=False): lib().or_amrm_exchange(self.h, l, field, int(corners))
    def gsrb(self, l, sweeps): lib().or_amrm_gsrb(self.h, l, sweeps)
    def level_residual(self, l): lib().or_amrm_level_residual(self.h, l)
    def update_operator(self, l): lib().or_amrm_update_operator(self.h, l)
    def average_down(self, l, field): lib().or_amrm_average_down(self.h, l, field)

    def vcycle(self, sp):
        s = make_solver_params(sp)
        lib().or_amrm_vcycle(self.h, C.byref(s))

    def solve(self, sp):
        s = make_solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = lib().or_amrm_solve(self.h, C.byref(s), _dp(hist))
        return n, hist[: n + 1]

    def close(self):
        if self.h:
            lib().or_amrm_destroy(self.h)
            self.h = None
            self.coarse.close()


class OracleAmrMModel:
    """Hydrology time loop on a hierarchy whose levels are unions of boxes (boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the
    index space of level l): oracle/amr_step_m.c"""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, model, boxes, max_box=64, nthreads=1):
        self.level = OracleLevel(nx0, ny0, dx0, dy0, bc, phys, 0.0, -1.0, max_box, nthreads)
        self.boxes = [[(0, 0, nx0 - 1, ny0 - 1)]] + [[tuple(int(v) for v in b) for b in bl] for bl in boxes]
        self.nlev = len(self.boxes)
        self._mp = make_model_params(model)
        nbox = (C.c_int * self.nlev)(*[len(bl) for bl in self.boxes])
        flat = [v for bl in self.boxes[1:] for b in bl for v in b]
        arr = (C.c_int * max(len(flat), 1))(*flat)
        self.h = lib().or_amrm_model_create(self.level.h, nx0, ny0, dx0, dy0, C.byref(self.level._bc), C.byref(self.level._ph),
                                            C.byref(self._mp), self.nlev, nbox, arr)
        if not self.h:
            self.level.close()
            raise ValueError("boxes misaligned, overlapping or not properly nested")
        self.level.set(F_ACOEF, np.zeros((ny0, nx0)))
        lib().or_amrm_model_gap_solver_layout(self.h, max_box, nthreads)

    def field(self, l, k, fid):
        lo0, lo1, hi0, hi1 = self.boxes[l][k]
        nx, ny = hi0 - lo0 + 1, hi1 - lo1 + 1
        p = lib().or_amrm_model_field(self.h, l, k, fid)
        shape = {OM_QWX: (ny, nx + 1), OM_QWY: (ny + 1, nx)}.get(fid, (ny + 2, nx + 2))
        return np.ctypeslib.as_array(p, shape=shape)

    def set_state(self, l, k, f):
        for key, fid in (("head", OM_H), ("B", OM_B), ("Pi", OM_PI), ("zb", OM_ZB), ("mask", OM_MASK)):
            self.field(l, k, fid)[:] = f[key]

    def set_states(self, sts):
        """sts as suhmo_amd.synthetic.shmip_amrm_states returns it"""
        for l, bl in enumerate(sts):
            for k, st in enumerate(bl):
                self.set_state(l, k, st)

    def moulin_source(self, positions, sigma, flux, time_factor=1.0):
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        sg, fl = np.ascontiguousarray(sigma, dtype=np.float64), np.ascontiguousarray(flux, dtype=np.float64)
        integ = np.zeros(sg.size)
        lib().or_amrm_model_moulin_source(self.h, sg.size, _dp(pos), _dp(sg), _dp(fl), float(time_factor), _dp(integ))
        return integ

    def timestep(self, dt):
        pi, nv = C.c_int(), C.c_int()
        rc = lib().or_amrm_model_timestep(self.h, dt, C.byref(pi), C.byref(nv))
        if rc:
            raise RuntimeError("AMR time step failed (rc %d)" % rc)
        return pi.value, nv.value

    def close(self):
        if self.h:
            lib().or_amrm_model_destroy(self.h)
            self.h = None
            self.level.close()
