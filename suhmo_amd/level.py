"""Host-side handle on one device-resident level: a thin Python mirror of the operator
interface the reference exposes through VCAMRNonLinearPoissonOp / AMRNonLinearPoissonOp
(src/VCAMRNonLinearPoissonOp.H:50-141, src/AMRNonLinearPoissonOp.H:138-460), calling the
C-ABI one level at a time.  Method names follow the reference's (levelGSRB -> gsrb /
relax, applyOpI -> apply_op, residualI -> residual, restrictResidual, restrictR,
prolongIncrement, UpdateOperator, AverageOperator, AMRNorm -> norm)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import check

F_PHI, F_RHS, F_ACOEF, F_B, F_PI, F_ZB, F_MASK, F_BX, F_BY, F_LAMBDA, F_RES, F_LPHI, F_NL, F_DNL, \
    F_PHIOLD, F_CORR, F_GRADX, F_GRADY, F_RE, F_MR, F_PW, F_QWX, F_QWY, F_HLAG, F_CD, F_RHS0, F_MSRC, F_DCX, F_DCY, F_DTERM, F_ZS, F_COVER, F_PHI2 = range(33)


def _phys(p):
    return capi.Phys(p["A"], p["omega"], p["nu"], p["cutOffbr"], p["maxOffbr"], p.get("rho_w_g", 9800.0),
                     p.get("grav", 9.8), int(p.get("cutOffB", 0)), int(p.get("use_NL", 1)),
                     int(p.get("use_mask_gradients", 0)))


def _bc(bc):
    b = capi.BC()
    for d in range(2):
        for s in range(2):
            b.type[d][s] = int(bc["type"][d][s])
            b.value[d][s] = float(bc["value"][d][s])
        b.periodic[d] = int(bc["periodic"][d])
    return b


def solver_params(sp):
    return capi.SolverParams(sp.get("num_smooth", 4), sp.get("num_bottom", 16), sp.get("max_iter", 100),
                             sp.get("iter_min", 2), sp.get("imin", 5), sp.get("eps", 1e-7), sp.get("hang", 0.01),
                             sp.get("norm_thresh", 1e-7), int(sp.get("bcoeff_otf", 1)), sp.get("max_depth", -1))


class HipLevel:
    """One AMR level (or this rank's strip of rows of it) resident in HBM."""

    def __init__(self, nx, ny, dx, dy, bc, phys, alpha=0.0, beta=-1.0, max_box=64, boxes=None,
                 j0=0, ny_global=None, device=0, halo_rows=1, stream=None, i0=0, nx_global=0, patch_j0=0, patch_ny=0):
        self.nx, self.ny, self.dx, self.dy = nx, ny, dx, dy
        self.j0, self.ny_global = j0, (ny if ny_global is None else ny_global)
        self.stream = C.c_void_p(stream) if stream else C.c_void_p(0)
        d = capi.LevelDesc()
        d.nx, d.ny, d.j0, d.ny_global, d.dx, d.dy = nx, ny, j0, self.ny_global, dx, dy
        self._boxes = None
        if boxes is not None:
            self._boxes = (C.c_int * (4 * len(boxes)))(*[int(x) for b in boxes for x in b])
            d.nbox, d.boxes = len(boxes), C.cast(self._boxes, C.POINTER(C.c_int))
        else:
            d.nbox, d.boxes = 0, None
        d.max_box, d.alpha, d.beta = max_box, alpha, beta
        d.bc, d.phys, d.device, d.halo_rows = _bc(bc), _phys(phys), device, halo_rows
        d.i0, d.nx_global = i0, nx_global            # AMR patch (see HipAmr2)
        d.patch_j0, d.patch_ny = patch_j0, patch_ny  # ... cut into rank strips
        self._desc = d
        h = C.c_void_p()
        check(capi.lib().suhmo_level_create(C.byref(h), C.byref(d)))
        self.h = h
        self.ndepth = capi.lib().suhmo_level_num_depths(h)
        self._hooks = None

    def close(self):
        if getattr(self, "h", None):
            capi.lib().suhmo_level_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data movement
    def shape(self, field, depth=0, ghosted=False):
        nx, ny = self.nx >> depth, self.ny >> depth
        if field in (F_BX, F_QWX, F_DCX):
            return (ny, nx + 1)
        if field in (F_BY, F_QWY, F_DCY):
            return (ny + 1, nx)
        return (ny + 2, nx + 2) if ghosted else (ny, nx)

    def set(self, field, arr, depth=0, ghosted=False):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.shape(field, depth, ghosted), (a.shape, self.shape(field, depth, ghosted))
        check(capi.lib().suhmo_level_set_field(self.h, depth, field, a.ctypes.data, int(ghosted), 0, self.stream))

    def get(self, field, depth=0, ghosted=False):
        out = np.zeros(self.shape(field, depth, ghosted), dtype=np.float64)
        check(capi.lib().suhmo_level_get_field(self.h, depth, field, out.ctypes.data, int(ghosted), 0, self.stream))
        return out

    def set_device(self, field, dev_ptr, depth=0, ghosted=False):
        check(capi.lib().suhmo_level_set_field(self.h, depth, field, C.c_void_p(dev_ptr), int(ghosted), 1, self.stream))

    def put_box(self, field, ibox, fab, lo, hi, depth=0, with_domain_ghosts=False):
        a = np.ascontiguousarray(fab, dtype=np.float64)
        check(capi.lib().suhmo_level_put_box(self.h, depth, field, ibox, a.ctypes.data_as(C.POINTER(C.c_double)),
                                             lo[0], lo[1], hi[0], hi[1], int(with_domain_ghosts), self.stream))

    def get_box(self, field, ibox, lo, hi, depth=0):
        out = np.zeros((hi[1] - lo[1] + 1, hi[0] - lo[0] + 1))
        check(capi.lib().suhmo_level_get_box(self.h, depth, field, ibox, out.ctypes.data_as(C.POINTER(C.c_double)),
                                             lo[0], lo[1], hi[0], hi[1], self.stream))
        return out

    def set_inputs(self, f):
        self.set(F_PHI, f["phi"])
        self.set(F_RHS, f["rhs"])
        self.set(F_ACOEF, f["aCoef"])
        for k, fid in (("B", F_B), ("Pi", F_PI), ("zb", F_ZB), ("mask", F_MASK)):
            self.set(fid, f[k], ghosted=True)
        if "bx" in f:
            self.set(F_BX, f["bx"])
            self.set(F_BY, f["by"])

    # ---- operator methods (reference names in the module docstring)
    def _call(self, name, *args):
        check(getattr(capi.lib(), "suhmo_level_" + name)(self.h, *args, self.stream))

    def gsrb(self, sweeps=1, depth=0): self._call("gsrb", depth, sweeps)
    relax = gsrb
    def apply_op(self, homogeneous=False, depth=0): self._call("apply_op", depth, int(homogeneous))
    def residual(self, depth=0): self._call("residual", depth)
    def restrict_residual(self, depth=0): self._call("restrict_residual", depth)
    def restrict_r(self, depth=0): self._call("restrict_r", depth)
    def prolong_increment(self, depth=0): self._call("prolong_increment", depth)
    def prolong_bilinear(self, depth=0): self._call("prolong_bilinear", depth)
    def update_operator(self, depth=0): self._call("update_operator", depth)
    def average_operator(self, depth): self._call("average_operator", depth)
    def build_mg_coefficients(self): check(capi.lib().suhmo_level_build_mg_coefficients(self.h, self.stream))
    def nonlinear(self, depth=0): self._call("nonlinear", depth)
    def compute_lambda(self, depth=0): self._call("compute_lambda", depth)
    def fill_ghosts(self, field, homogeneous=False, depth=0): self._call("fill_ghosts", depth, field, int(homogeneous))
    def divergence(self, dst_field, depth=0): self._call("divergence", depth, dst_field)
    def axby(self, dst, x, y, a, b, depth=0): self._call("axby", depth, dst, x, y, float(a), float(b))
    def set_value(self, field, v, depth=0): self._call("set_value", depth, field, float(v))

    def get_flux(self, direction, ref=1, depth=0):
        out = np.zeros(self.shape(F_BX if direction == 0 else F_BY, depth))
        check(capi.lib().suhmo_level_get_flux(self.h, depth, direction, ref,
                                              out.ctypes.data_as(C.POINTER(C.c_double)), self.stream))
        return out

    def norm(self, field, ord=0, depth=0):
        r = C.c_double()
        check(capi.lib().suhmo_level_norm(self.h, depth, field, ord, C.byref(r), self.stream))
        return r.value

    def dot(self, x, y, depth=0):
        """dotProduct (src/AMRNonLinearPoissonOp.cpp:519-551) over the valid cells, over all ranks of a strip partition"""
        r = C.c_double()
        check(capi.lib().suhmo_level_dot(self.h, depth, x, y, C.byref(r), self.stream))
        return r.value

    def vcycle(self, sp):
        s = solver_params(sp)
        check(capi.lib().suhmo_level_vcycle(self.h, C.byref(s), self.stream))

    def solve(self, sp):
        s = solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = C.c_int()
        check(capi.lib().suhmo_level_solve(self.h, C.byref(s), C.byref(n), hist.ctypes.data_as(C.POINTER(C.c_double)),
                                           self.stream))
        return n.value, hist[: n.value + 1]

    def synchronize(self):
        check(capi.lib().suhmo_level_synchronize(self.h, self.stream))

    def set_option(self, key, value):
        """kernel selection (suhmo_level_set_option): e.g. set_option("gsrb_tile", 0)"""
        check(capi.lib().suhmo_level_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_long()
        check(capi.lib().suhmo_level_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def rccl_exchanges(self):
        """halo message groups this strip has sent so far (native transport), or the calls of the Python exchanger"""
        n = capi.lib().suhmo_level_rccl_exchanges(self.h)
        m = capi.lib().suhmo_level_ipc_exchanges(self.h)           # peer-direct halo messages (suhmo_ipc.hip); RCCL then carries the all-gathers only
        if n >= 0 or m >= 0:
            return max(n, 0) + max(m, 0)
        ex = getattr(self, "_exchanger", None)
        return getattr(ex, "calls", 0)

    # ---- profiling of the relax kernel (HIP events on the launch stream)
    def profile(self, on=True):
        check(capi.lib().suhmo_level_profile_reset(self.h))
        check(capi.lib().suhmo_level_profile_enable(self.h, int(on)))

    def profile_read(self, restricting=False):
        """(ms, launches, cell-sweeps) of the plain depth-0 GSRB launches, or of those that also restrict"""
        ms, n, c = C.c_double(), C.c_long(), C.c_long()
        f = capi.lib().suhmo_level_profile_read_restricting if restricting else capi.lib().suhmo_level_profile_read
        check(f(self.h, self.stream, C.byref(ms), C.byref(n), C.byref(c)))
        return ms.value, n.value, c.value


class HipAmr2:
    """Base level + one fine patch (coarse cells ci0..ci1 x cj0..cj1, refined by 2), both resident in HBM; the
    mirror of what AMRFASMultiGrid drives through AMRNonLinearPoissonOp::{relaxNF, AMRResidual, AMRRestrictS,
    AMRProlongS_2, reflux} (src/AMRNonLinearPoissonOp.cpp:690-704, 889-1206)."""

    def __init__(self, nxc, nyc, dxc, dyc, bc, phys, patch, alpha=0.0, beta=-1.0, max_box=64, device=0):
        ci0, cj0, ci1, cj1 = [int(v) for v in patch]
        self.coarse = HipLevel(nxc, nyc, dxc, dyc, bc, phys, alpha, beta, max_box, device=device)
        self.fine = HipLevel(2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1), dxc / 2.0, dyc / 2.0, bc, phys, alpha, beta, max_box,
                             j0=2 * cj0, ny_global=2 * nyc, i0=2 * ci0, nx_global=2 * nxc, device=device)
        self.stream = self.coarse.stream

    def _call(self, name, *args):
        check(getattr(capi.lib(), "suhmo_amr2_" + name)(self.coarse.h, self.fine.h, *args, self.stream))

    def cf_interp(self, field_f=F_PHI, field_c=F_PHI): self._call("cf_interp", field_f, field_c)
    def average(self, field_f=F_PHI, field_c=F_PHI): self._call("average", field_f, field_c)
    def fine_update_operator(self): self._call("fine_update_operator")

    def residual(self):
        r = C.c_double()
        self._call("residual", r if False else C.cast(C.pointer(r), C.POINTER(C.c_double)))
        return r.value

    def vcycle(self, sp):
        s = solver_params(sp)
        self._call("vcycle", C.byref(s))

    def solve(self, sp):
        s = solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = C.c_int()
        self._call("solve", C.byref(s), C.byref(n), hist.ctypes.data_as(C.POINTER(C.c_double)))
        return n.value, hist[: n.value + 1]

    def close(self):
        self.fine.close()
        self.coarse.close()


class HipAmr:
    """Base level + nested patches (patches[k] = box of level k+1 in the index space of level k), all resident in
    HBM: suhmo_amr_vcycle / suhmo_amr_solve over the array of level handles."""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, patches, alpha=0.0, beta=-1.0, max_box=64, device=0):
        self.levels = [HipLevel(nx0, ny0, dx0, dy0, bc, phys, alpha, beta, max_box, device=device)]
        nxg, nyg, dx, dy = nx0, ny0, dx0, dy0
        for (ci0, cj0, ci1, cj1) in patches:
            nxg, nyg, dx, dy = 2 * nxg, 2 * nyg, dx / 2.0, dy / 2.0
            self.levels.append(HipLevel(2 * (ci1 - ci0 + 1), 2 * (cj1 - cj0 + 1), dx, dy, bc, phys, alpha, beta, max_box,
                                        j0=2 * cj0, ny_global=nyg, i0=2 * ci0, nx_global=nxg, device=device))
        self.stream = self.levels[0].stream
        self._arr = (C.c_void_p * len(self.levels))(*[lv.h for lv in self.levels])

    def residual(self):
        r = C.c_double()
        check(capi.lib().suhmo_amr_residual(self._arr, len(self.levels), C.cast(C.pointer(r), C.POINTER(C.c_double)), self.stream))
        return r.value

    def vcycle(self, sp):
        s = solver_params(sp)
        check(capi.lib().suhmo_amr_vcycle(self._arr, len(self.levels), C.byref(s), self.stream))

    def solve(self, sp):
        s = solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = C.c_int()
        check(capi.lib().suhmo_amr_solve(self._arr, len(self.levels), C.byref(s), C.byref(n),
                                         hist.ctypes.data_as(C.POINTER(C.c_double)), self.stream))
        return n.value, hist[: n.value + 1]

    def close(self):
        for lv in reversed(self.levels):
            lv.close()


class _BoxView(HipLevel):
    """A box of a hierarchy: the level handle belongs to the hierarchy (never destroyed from here)."""

    def __init__(self, h, nx, ny, dx, dy, stream):
        self.h, self.nx, self.ny, self.dx, self.dy, self.stream = C.c_void_p(h), nx, ny, dx, dy, stream
        self.ndepth = capi.lib().suhmo_level_num_depths(self.h)
        self._hooks = None

    def close(self):
        self.h = None


class HipHier:
    """Base level + levels that are unions of boxes (boxes[l-1] = list of (lo0, lo1, hi0, hi1) in the index space of
    level l), the reference's DisjointBoxLayout per AMR level: suhmo_hier_* (suhmo_amd/csrc/suhmo_hier.hip)."""

    def __init__(self, nx0, ny0, dx0, dy0, bc, phys, boxes, alpha=0.0, beta=-1.0, max_box=64, device=0, j0=0, ny_global=None, halo_rows=1, options=None):
        """ny0 rows of level 0 starting at row j0 of ny_global: this rank's strip (one process per GPU; the boxes of the finer
        levels are given whole on every rank); default: the whole level.  options: "key=value,..." of suhmo_hier_create_opts (shadow, push_ghosts)"""
        self.boxes = [[tuple(int(v) for v in b) for b in bl] for bl in boxes]
        self.nlev = 1 + len(self.boxes)
        d = capi.LevelDesc()
        d.nx, d.ny, d.j0, d.ny_global, d.dx, d.dy = nx0, ny0, j0, (ny0 if ny_global is None else ny_global), dx0, dy0
        d.nbox, d.boxes, d.max_box, d.alpha, d.beta = 0, None, max_box, alpha, beta
        d.bc, d.phys, d.device, d.halo_rows = _bc(bc), _phys(phys), device, halo_rows
        nbox = (C.c_int * self.nlev)(0, *[len(bl) for bl in self.boxes])
        flat = [v for bl in self.boxes for b in bl for v in b]
        arr = (C.c_int * max(len(flat), 1))(*flat)
        h = C.c_void_p()
        check(capi.lib().suhmo_hier_create_opts(C.byref(h), C.byref(d), self.nlev, nbox, arr, options.encode() if options else None))
        self.h = h
        self.j0, self.ny_global = j0, int(d.ny_global)
        self.stream = C.c_void_p(0)
        self.level = [[_BoxView(capi.lib().suhmo_hier_box(h, 0, 0), nx0, ny0, dx0, dy0, self.stream)]]
        self.level[0][0].j0, self.level[0][0].ny_global = j0, self.ny_global          # (this rank's strip of level 0)
        for l, bl in enumerate(self.boxes, start=1):
            self.level.append([_BoxView(capi.lib().suhmo_hier_box(h, l, k), b[2] - b[0] + 1, b[3] - b[1] + 1, dx0 / 2 ** l, dy0 / 2 ** l,
                                        self.stream) for k, b in enumerate(bl)])
        self.coarse = self.level[0][0]
        # levels dealt to the ranks (option partition_min_cells): owner[l][k] = the rank that owns box k (-1: every rank), held[l][k]: this
        # rank keeps storage for it (its own boxes and mirrors of neighbours'); a box that is not held is a stub without fields
        self.owner, self.held = [[-1]], [[True]]
        for l, bl in enumerate(self.boxes, start=1):
            ow, he = [], []
            for k in range(len(bl)):
                hf = C.c_int()
                ow.append(int(capi.lib().suhmo_hier_box_owner(h, l, k, C.byref(hf))))
                he.append(bool(hf.value))
            self.owner.append(ow); self.held.append(he)
        self.rank = (j0 // ny0) if ny_global is not None and ny_global != ny0 else 0

    def owns(self, l, k):
        """this rank computes box k of level l (every box of a replicated level; on a level dealt to the ranks: its own boxes)"""
        return self.owner[l][k] in (-1, self.rank)

    def set_inputs(self, fs):
        """fs as suhmo_amd.synthetic.amrm_fields returns it"""
        self.coarse.set_inputs(fs[0])
        self.coarse.build_mg_coefficients()
        for l in range(1, self.nlev):
            for k, f in enumerate(fs[l]):
                if self.held[l][k]:
                    self.level[l][k].set_inputs(f)

    def level_array(self, l, field):
        nx, ny = self.coarse.nx << l, self.ny_global << l
        out = np.full((ny, nx), np.nan)
        for k, (lo0, lo1, hi0, hi1) in enumerate(self.boxes[l - 1]):
            out[lo1:hi1 + 1, lo0:hi0 + 1] = self.level[l][k].get(field)
        return out

    def gathers(self):
        return int(capi.lib().suhmo_hier_gathers(self.h))

    def get_option(self, key):
        v = C.c_long()
        check(capi.lib().suhmo_hier_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, key, value): check(capi.lib().suhmo_hier_set_option(self.h, key.encode(), int(value)))

    def exchange(self, l, field, corners=False): check(capi.lib().suhmo_hier_exchange(self.h, l, field, int(corners), self.stream))
    def cf_interp(self, l, field_f=F_PHI, field_c=F_PHI): check(capi.lib().suhmo_hier_cf_interp(self.h, l, field_f, field_c, self.stream))
    def pwl_fill(self, l, field_f, field_c): check(capi.lib().suhmo_hier_pwl_fill(self.h, l, field_f, field_c, self.stream))
    def average(self, l, field_f, field_c): check(capi.lib().suhmo_hier_average(self.h, l, field_f, field_c, self.stream))
    def gsrb(self, l, sweeps): check(capi.lib().suhmo_hier_gsrb(self.h, l, sweeps, self.stream))
    def update_operator(self, l): check(capi.lib().suhmo_hier_update_operator(self.h, l, self.stream))

    def residual(self):
        r = C.c_double()
        check(capi.lib().suhmo_hier_residual(self.h, C.cast(C.pointer(r), C.POINTER(C.c_double)), self.stream))
        return r.value

    def vcycle(self, sp):
        s = solver_params(sp)
        check(capi.lib().suhmo_hier_vcycle(self.h, C.byref(s), self.stream))

    def solve(self, sp):
        s = solver_params(sp)
        hist = np.zeros(s.max_iter + 2)
        n = C.c_int()
        check(capi.lib().suhmo_hier_solve(self.h, C.byref(s), C.byref(n), hist.ctypes.data_as(C.POINTER(C.c_double)), self.stream))
        return n.value, hist[: n.value + 1]

    def close(self):
        if getattr(self, "h", None):
            capi.lib().suhmo_hier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
