// suhmo_amr.hip -- two AMR levels: a base level (with its multigrid depths) and ONE rectangular fine patch
// refined by 2 (cfg3 of BASELINE.json: exec/0_convergence_channelized/2lev_base, fixed refined box).
//
// From the reference's own source:
//   relaxNF / AMRResidualNF / AMROperator / AMRRestrictS / AMRProlongS_2 / AMRNorm
//                                    src/AMRNonLinearPoissonOp.cpp:690-704, 889-1069, 1143-1264
//   reflux + getFlux                 src/VCAMRNonLinearPoissonOp.cpp:555-652, 792-841
//   UpdateOperator / WFlx_level with a coarser level   src/VCAMRNonLinearPoissonOp.cpp:34-64, src/AmrHydro.cpp:1455-1488
// Restated from upstream Chombo's documented semantics because the fork is not vendored (UNPINNED, SURVEY.md
// Appendix E): QuadCFInterp, LevelFluxRegister, FORT_AVERAGE, the ghosted coarse copy of AMRProlongS_2 and the AMR
// FAS cycle order.  The oracle's amr2.c states the same arithmetic; parity is bitwise.
//
// The fine level is an ordinary suhmo_level created with desc.i0/nx_global/j0/ny_global: sides of its rectangle
// inside the domain are coarse-fine sides whose ghost cells are STORED (DV::cfx / DV::ext) and filled here.
#include "suhmo_common.h"

namespace {
struct Pair { suhmo_level *C, *F; };

int check_pair(suhmo_level *C, suhmo_level *F)
{
    ARG(C && F);
    const DV &c = C->d[0].v, &f = F->d[0].v;
    if (f.nxg != 2 * c.nxg || f.nyg != 2 * c.nyg || (f.i0 & 1) || (f.j0 & 1) || (f.nx & 1) || (f.ny & 1)) {
        suhmo_set_error("amr: the fine level must be a coarse-aligned patch of the domain refined by 2"); return -1; }
    // proper nesting: 2 coarse cells between the fine patch and the edge of a coarse PATCH (the tangential stencil of
    // the coarse-fine interpolation and the reflux cell live there); nothing required towards a domain boundary
    {
        const int lo[2] = {f.i0 / 2, f.j0 / 2}, hi[2] = {(f.i0 + f.nx) / 2 - 1, (f.j0 + f.ny) / 2 - 1};
        const int clo[2] = {c.i0, c.j0}, chi[2] = {c.i0 + c.nx - 1, c.j0 + c.ny - 1}, dom[2] = {c.nxg, c.nyg};
        for (int d = 0; d < 2; d++) {
            bool ok_lo = (clo[d] == 0) ? lo[d] >= 0 : lo[d] - clo[d] >= 2;      // coarse edge on the domain boundary: no margin needed
            bool ok_hi = (chi[d] == dom[d] - 1) ? hi[d] <= chi[d] : chi[d] - hi[d] >= 2;
            if (d == 1) {
                // rank strips: a rank boundary of the fine strip coincides with one of the coarse strip (same physical rows);
                // a coarse-fine end must keep its coarse neighbour row on this rank (the reflux target lives there)
                if (f.rk[0]) ok_lo = lo[1] == clo[1] && c.rk[0]; else if (c.rk[0]) ok_lo = lo[1] - clo[1] >= 1;
                if (f.rk[1]) ok_hi = hi[1] == chi[1] && c.rk[1]; else if (c.rk[1]) ok_hi = chi[1] - hi[1] >= 1;
            }
            if (!ok_lo || !ok_hi) { suhmo_set_error("amr: the fine patch is not properly nested in its coarse level (2 cells)"); return -1; }
        }
    }
    if (C->device != F->device) { suhmo_set_error("amr2: both levels on one device"); return -1; }
    return 0;
}

// coarse value at GLOBAL coarse cell (I, J), periodic wrap
__device__ __forceinline__ double cval(const DV &vc, const double *__restrict__ c, int I, int J)
{
    if (vc.per[0]) { if (I < 0) I += vc.nxg; else if (I >= vc.nxg) I -= vc.nxg; }
    if (vc.per[1]) { if (J < 0) J += vc.nyg; else if (J >= vc.nyg) J -= vc.nyg; }
    return c[cidx(vc, I - vc.i0, J - vc.j0)];
}

// [Chombo] QuadCFInterp::coarseFineInterp, ratio 2 (oracle/amr2.c:cf_interp): tangential quadratic on the coarse
// level (one-sided next to a non-periodic domain boundary), then a normal quadratic through that value and the two
// fine cells inside: ghost = 8/15 phistar + 2/3 near - 1/5 far.  One thread per coarse-fine ghost cell.
__global__ void k_cf_interp(DV vf, double *__restrict__ f, DV vc, const double *__restrict__ c)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int dir, side, tt;
    if (t < 2 * vf.ny) { dir = 0; side = t / vf.ny; tt = t % vf.ny; if (!vf.cfx[side]) return; }
    else { t -= 2 * vf.ny; if (t >= 2 * vf.nx) return; dir = 1; side = t / vf.nx; tt = t % vf.nx; if (!vf.ext[side] || vf.rk[side]) return; }
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    const int tdir = 1 - dir;
    int gl = dir == 0 ? (side ? vf.nx : -1) : (side ? vf.ny : -1);            // local normal index of the ghost cell
    int inward = side ? -1 : 1;
    int g = gl + (dir == 0 ? vf.i0 : vf.j0), tg = tt + (dir == 0 ? vf.j0 : vf.i0);   // global fine indices
    int icn = g >> 1, ict = tg >> 1;
    double xt = (tg & 1) ? 0.25 : -0.25;
    int per = vc.per[tdir], nct = tdir == 0 ? vc.nxg : vc.nyg;
    bool have_lo = per || ict - 1 >= 0, have_hi = per || ict + 1 <= nct - 1;
#define CV(o) (dir == 0 ? cval(vc, c, icn, ict + (o)) : cval(vc, c, ict + (o), icn))
    double c0 = CV(0), d1 = 0.0, d2 = 0.0;
    if (have_lo && have_hi) { double cm = CV(-1), cp = CV(1); d1 = 0.5 * (cp - cm); d2 = cp - 2.0 * c0 + cm; }
    else if (have_hi) { double cp = CV(1), cpp = CV(2); d1 = 0.5 * (-3.0 * c0 + 4.0 * cp - cpp); d2 = c0 - 2.0 * cp + cpp; }
    else if (have_lo) { double cm = CV(-1), cmm = CV(-2); d1 = 0.5 * (3.0 * c0 - 4.0 * cm + cmm); d2 = c0 - 2.0 * cm + cmm; }
#undef CV
    double phistar = c0 + xt * d1 + (0.5 * xt * xt) * d2;
    int ig = dir == 0 ? gl : tt, jg = dir == 0 ? tt : gl;
    int idx = cidx(vf, ig, jg), step = dir == 0 ? inward : inward * vf.P;
    f[idx] = c_s * phistar + c_b * f[idx + step] + c_a * f[idx + 2 * step];
}

// [Chombo] FORT_AVERAGE: covered coarse cell = (sum of its 4 fine cells, i fastest) * 1/4
__global__ void k_amr_average(DV vf, const double *__restrict__ f, DV vc, double *__restrict__ c)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vf.nx / 2 || J >= vf.ny / 2) return;
    int b = cidx(vf, 2 * I, 2 * J);
    double s = 0.0;
    s = s + f[b]; s = s + f[b + 1]; s = s + f[b + vf.P]; s = s + f[b + vf.P + 1];
    c[cidx(vc, I + vf.i0 / 2 - vc.i0, J + vf.j0 / 2 - vc.j0)] = s * 0.25;
}
__global__ void k_amr_set_covered(DV vf, DV vc, double *__restrict__ c, double val)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vf.nx / 2 || J >= vf.ny / 2) return;
    c[cidx(vc, I + vf.i0 / 2 - vc.i0, J + vf.j0 / 2 - vc.j0)] = val;
}

// [Chombo] LevelFluxRegister (oracle/amr2.c:reflux): on every coarse-fine face the coarse flux is replaced by the
// average of the two fine fluxes; L(phi) of the coarse cell outside the patch += sign * reg / (dx dy).
// Fluxes as VCAMRNonLinearPoissonOp::getFlux (:792-841).  One thread per coarse face of the patch boundary.
// residual != 0: lofphi is the composite residual, already rhs - L(phi) everywhere; the cell gets rhs - (L(phi) + register), what
// copy -> reflux -> axby(-1, 1) leaves there
__global__ void k_amr_reflux(DV vf, FP ff, DV vc, FP fc, double *__restrict__ lofphi, int residual = 0)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncx = vf.nx / 2, ncy = vf.ny / 2, ci0 = vf.i0 / 2, cj0 = vf.j0 / 2;
    int dir, side, T;
    if (t < 2 * ncy) { dir = 0; side = t / ncy; T = cj0 + t % ncy; }
    else { t -= 2 * ncy; if (t >= 2 * ncx) return; dir = 1; side = t / ncx; T = ci0 + t % ncx; if (vf.rk[side]) return; }   // rank boundary: no coarse-fine face
    int F = dir == 0 ? (side == 0 ? ci0 : ci0 + ncx) : (side == 0 ? cj0 : cj0 + ncy);      // coarse face index
    int outside = side == 0 ? F - 1 : F, ndomc = dir == 0 ? vc.nxg : vc.nyg;
    if (outside < 0 || outside > ndomc - 1) return;              // patch side on the domain boundary
    const double rscale = 1.0 / (vc.dx * vc.dy);
    const double dxd = dir == 0 ? vc.dx : vc.dy, tsize = dir == 0 ? vc.dy : vc.dx;
    const double cs = vc.beta * 1 / dxd, fs = vc.beta * 2 / dxd;
    const double sign = side == 0 ? 1.0 : -1.0;
    const double *__restrict__ phic = fc.f[SUHMO_F_PHI], *__restrict__ phif = ff.f[SUHMO_F_PHI];
    double phihi, philo, bc_;
    if (dir == 0) { int idx = cidx(vc, F - vc.i0, T - vc.j0); phihi = phic[idx]; philo = phic[idx - 1]; bc_ = fc.f[SUHMO_F_BX][idx]; }
    else { int idx = cidx(vc, T - vc.i0, F - vc.j0); phihi = phic[idx]; philo = phic[idx - vc.P]; bc_ = fc.f[SUHMO_F_BY][idx]; }
    double Fc = -bc_ * ((phihi - philo) * cs);
    double reg = -(tsize * Fc);
    for (int k = 0; k < 2; k++) {
        int fi = (dir == 0 ? 2 * F : 2 * T + k) - vf.i0, fj = (dir == 0 ? 2 * T + k : 2 * F) - vf.j0;   // local fine face
        int idx = cidx(vf, fi, fj);
        double ph_hi = phif[idx], ph_lo = dir == 0 ? phif[idx - 1] : phif[idx - vf.P];
        double bf = dir == 0 ? ff.f[SUHMO_F_BX][idx] : ff.f[SUHMO_F_BY][idx];
        double Ff = -bf * ((ph_hi - ph_lo) * fs);
        reg = reg + (tsize * Ff) * 0.5;
    }
    int oidx = dir == 0 ? cidx(vc, outside - vc.i0, T - vc.j0) : cidx(vc, T - vc.i0, outside - vc.j0);
    if (residual) lofphi[oidx] = -1.0 * (fc.f[SUHMO_F_LPHI][oidx] + sign * rscale * reg) + 1.0 * fc.f[SUHMO_F_RHS][oidx];
    else lofphi[oidx] = lofphi[oidx] + sign * rscale * reg;
}

// PROLONG_2_NL (src/AMRNonLinearPoissonOpF.ChF:660-705) from the coarse LEVEL's correction canvas (its ghost ring
// holds the BC values, AMRProlongS_2 :1160-1166)
__global__ void k_amr_prolong2(DV vf, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= vf.nx || j >= vf.ny) return;
    const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
    int gi = i + vf.i0, gj = j + vf.j0;
    int ic = gi / 2, jc = gj / 2, o1 = 2 * (gi % 2) - 1, o2 = 2 * (gj % 2) - 1;
    int idx = cidx(vf, i, j), cc = cidx(vc, ic - vc.i0, jc - vc.j0);
    // the diagonal neighbour: a ghost cell of the coarsened patch box that lies outside the domain is filled by
    // m_bc only along the box's own extent -- the corner cell beyond it is never written (:1160-1172), value 0
    double cd = c[cc + o1 + o2 * vc.P];
    {
        const int I = ic + o1, J = jc + o2, ci0 = vf.i0 / 2, ci1 = ci0 + vf.nx / 2 - 1, cj0 = vf.j0 / 2, cj1 = cj0 + vf.ny / 2 - 1;
        const bool xout = !vc.per[0] && (I < 0 || I >= vc.nxg), yout = !vc.per[1] && (J < 0 || J >= vc.nyg);
        if ((xout && (J < cj0 || J > cj1)) || (yout && (I < ci0 || I > ci1))) cd = 0.0;
    }
    double p = phi[idx];
    p = p + fx2 * c[cc] + f0 * cd;
    p = p + fx1 * (c[cc + o1] + c[cc + o2 * vc.P]);
    phi[idx] = p;
}
// [Chombo] PiecewiseLinearFillPatch, ratio 2, one ghost layer incl. corners (oracle/amr_step.c:or_pwl_fill): coarse value
// + limited slopes (central, one-sided next to the domain boundary, FORT_INTERPLIMIT over the 3 x 3 neighbourhood) times
// the offset of the fine cell centre (+-1/4).  One thread per ghost cell of the ring; cells outside the domain stay.
__global__ void k_pwl_fill(DV vf, double *__restrict__ f, DV vc, const double *__restrict__ c)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int i, j;
    if (t < 2 * (vf.nx + 2)) { j = t < vf.nx + 2 ? -1 : vf.ny; i = t % (vf.nx + 2) - 1; }
    else { t -= 2 * (vf.nx + 2); if (t >= 2 * vf.ny) return; i = t < vf.ny ? -1 : vf.nx; j = t % vf.ny; }
    const int gi = i + vf.i0, gj = j + vf.j0;
    if (gi < 0 || gi >= vf.nxg || gj < 0 || gj >= vf.nyg) return;
    if ((j < 0 && vf.rk[0]) || (j >= vf.ny && vf.rk[1])) return;         // rank boundary: exchanged, not interpolated
    const int I = gi >> 1, J = gj >> 1, cc = cidx(vc, I - vc.i0, J - vc.j0);
    const double c0 = c[cc];
    double s0, s1;
    if (I - 1 >= 0 && I + 1 <= vc.nxg - 1) s0 = 0.5 * (c[cc + 1] - c[cc - 1]);
    else if (I - 1 < 0) s0 = c[cc + 1] - c0;
    else s0 = c0 - c[cc - 1];
    if (J - 1 >= 0 && J + 1 <= vc.nyg - 1) s1 = 0.5 * (c[cc + vc.P] - c[cc - vc.P]);
    else if (J - 1 < 0) s1 = c[cc + vc.P] - c0;
    else s1 = c0 - c[cc - vc.P];
    double smax = c0, smin = c0;
    for (int jj = -1; jj <= 1; jj++)
        for (int ii = -1; ii <= 1; ii++) {
            int In = I + ii, Jn = J + jj;
            if (In < 0 || In > vc.nxg - 1 || Jn < 0 || Jn > vc.nyg - 1) continue;
            double v = c[cc + ii + jj * vc.P];
            smax = fmax(smax, v); smin = fmin(smin, v);
        }
    const double deltasum = 0.5 * (fabs(s0) + fabs(s1));
    if (deltasum > 0.0) {
        double etamax = (smax - c0) / deltasum, etamin = (c0 - smin) / deltasum;
        double eta = fmax(fmin(fmin(etamin, etamax), 1.0), 0.0);
        s0 = eta * s0; s1 = eta * s1;
    }
    double v = c0;
    v = v + s0 * ((gi & 1) ? 0.25 : -0.25);
    v = v + s1 * ((gj & 1) ? 0.25 : -0.25);
    f[cidx(vf, i, j)] = v;
}
// PROLONGNL with the AMR refinement ratio (AMRProlong / AMRProlongS, src/AMRNonLinearPoissonOp.cpp:1073-1140)
__global__ void k_amr_prolong_pc(DV vf, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= vf.nx || j >= vf.ny) return;
    int idx = cidx(vf, i, j);
    phi[idx] = phi[idx] + c[cidx(vc, (i + vf.i0) / 2 - vc.i0, (j + vf.j0) / 2 - vc.j0)];
}
// [Chombo] CoarseAverageFace with the AMR ratio: coarse face under the patch = (sum of its 2 fine faces) / 2
// (VCAMRNonLinearPoissonOp::finerOperatorChanged :1356-1439; same summation as AverageOperator's k_average_faces)
__global__ void k_amr_average_faces(DV vf, const double *__restrict__ bxf, const double *__restrict__ byf, DV vc,
                                    double *__restrict__ bxc, double *__restrict__ byc)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    const int ncx = vf.nx / 2, ncy = vf.ny / 2;
    if (I > ncx || J > ncy) return;
    const int co = cidx(vc, I + vf.i0 / 2 - vc.i0, J + vf.j0 / 2 - vc.j0), fo = cidx(vf, 2 * I, 2 * J);
    if (J < ncy) { double sm = 0.0; sm = sm + bxf[fo]; sm = sm + bxf[fo + vf.P]; bxc[co] = sm / 2.0; }
    if (I < ncx) { double sm = 0.0; sm = sm + byf[fo]; sm = sm + byf[fo + 1]; byc[co] = sm / 2.0; }
}
}  // namespace

int suhmo_grad_cc(suhmo_level *L, int depth, hipStream_t st);          // suhmo_bcoef.hip: k_gradcc (+ k_grad_ghosts)
int suhmo_re_bcoef_unfused(suhmo_level *L, int depth, hipStream_t st);  // suhmo_bcoef.hip: k_grad_ghosts, k_re, k_bcoef_faces

extern "C" int suhmo_amr2_cf_interp(suhmo_level_t *C, suhmo_level_t *F, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(F->device));
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    double *pf = suhmo_field(F, 0, field_f), *pc = suhmo_field(C, 0, field_c);
    if (!pf || !pc) { suhmo_set_error("field allocation failed"); return -2; }
    if (field_c == SUHMO_F_PHI && (rc = suhmo_ensure_phi_halo(C, 0, 1, (hipStream_t)s))) return rc;   // rank strips: the stencil reaches
                                                                                                    // one coarse halo row
    int n = 2 * vf.ny + 2 * vf.nx;
    hipLaunchKernelGGL(k_cf_interp, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, vf, pf, vc, pc);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int suhmo_amr2_average(suhmo_level_t *C, suhmo_level_t *F, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(F->device));
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    double *pf = suhmo_field(F, 0, field_f), *pc = suhmo_field(C, 0, field_c);
    if (!pf || !pc) { suhmo_set_error("field allocation failed"); return -2; }
    if (field_c == SUHMO_F_PHI) C->d[0].phi_fresh = 0;
    hipLaunchKernelGGL(k_amr_average, dim3((vf.nx / 2 + 63) / 64, (vf.ny / 2 + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, vf, pf, vc, pc);
    HIPCHK(hipGetLastError());
    return 0;
}

// UpdateOperator of the fine level with a coarser level (src/VCAMRNonLinearPoissonOp.cpp:34-64 ->
// AmrHydro::WFlx_level :1415-1539): fine cell-centred gradient, coarse gradient + QuadCFInterp of both components
// into the fine coarse-fine ghosts, ExtrapGhostCells on domain sides, Re on the ghosted box, bCoef on the faces
extern "C" int suhmo_amr2_fine_update_operator(suhmo_level_t *C, suhmo_level_t *F, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    HIPCHK(hipSetDevice(F->device));
    hipStream_t st = (hipStream_t)s;
    if ((rc = suhmo_grad_cc(F, 0, st))) return rc;
    if ((rc = suhmo_grad_cc(C, 0, st))) return rc;                 // includes exchange (periodic) + ExtrapGhostCells
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_GRADX, SUHMO_F_GRADX, s))) return rc;
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_GRADY, SUHMO_F_GRADY, s))) return rc;
    return suhmo_re_bcoef_unfused(F, 0, st);
}

// composite residual: fine RES = rhs1 - L1(phi1) after coarseFineInterp (AMRResidualNF :922-939); coarse RES =
// rhs0 - [applyOpI(phi0) + reflux] (AMRResidual :889-903, AMROperator :942-967); covered coarse cells are zeroed
// and the max norm over both levels is returned (AMRNorm :1222-1264)
extern "C" int suhmo_amr2_residual(suhmo_level_t *C, suhmo_level_t *F, double *norm, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    HIPCHK(hipSetDevice(F->device));
    hipStream_t st = (hipStream_t)s;
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    if ((rc = suhmo_level_residual(F, 0, s))) return rc;
    if ((rc = suhmo_level_apply_op(C, 0, 0, s))) return rc;                       // LPHI = L0(phi0), kept
    double *res = suhmo_field(C, 0, SUHMO_F_RES), *lphi = suhmo_field(C, 0, SUHMO_F_LPHI);
    HIPCHK(hipMemcpyAsync(res, lphi, C->d[0].elems * sizeof(double), hipMemcpyDeviceToDevice, st));
    int n = vf.ny + vf.nx;                                                        // 2 * (ncy + ncx) coarse faces
    hipLaunchKernelGGL(k_amr_reflux, dim3((n + 255) / 256), dim3(256), 0, st, vf, F->d[0].fp, vc, C->d[0].fp, res);
    HIPCHK(hipGetLastError());
    if ((rc = suhmo_level_axby(C, 0, SUHMO_F_RES, SUHMO_F_RES, SUHMO_F_RHS, -1.0, 1.0, s))) return rc;
    hipLaunchKernelGGL(k_amr_set_covered, dim3((vf.nx / 2 + 63) / 64, (vf.ny / 2 + 3) / 4), dim3(64, 4), 0, st, vf, vc, res, 0.0);
    HIPCHK(hipGetLastError());
    if (norm) {
        double a = 0.0, b = 0.0;
        if ((rc = suhmo_level_norm(C, 0, SUHMO_F_RES, 0, &a, s))) return rc;
        if ((rc = suhmo_level_norm(F, 0, SUHMO_F_RES, 0, &b, s))) return rc;
        *norm = a > b ? a : b;
    }
    return 0;
}

// reflux (src/VCAMRNonLinearPoissonOp.cpp:555-652): coarse field_c (= L(phi) of the coarse level, e.g. LPHI after
// applyOpI) += the flux mismatch on the coarse-fine faces; the fine coarse-fine ghosts are interpolated first (:602)
extern "C" int suhmo_amr2_reflux(suhmo_level_t *C, suhmo_level_t *F, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_c >= 0 && field_c < SUHMO_F_COUNT && field_c != SUHMO_F_PHI && field_c != SUHMO_F_BX && field_c != SUHMO_F_BY);
    HIPCHK(hipSetDevice(F->device));
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    double *p = suhmo_field(C, 0, field_c);
    if (!p) { suhmo_set_error("field allocation failed"); return -2; }
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    int n = vf.ny + vf.nx;
    hipLaunchKernelGGL(k_amr_reflux, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, vf, F->d[0].fp, vc, C->d[0].fp, p);
    HIPCHK(hipGetLastError());
    return 0;
}
// PiecewiseLinearFillPatch::fillInterp of one field (the time loop's coarse-fine ghosts of b, mR, Re: src/AmrHydro.cpp:2373-2380,
// 2499-2507, 2711-2719): fine ghost ring of field_f <- coarse field_c
extern "C" int suhmo_amr2_pwl_fill(suhmo_level_t *C, suhmo_level_t *F, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    ARG(field_f != SUHMO_F_BX && field_f != SUHMO_F_BY && field_f != SUHMO_F_QWX && field_f != SUHMO_F_QWY && field_f != SUHMO_F_DCX && field_f != SUHMO_F_DCY);
    HIPCHK(hipSetDevice(F->device));
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    // rank strips: the 3 x 3 coarse neighbourhood reaches one halo row of the coarse strip (the caller exchanged field_c)
    double *pf = suhmo_field(F, 0, field_f), *pc = suhmo_field(C, 0, field_c);
    if (!pf || !pc) { suhmo_set_error("field allocation failed"); return -2; }
    if (field_f == SUHMO_F_PHI) F->d[0].phi_fresh = 0;
    int n = 2 * (vf.nx + 2) + 2 * vf.ny;
    hipLaunchKernelGGL(k_pwl_fill, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, vf, pf, vc, pc);
    HIPCHK(hipGetLastError());
    return 0;
}
// AMRProlong / AMRProlongS (:1073-1140): fine PHI += coarse field_c, piecewise constant
extern "C" int suhmo_amr2_prolong_pc(suhmo_level_t *C, suhmo_level_t *F, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_c >= 0 && field_c < SUHMO_F_COUNT && field_c != SUHMO_F_BX && field_c != SUHMO_F_BY);
    HIPCHK(hipSetDevice(F->device));
    Depth &DC = C->d[0], &DF = F->d[0];
    double *corr = suhmo_field(C, 0, field_c);
    if (!corr) { suhmo_set_error("field allocation failed"); return -2; }
    DF.phi_fresh = 0;
    hipLaunchKernelGGL(k_amr_prolong_pc, dim3((DF.v.nx + 63) / 64, (DF.v.ny + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, DF.v, DF.fp.f[SUHMO_F_PHI], DC.v, corr);
    HIPCHK(hipGetLastError());
    return 0;
}
// finerOperatorChanged with the AMR ratio (:1356-1439): the coarse level's aCoef, B, Pi, zb, iceMask and bCoef under the
// patch <- averages of the fine level's
extern "C" int suhmo_amr2_finer_operator_changed(suhmo_level_t *C, suhmo_level_t *F, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    HIPCHK(hipSetDevice(F->device));
    for (int f : {SUHMO_F_ACOEF, SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK})
        if ((rc = suhmo_amr2_average(C, F, f, f, s))) return rc;
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    hipLaunchKernelGGL(k_amr_average_faces, dim3((vf.nx / 2 + 1 + 63) / 64, (vf.ny / 2 + 1 + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, vf,
                       F->d[0].fp.f[SUHMO_F_BX], F->d[0].fp.f[SUHMO_F_BY], vc, C->d[0].fp.f[SUHMO_F_BX], C->d[0].fp.f[SUHMO_F_BY]);
    HIPCHK(hipGetLastError());
    C->coarse_mask_ok = 0;
    suhmo_level_drop_graphs(C);
    return 0;
}

// AMRProlongS_2 (:1143-1206): fine PHI += PROLONG_2_NL(coarse field_c), the coarse field first gets its physical-BC
// ghost ring (inhomogeneous in FAS mode :1163-1165)
extern "C" int suhmo_amr2_prolong2(suhmo_level_t *C, suhmo_level_t *F, int field_c, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_c >= 0 && field_c < SUHMO_F_COUNT && field_c != SUHMO_F_BX && field_c != SUHMO_F_BY);
    HIPCHK(hipSetDevice(F->device));
    Depth &DC = C->d[0], &DF = F->d[0];
    double *corr = suhmo_field(C, 0, field_c);
    if (!corr) { suhmo_set_error("field allocation failed"); return -2; }
    if ((rc = suhmo_level_exchange(C, 0, field_c, s))) return rc;         // rank strips: one halo row of the coarse correction
    if ((rc = suhmo_level_fill_ghosts(C, 0, field_c, 0, s))) return rc;
    DF.phi_fresh = 0;
    hipLaunchKernelGGL(k_amr_prolong2, dim3((DF.v.nx + 63) / 64, (DF.v.ny + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, DF.v, DF.fp.f[SUHMO_F_PHI], DC.v, corr);
    HIPCHK(hipGetLastError());
    return 0;
}
// coarse cells under the patch <- value (AMRNorm / zeroCovered :1222-1264)
extern "C" int suhmo_amr2_set_covered(suhmo_level_t *C, suhmo_level_t *F, int field_c, double value, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(F->device));
    const DV &vf = F->d[0].v, &vc = C->d[0].v;
    double *p = suhmo_field(C, 0, field_c);
    if (!p) { suhmo_set_error("field allocation failed"); return -2; }
    if (field_c == SUHMO_F_PHI) C->d[0].phi_fresh = 0;
    hipLaunchKernelGGL(k_amr_set_covered, dim3((vf.nx / 2 + 63) / 64, (vf.ny / 2 + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, vf, vc, p, value);
    HIPCHK(hipGetLastError());
    return 0;
}

// one AMR FAS V-cycle (VCycleAMR; same order as oracle/amr2.c:or_amr2_vcycle)
extern "C" int suhmo_amr2_vcycle(suhmo_level_t *C, suhmo_level_t *F, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    int rc = check_pair(C, F); if (rc) return rc;
    ARG(sp);
    HIPCHK(hipSetDevice(F->device));
    hipStream_t st = (hipStream_t)s;
    Depth &DC = C->d[0];
    const size_t cbytes = DC.elems * sizeof(double);
    double *rhs0 = suhmo_field(C, 0, SUHMO_F_RHS0), *phiold = suhmo_field(C, 0, SUHMO_F_PHIOLD), *corr = suhmo_field(C, 0, SUHMO_F_CORR);
    if (!rhs0 || !phiold || !corr) { suhmo_set_error("field allocation failed"); return -2; }
    // operator of the fine level from the current head
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    if (sp->bcoeff_otf && (rc = suhmo_amr2_fine_update_operator(C, F, s))) return rc;
    // relaxNF(phi1, phi0, rhs1, pre)
    if ((rc = suhmo_level_gsrb(F, 0, sp->num_smooth, s))) return rc;
    // AMRRestrictS(skip_res): phi0 under the patch <- average(phi1)
    if ((rc = suhmo_amr2_average(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    // residuals; covered coarse cells <- average(res1); FAS rhs of the base level = res0' + L0(phi0)
    if ((rc = suhmo_amr2_residual(C, F, nullptr, s))) return rc;
    if ((rc = suhmo_amr2_average(C, F, SUHMO_F_RES, SUHMO_F_RES, s))) return rc;
    HIPCHK(hipMemcpyAsync(rhs0, DC.fp.f[SUHMO_F_RHS], cbytes, hipMemcpyDeviceToDevice, st));
    if ((rc = suhmo_level_axby(C, 0, SUHMO_F_RHS, SUHMO_F_RES, SUHMO_F_LPHI, 1.0, 1.0, s))) return rc;
    HIPCHK(hipMemcpyAsync(phiold, DC.fp.f[SUHMO_F_PHI], cbytes, hipMemcpyDeviceToDevice, st));
    if ((rc = suhmo_level_vcycle(C, sp, s))) return rc;                          // MGCycle of the base level
    HIPCHK(hipMemcpyAsync(DC.fp.f[SUHMO_F_RHS], rhs0, cbytes, hipMemcpyDeviceToDevice, st));
    // AMRProlongS_2: phi1 += PROLONG_2_NL(phi0 - phi0_old), coarse correction with inhomogeneous-BC ghosts
    if ((rc = suhmo_level_axby(C, 0, SUHMO_F_CORR, SUHMO_F_PHI, SUHMO_F_PHIOLD, 1.0, -1.0, s))) return rc;
    if ((rc = suhmo_amr2_prolong2(C, F, SUHMO_F_CORR, s))) return rc;
    // relaxNF(phi1, phi0, rhs1, post)
    if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    return suhmo_level_gsrb(F, 0, sp->num_smooth, s);
}

// AMRMultiGrid::solveNoInit stopping rule on the composite residual norm
extern "C" int suhmo_amr2_solve(suhmo_level_t *C, suhmo_level_t *F, const suhmo_solver_params_t *sp, int *iters, double *hist, suhmo_stream_t s)
{
    ARG(sp);
    int rc;
    double rnorm = 0.0;
    if ((rc = suhmo_amr2_residual(C, F, &rnorm, s))) return rc;
    double initial_rnorm = rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    bool goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    bool goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        if ((rc = suhmo_amr2_vcycle(C, F, sp, s))) return rc;
        if ((rc = suhmo_amr2_residual(C, F, &rnorm, s))) return rc;
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    if (iters) *iters = iter;
    return 0;
}


// ================================================================ N nested levels
// levels[0] = base level, levels[l] = patch of level l (properly nested in level l-1): oracle/amrn.c states the same
// cycle.  The two-level entry points above are the nlev = 2 case.
// Rank strips: a rank holds of every level the rows of its own physical slab, so levels[l] may be NULL on a rank the
// patch of level l does not reach.  Such a rank still runs the coarse half of every pair it has a coarse strip of (the
// FAS right-hand side res' + L(phi) replaces rhs on the WHOLE coarse level) and every base-level collective.
int suhmo_apply_and_residual(suhmo_level *L, int depth, hipStream_t st);      // suhmo_ops.hip: LPHI and RES = rhs - LPHI in one pass
namespace {
// head of level l: coarse-fine ghosts from level l-1.  A rank with a strip of level l-1 but none of level l MIRRORS the
// halo demand the interpolation puts on level l-1, so that every rank of that level's communicator runs the same sequence
// of exchanges (they are demand-driven by Depth::phi_fresh, which must evolve identically on all of them).
int cf_phi(suhmo_level_t **lv, int l, suhmo_stream_t s)
{
    if (l == 0) return 0;
    if (lv[l]) return suhmo_amr2_cf_interp(lv[l - 1], lv[l], SUHMO_F_PHI, SUHMO_F_PHI, s);
    if (lv[l - 1]) return suhmo_ensure_phi_halo(lv[l - 1], 0, 1, (hipStream_t)s);
    return 0;
}
// RES of level l-1 = rhs - [applyOpI(phi) + reflux from level l]; LPHI of level l-1 keeps the plain L(phi)
int composite_residual(suhmo_level_t **lv, int l, suhmo_stream_t s)
{
    suhmo_level *C = lv[l - 1], *F = lv[l];
    if (!C) return 0;
    hipStream_t st = (hipStream_t)s;
    int rc;
    // one pass writes L(phi) and rhs - L(phi); the reflux kernel rewrites the cells next to the coarse-fine faces (as suhmo_hier.hip)
    if ((rc = cf_phi(lv, l - 1, s))) return rc;
    if ((rc = suhmo_apply_and_residual(C, 0, st))) return rc;
    double *res = suhmo_field(C, 0, SUHMO_F_RES);
    if ((rc = cf_phi(lv, l, s))) return rc;
    if (F) {
        const DV &vf = F->d[0].v, &vc = C->d[0].v;
        int n = vf.ny + vf.nx;
        hipLaunchKernelGGL(k_amr_reflux, dim3((n + 255) / 256), dim3(256), 0, st, vf, F->d[0].fp, vc, C->d[0].fp, res, 1);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
int vcycle_amr(suhmo_level_t **lv, int l, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    if (l == 0) return suhmo_level_vcycle(lv[0], sp, s);
    suhmo_level *C = lv[l - 1], *F = lv[l];
    hipStream_t st = (hipStream_t)s;
    int rc;
    if ((rc = cf_phi(lv, l, s))) return rc;
    if (sp->bcoeff_otf) {                                          // UpdateOperator of level l with its coarser level
        if ((rc = cf_phi(lv, l - 1, s))) return rc;                // the coarser level's own coarse-fine ghosts (its gradient reads them)
        if (F && (rc = suhmo_grad_cc(F, 0, st))) return rc;
        if (C && (rc = suhmo_grad_cc(C, 0, st))) return rc;        // every rank of the coarser level's communicator (gradient halo exchange)
        if (F) {
            if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_GRADX, SUHMO_F_GRADX, s))) return rc;
            if ((rc = suhmo_amr2_cf_interp(C, F, SUHMO_F_GRADY, SUHMO_F_GRADY, s))) return rc;
            if ((rc = suhmo_re_bcoef_unfused(F, 0, st))) return rc;
        }
    }
    if (F) {
        if ((rc = suhmo_level_gsrb(F, 0, sp->num_smooth, s))) return rc;                       // relaxNF
        if ((rc = suhmo_amr2_average(C, F, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;           // AMRRestrictS(skip_res)
    } else if (C) C->d[0].phi_fresh = 0;                                                       // phi of level l-1 changed on the other ranks
    if ((rc = cf_phi(lv, l, s))) return rc;
    if (F && (rc = suhmo_level_residual(F, 0, s))) return rc;                                  // res_l = rhs_l - L_l(phi_l)
    double *rhs0 = nullptr, *phiold = nullptr;
    size_t cbytes = 0;
    SwapGuard rhs_aside;
    if (C) {
        Depth &DC = C->d[0];
        cbytes = DC.elems * sizeof(double);
        rhs0 = suhmo_field(C, 0, SUHMO_F_RHS0); phiold = suhmo_field(C, 0, SUHMO_F_PHIOLD);
        if (!rhs0 || !phiold || !suhmo_field(C, 0, SUHMO_F_CORR)) { suhmo_set_error("field allocation failed"); return -2; }
        if ((rc = composite_residual(lv, l, s))) return rc;
        if (F && (rc = suhmo_amr2_average(C, F, SUHMO_F_RES, SUHMO_F_RES, s))) return rc;
        // the right-hand side of level l-1 is set aside while its FAS problem runs: two canvases trade places (the captured V-cycle
        // graphs of a level check which one they were recorded with)
        rhs_aside.arm(&DC.fp.f[SUHMO_F_RHS], &DC.fp.f[SUHMO_F_RHS0]);
        if ((rc = suhmo_level_axby(C, 0, SUHMO_F_RHS, SUHMO_F_RES, SUHMO_F_LPHI, 1.0, 1.0, s))) return rc;
        if ((rc = suhmo_level_exchange(C, 0, SUHMO_F_RHS, s))) return rc;                      // rank strips: rhs halo rows
        HIPCHK(hipMemcpyAsync(phiold, DC.fp.f[SUHMO_F_PHI], cbytes, hipMemcpyDeviceToDevice, st));
    }
    if ((rc = vcycle_amr(lv, l - 1, sp, s))) return rc;
    if (C) {
        rhs_aside.back();                                                                     // the halo rows never left
        if ((rc = suhmo_level_axby(C, 0, SUHMO_F_CORR, SUHMO_F_PHI, SUHMO_F_PHIOLD, 1.0, -1.0, s))) return rc;
        if (F) {
            if ((rc = suhmo_amr2_prolong2(C, F, SUHMO_F_CORR, s))) return rc;                  // AMRProlongS_2
        } else if ((rc = suhmo_level_exchange(C, 0, SUHMO_F_CORR, s))) return rc;              // the exchange inside prolong2, for the ranks that prolong
    }
    if ((rc = cf_phi(lv, l, s))) return rc;
    if (F) return suhmo_level_gsrb(F, 0, sp->num_smooth, s);
    return 0;
}
int check_hierarchy(suhmo_level_t **lv, int nlev)
{
    ARG(lv && nlev >= 1 && nlev <= 8 && lv[0]);
    const DV &b = lv[0]->d[0].v;
    if (b.i0 || b.nx != b.nxg) { suhmo_set_error("amr: level 0 must span the domain in x"); return -1; }
    for (int l = 1; l < nlev; l++) {
        if (lv[l] && !lv[l - 1]) { suhmo_set_error("amr: a rank holding a strip of level %d must hold one of level %d", l, l - 1); return -1; }
        if (lv[l]) { int rc = check_pair(lv[l - 1], lv[l]); if (rc) return rc; }
    }
    return 0;
}
}  // namespace
int suhmo_amr_check_hierarchy(suhmo_level_t **lv, int nlev) { return check_hierarchy(lv, nlev); }   // for suhmo_step.hip

extern "C" int suhmo_amr_residual(suhmo_level_t **lv, int nlev, double *norm, suhmo_stream_t s)
{
    int rc = check_hierarchy(lv, nlev); if (rc) return rc;
    HIPCHK(hipSetDevice(lv[0]->device));
    int top = nlev - 1;
    if ((rc = cf_phi(lv, top, s))) return rc;
    if (lv[top] && (rc = suhmo_level_residual(lv[top], 0, s))) return rc;                  // AMRResidualNF on the finest level
    for (int l = top; l >= 1; l--) if ((rc = composite_residual(lv, l, s))) return rc;
    for (int l = top; l >= 1; l--)
        if (lv[l] && (rc = suhmo_amr2_set_covered(lv[l - 1], lv[l], SUHMO_F_RES, 0.0, s))) return rc;   // AMRNorm
    if (norm) {
        double m = 0.0;
        for (int l = 0; l <= top; l++)
            if (lv[l]) { double a = 0.0; if ((rc = suhmo_level_norm(lv[l], 0, SUHMO_F_RES, 0, &a, s))) return rc; if (a > m) m = a; }
        if (lv[0]->ar && (rc = lv[0]->ar(lv[0]->user, &m))) return rc;                    // ranks without the finer levels
        *norm = m;
    }
    return 0;
}
extern "C" int suhmo_amr_vcycle(suhmo_level_t **lv, int nlev, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    int rc = check_hierarchy(lv, nlev); if (rc) return rc;
    ARG(sp);
    HIPCHK(hipSetDevice(lv[0]->device));
    return vcycle_amr(lv, nlev - 1, sp, s);
}
extern "C" int suhmo_amr_solve(suhmo_level_t **lv, int nlev, const suhmo_solver_params_t *sp, int *iters, double *hist, suhmo_stream_t s)
{
    ARG(sp);
    int rc;
    double rnorm = 0.0;
    if ((rc = suhmo_amr_residual(lv, nlev, &rnorm, s))) return rc;
    double initial_rnorm = rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    bool goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    bool goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        if ((rc = suhmo_amr_vcycle(lv, nlev, sp, s))) return rc;
        if ((rc = suhmo_amr_residual(lv, nlev, &rnorm, s))) return rc;
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    if (iters) *iters = iter;
    return 0;
}
