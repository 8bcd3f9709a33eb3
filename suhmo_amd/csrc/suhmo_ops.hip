// suhmo_ops.hip -- the level's operator kernels: ghost fill, applyOp / residual, restriction, prolongation, vector operations and norms, and the
// same for every box of a multi-box AMR level in one launch.  (Split off suhmo_level.hip in round 4; reference citations: file:line in the SUHMO checkout.)
#include "suhmo_hier.h"
#include "suhmo_level_int.h"
#include <algorithm>
#include <cmath>
#include <initializer_list>
// ------------------------------------------------------------------ kernels

// exchange (periodic wrap) + mixBCValues into the stored ghost ring of a cell field
// (src/AmrHydro.cpp:248-309).  One thread per perimeter cell.
__device__ __forceinline__ void d_fill_ghosts(const DV &v, double *__restrict__ p, int homog)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * v.ny) {                       // x sides
        int side = t / v.ny, j = t % v.ny;
        if (v.cfx[side]) return;              // coarse-fine side: ghost columns hold interpolated data
        int i = side ? v.nx - 1 : 0;
        int idx = cidx(v, i, j);
        double c = p[idx];
        if (side) p[idx + 1] = phiE(v, p, idx, i, c, homog); else p[idx - 1] = phiW(v, p, idx, i, c, homog);
        return;
    }
    t -= 2 * v.ny;
    if (t < 2 * v.nx) {                       // y sides
        int side = t / v.nx, i = t % v.nx;
        if (v.ext[side]) return;              // rank boundary: ghost rows hold exchanged data
        int j = side ? v.ny - 1 : 0;
        int idx = cidx(v, i, j);
        double c = p[idx];
        if (side) p[idx + v.P] = phiN(v, p, idx, j, c, homog); else p[idx - v.P] = phiS(v, p, idx, j, c, homog);
    }
}
__global__ void k_fill_ghosts(DV v, double *__restrict__ p, int homog)
{
    d_fill_ghosts(v, p, homog);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_fill_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field, int homog)
{
    d_fill_ghosts(vt[blockIdx.z], ft[blockIdx.z].f[field], homog);
}

extern "C" int suhmo_level_fill_ghosts(suhmo_level_t *L, int depth, int field, int homogeneous, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(!is_face(field));
    HIPCHK(hipSetDevice(L->device));
    const DV &v = L->d[depth].v;
    double *p = suhmo_field(L, depth, field);
    int n = 2 * v.ny + 2 * v.nx;
    hipLaunchKernelGGL(k_fill_ghosts, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, v, p, homogeneous);
    HIPCHK(hipGetLastError());
    return 0;
}

// VCNLCOMPUTEOP2D / VCNLCOMPUTERES2D with BC, NL fused.  MODE 0: LPHI = L(phi); 1: RES = rhs - L(phi);
// 2: the FAS right-hand side of a coarse depth in one pass: LPHI = L(phi), RHS = axby(RES, LPHI, 1, 1), PHIOLD = phi
// 3: LPHI = L(phi) and RES = axby(LPHI, RHS, -1, 1) in one pass (the composite residual of an AMR level, suhmo_hier.hip)
// (returns what it stored in RES, MODE 1 / 3; 0 for a thread outside the level)
template <bool HAS_ALPHA, int MODE>
__device__ __forceinline__ double d_apply_at(const DV &v, const FP &fp, suhmo_phys_t ph, int homog, int halo, int hcomp, int i, int j)
{
    if (i >= v.nx || j >= v.ny + halo) return 0.0;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    if (MODE == 2 && (j < 0 || j >= v.ny)) {
        // rank strip: the first hcomp halo rows beyond a rank boundary get the neighbour's right-hand side computed here (its phi
        // and RES were exchanged together), which saves the exchange of RHS; the rows further out only copy phi
        const bool comp = (j < 0 && v.rk[0] && j >= -hcomp) || (j >= v.ny && v.rk[1] && j < v.ny + hcomp);
        if (!comp) { fp.f[SUHMO_F_PHIOLD][idx] = phi[idx]; return 0.0; }
    }
    double c = phi[idx];
    double e = phiE(v, phi, idx, i, c, homog), w = phiW(v, phi, idx, i, c, homog);
    double n = phiN(v, phi, idx, j, c, homog), s = phiS(v, phi, idx, j, c, homog);
    double bxW = fp.f[SUHMO_F_BX][idx], bxE = fp.f[SUHMO_F_BX][idx + 1];
    double byS = fp.f[SUHMO_F_BY][idx], byN = fp.f[SUHMO_F_BY][idx + v.P];
    double nl, dnl;
    nl_terms(ph, c, fp.f[SUHMO_F_B][idx], fp.f[SUHMO_F_PI][idx], fp.f[SUHMO_F_ZB][idx], fp.f[SUHMO_F_MASK][idx], nl, dnl);
    double aterm = HAS_ALPHA ? v.alpha * fp.f[SUHMO_F_ACOEF][idx] : v.alpha;
    double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, byN, byS, nl);
    double res = 0.0;
    if (MODE == 0) fp.f[SUHMO_F_LPHI][idx] = lofphi;
    else if (MODE == 1) fp.f[SUHMO_F_RES][idx] = res = fp.f[SUHMO_F_RHS][idx] - lofphi;
    else if (MODE == 3) { fp.f[SUHMO_F_LPHI][idx] = lofphi; fp.f[SUHMO_F_RES][idx] = res = -1.0 * lofphi + 1.0 * fp.f[SUHMO_F_RHS][idx]; }
    else {
        fp.f[SUHMO_F_LPHI][idx] = lofphi;
        fp.f[SUHMO_F_RHS][idx] = 1.0 * fp.f[SUHMO_F_RES][idx] + 1.0 * lofphi;
        fp.f[SUHMO_F_PHIOLD][idx] = c;
    }
    return res;
}
template <bool HAS_ALPHA, int MODE>
__device__ __forceinline__ void d_apply(const DV &v, const FP &fp, suhmo_phys_t ph, int homog, int halo, int hcomp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - halo;     // MODE 2 on a rank strip: the halo rows only copy phi
    d_apply_at<HAS_ALPHA, MODE>(v, fp, ph, homog, halo, hcomp, i, j);
}
template <bool HAS_ALPHA, int MODE>
__global__ __launch_bounds__(256) void k_apply(DV v, FP fp, suhmo_phys_t ph, int homog, int halo = 0, int hcomp = 0)
{
    d_apply<HAS_ALPHA, MODE>(v, fp, ph, homog, halo, hcomp);
}
// RES = rhs - L(phi) and, in the same pass, the first stage of its max norm (one partial per workgroup, as k_norm_partial leaves them for
// k_norm_final): the solve loop's residual evaluation on levels whose cycle's last launch cannot leave it behind (the tile-kernel sizes)
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_residual_norm(DV v, FP fp, suhmo_phys_t ph, double *__restrict__ partial)
{
    __shared__ double sm[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    double r = fabs(d_apply_at<HAS_ALPHA, 1>(v, fp, ph, 0, 0, 0, i, j));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r = fmax(r, __shfl_xor(r, o));
    if (threadIdx.x == 0) sm[threadIdx.y] = r;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
// LPHI and RES = rhs - LPHI on a list of rectangles (x = first column, y = first row, z = columns, w = rows) of the level: the part of
// a composite residual that has changed since the whole level was evaluated (suhmo_hier.hip); overlapping rectangles write the same values
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_apply_rects(DV v, FP fp, suhmo_phys_t ph, const int4 *__restrict__ rects)
{
    const int4 r = rects[blockIdx.z];
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
    if (a >= r.z || b >= r.w) return;
    d_apply_at<HAS_ALPHA, 3>(v, fp, ph, 0, 0, 0, r.x + a, r.y + b);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
template <bool HAS_ALPHA, int MODE>
__global__ __launch_bounds__(256) void k_apply_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph, int homog)
{
    d_apply<HAS_ALPHA, MODE>(vt[blockIdx.z], ft[blockIdx.z], ph, homog, 0, 0);
}

// the box of blockIdx.z among the boxes of several levels: its level's slot q (unrolled: constant indices only), its index in that level
__device__ __forceinline__ bool lv_find(const suhmo_lvboxes &lv, int &q, int &k)
{
    int z = blockIdx.z;
    q = -1; k = 0;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t < lv.n && q < 0) { if (z < lv.nbox[t]) { q = t; k = z; } else z -= lv.nbox[t]; }
    return q >= 0;
}
#define LV_PICK(lv, q, member, out) do { _Pragma("unroll") for (int t_ = 0; t_ < SUHMO_LVMAX; t_++) if (t_ == (q)) (out) = (lv).member[t_]; } while (0)
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_apply_lv(suhmo_lvboxes lv, suhmo_phys_t ph)
{
    int q, k;
    if (!lv_find(lv, q, k)) return;
    const DV *dv = nullptr; const FP *fp = nullptr; int mode = 1;
    LV_PICK(lv, q, dv, dv); LV_PICK(lv, q, fp, fp); LV_PICK(lv, q, mode, mode);
    if (mode == 1) d_apply<HAS_ALPHA, 1>(dv[k], fp[k], ph, 0, 0, 0);
    else d_apply<HAS_ALPHA, 3>(dv[k], fp[k], ph, 0, 0, 0);
}
int suhmo_levels_apply(const suhmo_lvboxes &lv, const suhmo_phys_t &ph, bool has_alpha, hipStream_t st)
{
    int nz = 0;
    for (int q = 0; q < lv.n; q++) nz += lv.nbox[q];
    if (nz <= 0) return 0;
    const dim3 grd((lv.maxnx + 63) / 64, (lv.maxny + 3) / 4, nz);
    if (has_alpha) hipLaunchKernelGGL(k_apply_lv<true>, grd, BLK2D, 0, st, lv, ph);
    else hipLaunchKernelGGL(k_apply_lv<false>, grd, BLK2D, 0, st, lv, ph);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_exchange_fields(suhmo_level *L, int depth, std::initializer_list<int> fields, hipStream_t st)
{
    const DV &v = L->d[depth].v;
    if (!(L->ex && (v.ext[0] || v.ext[1]))) return 0;
    for (int f : fields) if (!suhmo_field(L, depth, f)) return -2;
    int rc = L->ex(L->user, L, depth, fields.begin(), (int)fields.size(), (suhmo_stream_t)st);
    if (!rc) for (int f : fields) if (f == SUHMO_F_PHI) L->d[depth].phi_fresh = suhmo_halo_rows(v);
    return rc;
}
int suhmo_exchange_list(suhmo_level *L, int depth, const int *fields, int n, hipStream_t st)    // for suhmo_step.hip
{
    const DV &v = L->d[depth].v;
    if (!(L->ex && (v.ext[0] || v.ext[1]))) return 0;
    for (int k = 0; k < n; k++) if (!suhmo_field(L, depth, fields[k])) return -2;
    int rc = L->ex(L->user, L, depth, fields, n, (suhmo_stream_t)st);
    if (!rc) for (int k = 0; k < n; k++) if (fields[k] == SUHMO_F_PHI) L->d[depth].phi_fresh = suhmo_halo_rows(v);
    return rc;
}
// strips: make sure `need` halo rows of phi hold the neighbours' current values.  Every kernel that
// changes phi lowers Depth::phi_fresh (a colour pass that also advances the halo rows redundantly loses
// one row, a K-sweep fused launch 2K), so an exchange happens only when the stencil about to run would
// reach stale rows -- LevelData::exchange of the reference (src/VCAMRNonLinearPoissonOp.cpp:47,124,304,405,692)
int suhmo_ensure_phi_halo(suhmo_level *L, int depth, int need, hipStream_t st)
{
    Depth &D = L->d[depth];
    if (!(L->ex && (D.v.ext[0] || D.v.ext[1]))) return 0;
    if (D.phi_fresh >= need) return 0;
    return suhmo_exchange_fields(L, depth, {SUHMO_F_PHI}, st);
}

extern "C" int suhmo_level_apply_op(suhmo_level_t *L, int depth, int homogeneous, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, (hipStream_t)s); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 0>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, homogeneous);
    else hipLaunchKernelGGL((k_apply<false, 0>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, homogeneous);
    HIPCHK(hipGetLastError());
    return 0;
}

// applyOpI (inhomogeneous) and the residual of it in one pass: LPHI = L(phi), RES = rhs - L(phi)
int suhmo_apply_and_residual(suhmo_level *L, int depth, hipStream_t st)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_RES)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 3>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph, 0);
    else hipLaunchKernelGGL((k_apply<false, 3>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_apply_and_residual_rects(suhmo_level *L, int depth, const int4 *d_rects, int n, int maxw, int maxh, hipStream_t st)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_RES)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (!n) return 0;
    dim3 grd((maxw + 63) / 64, (maxh + 3) / 4, n);
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply_rects<true>), grd, BLK2D, 0, st, D.v, D.fp, L->ph, d_rects);
    else hipLaunchKernelGGL((k_apply_rects<false>), grd, BLK2D, 0, st, D.v, D.fp, L->ph, d_rects);
    HIPCHK(hipGetLastError());
    return 0;
}

// FAS cycle, coarse depth after the restriction: rhs_c = res_c + L_c(R phi) and the copy of R phi the prolongation
// subtracts (on a rank strip with its exchanged halo rows), one pass instead of applyOp + axby + a device copy
int suhmo_fas_coarse_rhs(suhmo_level *L, int depth, hipStream_t st, int hcomp)
{
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_PHIOLD)) return -2;
    const int halo = (D.v.ext[0] || D.v.ext[1]) ? D.v.gy : 0;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 2>), grid2d(D.v.nx, D.v.ny + 2 * halo), BLK2D, 0, st, D.v, D.fp, L->ph, 0, halo, hcomp);
    else hipLaunchKernelGGL((k_apply<false, 2>), grid2d(D.v.nx, D.v.ny + 2 * halo), BLK2D, 0, st, D.v, D.fp, L->ph, 0, halo, hcomp);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int suhmo_level_residual(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::residualI");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    int rc = suhmo_ensure_phi_halo(L, depth, 1, (hipStream_t)s); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 1>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, 0);
    else hipLaunchKernelGGL((k_apply<false, 1>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// COMPUTENONLINEARTERMS / lambda as stand-alone kernels (parity of a2, a9)
__global__ void k_nonlinear(DV v, FP fp, suhmo_phys_t ph)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double nl, dnl;
    nl_terms(ph, fp.f[SUHMO_F_PHI][idx], fp.f[SUHMO_F_B][idx], fp.f[SUHMO_F_PI][idx], fp.f[SUHMO_F_ZB][idx], fp.f[SUHMO_F_MASK][idx], nl, dnl);
    fp.f[SUHMO_F_NL][idx] = nl; fp.f[SUHMO_F_DNL][idx] = dnl;
}
__global__ void k_lambda(DV v, FP fp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double aterm = fp.f[SUHMO_F_ACOEF][idx] * v.alpha;
    fp.f[SUHMO_F_LAMBDA][idx] = lambda_cell(v, aterm, fp.f[SUHMO_F_BX][idx + 1], fp.f[SUHMO_F_BX][idx],
                                            fp.f[SUHMO_F_BY][idx + v.P], fp.f[SUHMO_F_BY][idx]);
}
extern "C" int suhmo_level_nonlinear(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_NL) || !suhmo_field(L, depth, SUHMO_F_DNL)) return -2;
    hipLaunchKernelGGL(k_nonlinear, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_compute_lambda(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LAMBDA)) return -2;
    hipLaunchKernelGGL(k_lambda, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int suhmo_level_gsrb(suhmo_level_t *L, int depth, int sweeps, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::levelGSRB");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(sweeps >= 0);
    HIPCHK(hipSetDevice(L->device));
    int rc = suhmo_launch_gsrb(L, depth, sweeps, 0, (hipStream_t)s);
    if (rc) return rc;
    // levelGSRB leaves the ghosts with the HOMOGENEOUS BC applied (:757-759)
    if (sweeps > 0) return suhmo_level_fill_ghosts(L, depth, SUHMO_F_PHI, 1, s);
    return 0;
}

// RESTRICTRESVCNL2D (src/VCAMRNonLinearPoissonOpF.ChF:516-558) fused with BC + NL: one thread
// per coarse cell; the four fine contributions are accumulated in the reference's loop order
// (2I,2J), (2I+1,2J), (2I,2J+1), (2I+1,2J+1) onto a zero-initialised coarse value.
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_restrict_residual(DV v, FP fp, DV vc, double *__restrict__ resC, double *__restrict__ phiC, suhmo_phys_t ph)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vc.nx || J >= vc.ny) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    // the thread's 2 x 2 fine cells start at an even column: every row is read as 16-byte pairs (canvas column
    // SUHMO_XOFF + 2I is 16-byte aligned), the W / E neighbours of the pair as single values
    const int i0 = 2 * I, j0 = 2 * J, base = cidx(v, i0, j0);
    auto ld2 = [&](const double *__restrict__ p, int idx) { return *reinterpret_cast<const double2 *>(p + idx); };
    double2 pc[2], pS, pN;
    pc[0] = ld2(phi, base); pc[1] = ld2(phi, base + v.P);
    // south of row j0 / north of row j0 + 1 (physical BC evaluated from the adjacent interior value)
    if (j0 > 0 || v.ext[0]) pS = ld2(phi, base - v.P);
    else if (v.per[1]) pS = ld2(phi, base + (v.ny - 1) * v.P);
    else { pS.x = phiS(v, phi, base, j0, pc[0].x, false); pS.y = phiS(v, phi, base + 1, j0, pc[0].y, false); }
    if (j0 + 1 < v.ny - 1 || v.ext[1]) pN = ld2(phi, base + 2 * v.P);
    else if (v.per[1]) pN = ld2(phi, base + v.P - (v.ny - 1) * v.P);
    else { pN.x = phiN(v, phi, base + v.P, j0 + 1, pc[1].x, false); pN.y = phiN(v, phi, base + v.P + 1, j0 + 1, pc[1].y, false); }
    double acc = 0.0, accp = 0.0;      // accp: RESTRICTVCNL of phi (restrictR), same visiting order
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int j = j0 + b, idx = base + b * v.P;
        const double2 cc = pc[b];
        const double w0 = phiW(v, phi, idx, i0, cc.x, false), e1 = phiE(v, phi, idx + 1, i0 + 1, cc.y, false);
        const double2 sS = b == 0 ? pS : pc[0], nN = b == 0 ? pc[1] : pN;
        const double2 bx01 = ld2(fp.f[SUHMO_F_BX], idx); const double bx2 = fp.f[SUHMO_F_BX][idx + 2];
        const double2 byS = ld2(fp.f[SUHMO_F_BY], idx), byN = ld2(fp.f[SUHMO_F_BY], idx + v.P);
        const double2 B2 = ld2(fp.f[SUHMO_F_B], idx), Pi2 = ld2(fp.f[SUHMO_F_PI], idx), zb2 = ld2(fp.f[SUHMO_F_ZB], idx), mk2 = ld2(fp.f[SUHMO_F_MASK], idx);
        const double2 rhs2 = ld2(fp.f[SUHMO_F_RHS], idx);
        double2 a2 = make_double2(0.0, 0.0);
        if (HAS_ALPHA) a2 = ld2(fp.f[SUHMO_F_ACOEF], idx);
        (void)j;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const double c = a ? cc.y : cc.x, w = a ? cc.x : w0, e = a ? e1 : cc.y;
            const double n = a ? nN.y : nN.x, s_ = a ? sS.y : sS.x;
            const double bxW = a ? bx01.y : bx01.x, bxE = a ? bx2 : bx01.y;
            double nl, dnl;
            nl_terms(ph, c, a ? B2.y : B2.x, a ? Pi2.y : Pi2.x, a ? zb2.y : zb2.x, a ? mk2.y : mk2.x, nl, dnl);
            double aterm = HAS_ALPHA ? v.alpha * (a ? a2.y : a2.x) : v.alpha;
            double lofphi = lofphi_cell(v, aterm, c, e, w, n, s_, bxE, bxW, a ? byN.y : byN.x, a ? byS.y : byS.x, nl);
            acc = acc + ((a ? rhs2.y : rhs2.x) - lofphi) / 4.0;
            accp = accp + c / 4.0;
        }
    }
    resC[cidx(vc, I, J)] = acc;
    if (phiC) phiC[cidx(vc, I, J)] = accp;
}

static int restrict_residual_impl(suhmo_level *L, int depth, bool also_phi, hipStream_t st)
{
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (also_phi) phi_changed(L, depth + 1);
    double *phiC = also_phi ? C.fp.f[SUHMO_F_PHI] : nullptr;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL(k_restrict_residual<true>, grid2d(C.v.nx, C.v.ny), BLK2D, 0, st, D.v, D.fp, C.v, C.fp.f[SUHMO_F_RES], phiC, L->ph);
    else hipLaunchKernelGGL(k_restrict_residual<false>, grid2d(C.v.nx, C.v.ny), BLK2D, 0, st, D.v, D.fp, C.v, C.fp.f[SUHMO_F_RES], phiC, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_restrict_residual(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::restrictResidual");
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    return restrict_residual_impl(L, depth, false, (hipStream_t)s);
}
// restrictResidual + restrictR of the FAS cycle in one pass over the fine level
int suhmo_restrict_both(suhmo_level *L, int depth, hipStream_t st) { return restrict_residual_impl(L, depth, true, st); }

// RESTRICTVCNL (src/VCAMRNonLinearPoissonOpF.ChF:432-446)
__global__ void k_restrict_r(DV v, const double *__restrict__ f, DV vc, double *__restrict__ c)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vc.nx || J >= vc.ny) return;
    int idx = cidx(v, 2 * I, 2 * J);
    double acc = 0.0;
    acc = acc + f[idx] / 4.0;
    acc = acc + f[idx + 1] / 4.0;
    acc = acc + f[idx + v.P] / 4.0;
    acc = acc + f[idx + v.P + 1] / 4.0;
    c[cidx(vc, I, J)] = acc;
}
extern "C" int suhmo_level_restrict_r(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth + 1);
    hipLaunchKernelGGL(k_restrict_r, grid2d(C.v.nx, C.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_PHI]);
    HIPCHK(hipGetLastError());
    return 0;
}

// PROLONGNL (src/AMRNonLinearPoissonOpF.ChF:617-628), m = 2
__global__ void k_prolong(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    phi[idx] = phi[idx] + c[cidx(vc, i / 2, j / 2)];
}
extern "C" int suhmo_level_prolong_increment(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::prolongIncrement");
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth);
    hipLaunchKernelGGL(k_prolong, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_CORR]);
    HIPCHK(hipGetLastError());
    return 0;
}

// FAS correction of the cycle: CORR_c = 1*phi_c + (-1)*phi_c,old (LevelDataOps::axby), phi += P(CORR_c) (PROLONGNL).
// On rank strips both kernels also cover the halo rows that are valid on BOTH depths (fine: phi_fresh rows, coarse:
// phi_fresh rows of phi_c; PHIOLD was copied after the exchange), so the post-smoothing can start without an exchange.
__global__ void k_axby_rows(DV v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b, int jlo, int jhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i >= v.nx || j > jhi) return;
    int idx = cidx(v, i, j);
    dst[idx] = a * x[idx] + b * y[idx];
}
__global__ void k_prolong_rows(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c, int jlo, int jhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i >= v.nx || j > jhi) return;
    int idx = cidx(v, i, j);
    phi[idx] = phi[idx] + c[cidx(vc, i / 2, j >> 1)];          // j >> 1: floor, halo rows have j < 0
}
int suhmo_prolong_with_halo(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    const bool ext = L->ex && (D.v.ext[0] || D.v.ext[1]);
    int R = 0;                                              // fine halo rows that stay valid through the prolongation
    if (ext) { R = D.phi_fresh < 2 * C.phi_fresh ? D.phi_fresh : 2 * C.phi_fresh; R &= ~1; }
    const int Rc = R / 2;
    const int jlo = D.v.ext[0] ? -R : 0, jhi = D.v.ny - 1 + (D.v.ext[1] ? R : 0);
    const int cjlo = C.v.ext[0] ? -Rc : 0, cjhi = C.v.ny - 1 + (C.v.ext[1] ? Rc : 0);
    double *corr = suhmo_field(L, depth + 1, SUHMO_F_CORR);
    if (!corr) return -2;
    hipLaunchKernelGGL(k_axby_rows, grid2d(C.v.nx, cjhi - cjlo + 1), BLK2D, 0, st, C.v, corr, C.fp.f[SUHMO_F_PHI], C.fp.f[SUHMO_F_PHIOLD], 1.0, -1.0, cjlo, cjhi);
    hipLaunchKernelGGL(k_prolong_rows, grid2d(D.v.nx, jhi - jlo + 1), BLK2D, 0, st, D.v, D.fp.f[SUHMO_F_PHI], C.v, corr, jlo, jhi);
    HIPCHK(hipGetLastError());
    D.phi_fresh = R;
    return 0;
}

// PROLONG_2_NL (src/AMRNonLinearPoissonOpF.ChF:660-705), coarse data read with its stored ghosts
__global__ void k_prolong2(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
    int ic = i / 2, jc = j / 2, o1 = 2 * (i % 2) - 1, o2 = 2 * (j % 2) - 1;
    int idx = cidx(v, i, j), cc = cidx(vc, ic, jc);
    double p = phi[idx];
    p = p + fx2 * c[cc] + f0 * c[cc + o1 + o2 * vc.P];
    p = p + fx1 * (c[cc + o1] + c[cc + o2 * vc.P]);
    phi[idx] = p;
}
extern "C" int suhmo_level_prolong_bilinear(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth);
    hipLaunchKernelGGL(k_prolong2, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_CORR]);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ small operators
// DIVERGENCE (util/DivergenceF.ChF:38-54), called for dir 0 then dir 1
__global__ void k_divergence(DV v, const double *__restrict__ ux, const double *__restrict__ uy, double *__restrict__ div)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double d = div[idx];
    d = d + v.fdx * (ux[idx + 1] - ux[idx]);
    d = d + v.fdy * (uy[idx + v.P] - uy[idx]);
    div[idx] = d;
}
extern "C" int suhmo_level_divergence(suhmo_level_t *L, int depth, int dst_field, suhmo_stream_t s)
{
    CHECK_DF(L, depth, dst_field); ARG(!is_face(dst_field));
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    double *dst = suhmo_field(L, depth, dst_field);
    hipLaunchKernelGGL(k_divergence, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_BX], D.fp.f[SUHMO_F_BY], dst);
    HIPCHK(hipGetLastError());
    return 0;
}

// getFlux (src/VCAMRNonLinearPoissonOp.cpp:820-840): F = -b * ((phi_hi - phi_lo) * (beta*ref/dx))
__global__ void k_getflux(DV v, FP fp, int dir, double scale, double *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    int nxf = dir == 0 ? v.nx + 1 : v.nx, nyf = dir == 0 ? v.ny : v.ny + 1;
    if (i >= nxf || j >= nyf) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    double phihi = phi[idx], philo = dir == 0 ? phi[idx - 1] : phi[idx - v.P];   // stored ghosts
    double gradphi = (phihi - philo) * scale;
    out[(size_t)j * nxf + i] = -fp.f[dir == 0 ? SUHMO_F_BX : SUHMO_F_BY][idx] * gradphi;
}
extern "C" int suhmo_level_get_flux(suhmo_level_t *L, int depth, int dir, int ref, double *flux_host, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(dir == 0 || dir == 1); ARG(flux_host);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    int rc = suhmo_level_fill_ghosts(L, depth, SUHMO_F_PHI, 0, s); if (rc) return rc;
    int nxf = dir == 0 ? D.v.nx + 1 : D.v.nx, nyf = dir == 0 ? D.v.ny : D.v.ny + 1;
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, (size_t)nxf * nyf * 8));
    double scale = D.v.beta * ref / (dir == 0 ? D.v.dx : D.v.dy);
    hipLaunchKernelGGL(k_getflux, grid2d(nxf, nyf), BLK2D, 0, st, D.v, D.fp, dir, scale, tmp);
    HIPCHK(hipMemcpyAsync(flux_host, tmp, (size_t)nxf * nyf * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipFree(tmp));
    return 0;
}

// LevelDataOps::axby / setVal on valid cells
__device__ __forceinline__ void d_axby(const DV &v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    dst[idx] = a * x[idx] + b * y[idx];
}
__global__ void k_axby(DV v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b)
{
    d_axby(v, dst, x, y, a, b);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_axby_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int fd, int fx, int fy, double a, double b)
{
    d_axby(vt[blockIdx.z], ft[blockIdx.z].f[fd], ft[blockIdx.z].f[fx], ft[blockIdx.z].f[fy], a, b);
}
__global__ void k_setval(DV v, double *__restrict__ dst, double val)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    dst[cidx(v, i, j)] = val;
}
extern "C" int suhmo_level_axby(suhmo_level_t *L, int depth, int dst, int x, int y, double a, double b, suhmo_stream_t s)
{
    CHECK_DF(L, depth, dst); CHECK_DF(L, depth, x); CHECK_DF(L, depth, y);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    double *pd = suhmo_field(L, depth, dst), *px = suhmo_field(L, depth, x), *py = suhmo_field(L, depth, y);
    if (dst == SUHMO_F_PHI) phi_changed(L, depth);
    if (dst == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }      // (the reports about the ice mask end with any write to it)
    hipLaunchKernelGGL(k_axby, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, pd, px, py, a, b);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_set_value(suhmo_level_t *L, int depth, int field, double val, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (field == SUHMO_F_PHI) phi_changed(L, depth);
    if (field == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }
    hipLaunchKernelGGL(k_setval, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, suhmo_field(L, depth, field), val);
    HIPCHK(hipGetLastError());
    return 0;
}

// norms over valid cells.  ord 0: max |x| (exact, order independent).  ord 2: sqrt(sum x^2),
// two-stage deterministic reduction (fixed partial order; differs from the serial CPU sum
// by rounding only).
__global__ __launch_bounds__(256) void k_norm_partial(DV v, const double *__restrict__ x, int ord, double *__restrict__ partial)
{
    __shared__ double sm[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) {
            double val = x[cidx(v, i, j)];
            if (ord == 0) acc = fmax(acc, fabs(val)); else acc += val * val;
        }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] = ord == 0 ? fmax(sm[tid], sm[tid + s]) : sm[tid] + sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}
__global__ void k_norm_final(const double *__restrict__ partial, int n, int ord, double *__restrict__ out, HostSlot hs)
{
    __shared__ double sm[256];
    int tid = threadIdx.x;
    double acc = 0.0;
    for (int k = tid; k < n; k += 256) acc = ord == 0 ? fmax(acc, partial[k]) : acc + partial[k];
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] = ord == 0 ? fmax(sm[tid], sm[tid + s]) : sm[tid] + sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) { out[0] = sm[0]; suhmo_publish(hs, sm[0]); }
}
// The 8-byte result of a reduction.  Synchronising the stream costs ~17 us of idle GPU per read-back on this platform (the copy
// kernel, the interrupt, the wake-up); instead the reduction's last kernel stores the value and then, with a system-scope release,
// a sequence number into pinned coherent host memory, and the host spins on the number.  In-order stream: when the number is
// there, everything enqueued before is done.  A kernel that never publishes (a fault) ends in the stream synchronisation below,
// which reports it.
HostSlot suhmo_host_slot(suhmo_level *L)
{
    HostSlot h{nullptr, nullptr, 0};
    if (L->poll_readback) { h.val = L->hscratch_dev; h.flag = (unsigned long long *)(L->hscratch_dev + 8); h.seq = ++L->hseq; }
    return h;
}
int suhmo_readback(suhmo_level *L, hipStream_t st, double *out, double *out2)
{
    if (L->poll_readback) {
        HIPCHK(hipGetLastError());
        volatile unsigned long long *flag = (volatile unsigned long long *)(L->hscratch + 8);
        for (long spin = 0; spin < 400000000L; spin++) {
            if (__atomic_load_n((unsigned long long *)flag, __ATOMIC_ACQUIRE) == L->hseq) { *out = L->hscratch[0]; if (out2) *out2 = L->hscratch[1]; return 0; }
            if ((spin & 0xffff) == 0xffff && hipStreamQuery(st) == hipSuccess) break;     // finished without us seeing the store: read below
        }
        HIPCHK(hipStreamSynchronize(st));
        if (__atomic_load_n((unsigned long long *)flag, __ATOMIC_ACQUIRE) == L->hseq) { *out = L->hscratch[0]; if (out2) *out2 = L->hscratch[1]; return 0; }
        // not published (mapping not coherent on this system): fall back for good
        L->poll_readback = 0;
    }
    HIPCHK(hipMemcpyAsync(L->hscratch, L->scratch, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out = L->hscratch[0];
    if (out2) *out2 = L->hscratch[1];
    return 0;
}
__global__ void k_publish2(const double *__restrict__ v, HostSlot hs)
{
    if (hs.val) hs.val[1] = v[1];
    suhmo_publish(hs, v[0]);
}
HostSlot suhmo_reduce_slot(suhmo_level *L)
{
    if (on_strip(L) && L->ard) return HostSlot{nullptr, nullptr, 0};       // published after the device all-reduce
    return suhmo_host_slot(L);
}
int suhmo_reduce_finish(suhmo_level *L, hipStream_t st, int n, int op, double *out, double *out2)
{
    ARG(n == 1 || n == 2);
    double v[2] = {0.0, 0.0};
    int rc;
    if (on_strip(L) && L->ard) {
        if ((rc = L->ard(L->user, L->scratch, n, op, (suhmo_stream_t)st))) return rc;
        hipLaunchKernelGGL(k_publish2, dim3(1), dim3(1), 0, st, L->scratch, suhmo_host_slot(L));
        HIPCHK(hipGetLastError());
        if ((rc = suhmo_readback(L, st, &v[0], &v[1]))) return rc;
    } else {
        if ((rc = suhmo_readback(L, st, &v[0], &v[1]))) return rc;
        if (on_strip(L) && (L->ar || L->ar2)) {
            if (L->ar2) { if ((rc = L->ar2(L->user, v, n, op))) return rc; }
            else if (op == 0) { for (int k = 0; k < n; k++) if ((rc = L->ar(L->user, &v[k]))) return rc; }
            else { suhmo_set_error("a SUM over the ranks of a strip needs suhmo_level_set_reduce_hook or the native transport (the hook of suhmo_level_set_hooks reduces MAX only)"); return -5; }
        }
    }
    *out = v[0];
    if (out2) *out2 = v[1];
    return 0;
}
__global__ __launch_bounds__(256) void k_dot_partial(DV v, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partial)
{
    __shared__ double sm[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) { int idx = cidx(v, i, j); acc += x[idx] * y[idx]; }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = sm[tid] + sm[tid + s]; __syncthreads(); }
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}
extern "C" int suhmo_level_dot(suhmo_level_t *L, int depth, int x, int y, double *out, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::dotProduct");
    CHECK_DF(L, depth, x); CHECK_DF(L, depth, y); ARG(out);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    const double *px = suhmo_field(L, depth, x), *py = suhmo_field(L, depth, y);
    if (!px || !py) { suhmo_set_error("field allocation failed"); return -2; }
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    hipLaunchKernelGGL(k_dot_partial, grd, BLK2D, 0, st, D.v, px, py, L->scratch + 2);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, (int)(grd.x * grd.y), 2, L->scratch, suhmo_reduce_slot(L));
    return suhmo_reduce_finish(L, st, 1, 1, out);
}
extern "C" int suhmo_level_norm(suhmo_level_t *L, int depth, int field, int ord, double *out, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::norm");
    CHECK_DF(L, depth, field); ARG(out); ARG(ord == 0 || ord == 2);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    int np = grd.x * grd.y;
    hipLaunchKernelGGL(k_norm_partial, grd, BLK2D, 0, st, D.v, suhmo_field(L, depth, field), ord, L->scratch + 2);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, np, ord, L->scratch, suhmo_reduce_slot(L));
    double r = 0.0;
    // the reference's norm() reduces over the ranks (src/AMRNonLinearPoissonOp.cpp:1222-1264): MAX for the max norm, SUM of the squares for l2
    { int rc = suhmo_reduce_finish(L, st, 1, ord == 0 ? 0 : 1, &r); if (rc) return rc; }
    if (ord == 2) r = sqrt(r);
    *out = r;
    return 0;
}

// residualI of depth 0 and the max norm of the result in two launches instead of three (levels of up to scratch-many workgroups)
int suhmo_level_residual_and_norm(suhmo_level *L, double *out, hipStream_t st)
{
    Depth &D = L->d[0];
    const dim3 grd = grid2d(D.v.nx, D.v.ny);
    const size_t np = (size_t)grd.x * grd.y;
    int rc;
    if (np + 4 >= L->scratch_elems) {
        if ((rc = suhmo_level_residual(L, 0, (suhmo_stream_t)st))) return rc;
        return suhmo_level_norm(L, 0, SUHMO_F_RES, 0, out, (suhmo_stream_t)st);
    }
    if ((rc = suhmo_ensure_phi_halo(L, 0, 1, st))) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL(k_residual_norm<true>, grd, BLK2D, 0, st, D.v, D.fp, L->ph, L->scratch + 2);
    else hipLaunchKernelGGL(k_residual_norm<false>, grd, BLK2D, 0, st, D.v, D.fp, L->ph, L->scratch + 2);
    HIPCHK(hipGetLastError());
    return suhmo_level_norm_from_partials(L, (int)np, out, st);
}
// max norm of RES at depth 0 from the partial maxima the cycle's last launch left behind (suhmo_gsrb.hip, residual output): the second
// stage of suhmo_level_norm alone
int suhmo_level_norm_from_partials(suhmo_level *L, int np, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, np, 0, L->scratch, suhmo_reduce_slot(L));
    double r = 0.0;
    int rc = suhmo_reduce_finish(L, st, 1, 0, &r); if (rc) return rc;
    *out = r;
    return 0;
}

// ------------------------------------------------------------------ every box of a multi-box AMR level in one launch
__global__ void k_copy_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int fd, int fs)
{
    const DV &v = vt[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    ft[blockIdx.z].f[fd][idx] = ft[blockIdx.z].f[fs][idx];
}
// A level of boxes entering / leaving its FAS problem in an AMR V-cycle (suhmo_hier.hip:vcycle_amr), one launch each instead of three / two:
//   enter: RHS0 <- RHS (copy, ghost ring included), RHS <- 1 RES + 1 LPHI (axby, valid cells), PHIOLD <- PHI (copy)
//   leave: RHS <- RHS0 (copy), CORR <- 1 PHI + (-1) PHIOLD (axby)               -- the expressions of k_copy_m / k_axby_m on the same operands
__global__ void k_fas_enter_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    const DV &v = vt[blockIdx.z];
    const FP &f = ft[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    f.f[SUHMO_F_RHS0][idx] = f.f[SUHMO_F_RHS][idx];
    f.f[SUHMO_F_PHIOLD][idx] = f.f[SUHMO_F_PHI][idx];
    if (i >= 0 && i < v.nx && j >= 0 && j < v.ny) f.f[SUHMO_F_RHS][idx] = 1.0 * f.f[SUHMO_F_RES][idx] + 1.0 * f.f[SUHMO_F_LPHI][idx];
}
__global__ void k_fas_leave_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    const DV &v = vt[blockIdx.z];
    const FP &f = ft[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    f.f[SUHMO_F_RHS][idx] = f.f[SUHMO_F_RHS0][idx];
    if (i >= 0 && i < v.nx && j >= 0 && j < v.ny) f.f[SUHMO_F_CORR][idx] = 1.0 * f.f[SUHMO_F_PHI][idx] + -1.0 * f.f[SUHMO_F_PHIOLD][idx];
}
// fields of the boxes of one hierarchy <- fields of the same boxes of another (the implicit gap-height operator's copy of a level)
struct CopyPairs { int n, fd[4], fs[4]; };
__global__ void k_copy_between_m(const DV *__restrict__ vt, const FP *__restrict__ fdst, const FP *__restrict__ fsrc, CopyPairs cp)
{
    const DV &v = vt[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    for (int q = 0; q < cp.n; q++) fdst[blockIdx.z].f[cp.fd[q]][idx] = fsrc[blockIdx.z].f[cp.fs[q]][idx];
}
__global__ __launch_bounds__(256) void k_norm_max_partial_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field, double *__restrict__ partial)
{
    __shared__ double sm[256];
    const DV &v = vt[blockIdx.z];
    const double *__restrict__ x = ft[blockIdx.z].f[field];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) acc = fmax(acc, fabs(x[cidx(v, i, j)]));
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = fmax(sm[tid], sm[tid + s]); __syncthreads(); }
    if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sm[0];
}
int suhmo_multi_fill_ghosts(const suhmo_multi &m, int field, int homog, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_fill_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp, field, homog);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_apply(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, int mode, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    if (has_alpha) { if (mode == 0) hipLaunchKernelGGL((k_apply_m<true, 0>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
                     else if (mode == 1) hipLaunchKernelGGL((k_apply_m<true, 1>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
                     else hipLaunchKernelGGL((k_apply_m<true, 3>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0); }
    else { if (mode == 0) hipLaunchKernelGGL((k_apply_m<false, 0>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
           else if (mode == 1) hipLaunchKernelGGL((k_apply_m<false, 1>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
           else hipLaunchKernelGGL((k_apply_m<false, 3>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0); }
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_axby(const suhmo_multi &m, int fd, int fx, int fy, double a, double b, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_axby_m, grid_m(m), BLK2D, 0, st, m.dv, m.fp, fd, fx, fy, a, b);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_copy(const suhmo_multi &m, int fd, int fs, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_copy_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp, fd, fs);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_fas_enter(const suhmo_multi &m, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    hipLaunchKernelGGL(k_fas_enter_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_fas_leave(const suhmo_multi &m, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    hipLaunchKernelGGL(k_fas_leave_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_copy_between(const suhmo_multi &dst, const suhmo_multi &src, const int *fd, const int *fs, int n, hipStream_t st)
{
    if (n < 1 || n > 4 || dst.nbox != src.nbox) { suhmo_set_error("internal: copy between hierarchies"); return -4; }
    if (src.nbox <= 0) return 0;
    CopyPairs cp;
    cp.n = n;
    for (int q = 0; q < n; q++) { cp.fd[q] = fd[q]; cp.fs[q] = fs[q]; }
    hipLaunchKernelGGL(k_copy_between_m, grid_m(src, 2, 2), BLK2D, 0, st, src.dv, dst.fp, src.fp, cp);
    HIPCHK(hipGetLastError());
    return 0;
}
// AMRNorm over a hierarchy in ONE read-back: the first stages of level 0 (suhmo_level_norm_partials) and of every level of boxes
// (suhmo_multi_norm_max_partials) leave their partial maxima where they always do; one launch takes the maximum of all the lists
// (a maximum does not care about the order) and publishes it
constexpr int NORM_LISTS = 8;                        // levels of a hierarchy at most (suhmo_hier_create)
struct NormLists { const double *p[NORM_LISTS]; int n[NORM_LISTS]; int cnt; };
__global__ void k_norm_max_final_lists(NormLists nl, double *__restrict__ out, HostSlot hs)
{
    __shared__ double sm[256];
    int tid = threadIdx.x;
    double acc = 0.0;
    for (int q = 0; q < nl.cnt; q++)
        for (int k = tid; k < nl.n[q]; k += 256) acc = fmax(acc, nl.p[q][k]);
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = fmax(sm[tid], sm[tid + s]); __syncthreads(); }
    if (tid == 0) { out[0] = sm[0]; suhmo_publish(hs, sm[0]); }
}
int suhmo_level_norm_max_partials(suhmo_level *L, int field, const double **partials, int *np, hipStream_t st)
{
    Depth &D = L->d[0];
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    hipLaunchKernelGGL(k_norm_partial, grd, BLK2D, 0, st, D.v, suhmo_field(L, 0, field), 0, L->scratch + 2);
    HIPCHK(hipGetLastError());
    *partials = L->scratch + 2; *np = grd.x * grd.y;
    return 0;
}
int suhmo_multi_norm_max_partials(const suhmo_multi &m, int field, const double **partials, int *np, hipStream_t st)
{
    *partials = m.red; *np = 0;
    if (m.nbox <= 0) return 0;
    dim3 grd(std::min((m.maxnx + 63) / 64, 4), std::min((m.maxny + 3) / 4, 16), m.nbox);
    hipLaunchKernelGGL(k_norm_max_partial_m, grd, BLK2D, 0, st, m.dv, m.fp, field, m.red);
    HIPCHK(hipGetLastError());
    *np = (int)(grd.x * grd.y * grd.z);
    return 0;
}
int suhmo_norm_max_of_lists(suhmo_level *slot, const double *const *partials, const int *np, int cnt, double *out, hipStream_t st)
{
    if (cnt > NORM_LISTS) { suhmo_set_error("internal: norm over more lists than levels"); return -4; }
    NormLists nl;
    nl.cnt = cnt;
    for (int q = 0; q < cnt; q++) { nl.p[q] = partials[q]; nl.n[q] = np[q]; }
    hipLaunchKernelGGL(k_norm_max_final_lists, dim3(1), dim3(256), 0, st, nl, slot->scratch, suhmo_host_slot(slot));
    HIPCHK(hipGetLastError());
    return suhmo_readback(slot, st, out);
}
// the composite norm's covered cells (AMRNorm zeroes them, src/AMRNonLinearPoissonOp.cpp:1241-1258) and its first stage in one pass
__device__ __forceinline__ void d_norm_max_cover(const DV &v, double *__restrict__ x, const double *__restrict__ cover, double *__restrict__ partial, int slot)
{
    __shared__ double sm[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) {
            const int idx = cidx(v, i, j);
            if (cover && cover[idx] != 0.0) x[idx] = 0.0; else acc = fmax(acc, fabs(x[idx]));
        }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = fmax(sm[tid], sm[tid + s]); __syncthreads(); }
    if (tid == 0) partial[slot] = sm[0];
}
__global__ __launch_bounds__(256) void k_norm_max_cover(DV v, double *__restrict__ x, const double *__restrict__ cover, double *__restrict__ partial)
{
    d_norm_max_cover(v, x, cover, partial, blockIdx.y * gridDim.x + blockIdx.x);
}
__global__ __launch_bounds__(256) void k_norm_max_cover_lv(suhmo_lvboxes lv, int field, double *__restrict__ partial)
{
    int q, k;
    const int slot = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (!lv_find(lv, q, k)) { if (threadIdx.x == 0 && threadIdx.y == 0) partial[slot] = 0.0; return; }
    const DV *dv = nullptr; const FP *fp = nullptr; int covered = 0;
    LV_PICK(lv, q, dv, dv); LV_PICK(lv, q, fp, fp); LV_PICK(lv, q, mode, covered);
    d_norm_max_cover(dv[k], fp[k].f[field], covered ? fp[k].f[SUHMO_F_COVER] : nullptr, partial, slot);
}
// lv.mode[q] != 0: the level has cells under a finer one (every level but the finest)
int suhmo_levels_norm_max_cover_partials(const suhmo_lvboxes &lv, int field, double *partial, int *np, hipStream_t st)
{
    int nz = 0;
    for (int q = 0; q < lv.n; q++) nz += lv.nbox[q];
    *np = 0;
    if (nz <= 0) return 0;
    dim3 grd(std::min((lv.maxnx + 63) / 64, 4), std::min((lv.maxny + 3) / 4, 16), nz);
    hipLaunchKernelGGL(k_norm_max_cover_lv, grd, BLK2D, 0, st, lv, field, partial);
    HIPCHK(hipGetLastError());
    *np = (int)(grd.x * grd.y * grd.z);
    return 0;
}
int suhmo_level_norm_max_cover_partials(suhmo_level *L, int field, const double **partials, int *np, hipStream_t st)
{
    Depth &D = L->d[0];
    double *x = suhmo_field(L, 0, field), *cover = suhmo_field(L, 0, SUHMO_F_COVER);
    if (!x || !cover) { suhmo_set_error("field allocation failed"); return -2; }
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    hipLaunchKernelGGL(k_norm_max_cover, grd, BLK2D, 0, st, D.v, x, cover, L->scratch + 2);
    HIPCHK(hipGetLastError());
    *partials = L->scratch + 2; *np = grd.x * grd.y;
    return 0;
}
int suhmo_multi_norm_max(const suhmo_multi &m, suhmo_level *slot, int field, double *out, hipStream_t st)
{
    if (m.nbox <= 0) { *out = 0.0; return 0; }
    dim3 grd(std::min((m.maxnx + 63) / 64, 4), std::min((m.maxny + 3) / 4, 16), m.nbox);
    hipLaunchKernelGGL(k_norm_max_partial_m, grd, BLK2D, 0, st, m.dv, m.fp, field, m.red);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, m.red, (int)(grd.x * grd.y * grd.z), 0, slot->scratch, suhmo_host_slot(slot));
    HIPCHK(hipGetLastError());
    return suhmo_readback(slot, st, out);
}

