// suhmo_step.hip -- the caller of the head solve, device resident: one AmrHydro::timeStepFAS
// (src/AmrHydro.cpp:2254-3460) on one level (suhmo_level_timestep) or on an AMR hierarchy (suhmo_amr_timestep), each whole
// or cut into rank strips: distributed, time-varying or moulin water input, with or without the diffusive term
// (suhmo.diffFactor), explicit or implicit (solver.use_ImplDiff) gap-height update.  "Next rows" of SURVEY.md 8(f).
//
//   [II]  Picard loop (:2477-3235): ghosts of h and b -> grad h (compute_grad_head :1610-1674) ->
//         Re (evaluate_Re_quadratic :1711-1778) -> Qw on faces (evaluate_Qw_ec :1677-1709, COMPUTEQW)
//         -> Qw.grad(h), Qw.grad(zb) (COMPUTESCAPROD + EdgeToCell :2954-2979) -> melt rate
//         (Calc_meltingRate :2174-2252) -> RHS_h (:3031-3078) -> SolveForHead_nl -> Picard test
//   [III] the same chain with the new head, CalcRHS_gapHeightFAS (:2069-2171), forward Euler (:3406)
// Head = PHI, gap height = B: both stay in HBM across Picard iterations and timesteps; per step
// the host sees two scalars per Picard iteration (max head, max relative change).
// On a rank strip (suhmo_level_desc.j0 / ny_global) the same step runs on every rank: wherever the reference exchanges
// a field (b, mR, grad h, RHS halos for the redundant halo-row relaxation) the strip's exchange hook is called, the
// Picard test is MAX all-reduced; the moulin integrals are evaluated redundantly over the whole domain (analytic
// integrand, no data), so the result does not depend on the partition bit for bit.
#include "suhmo_hier.h"
#include <cmath>

// Qw on x- and y-faces: B_ec, Re_ec by CellToEdge (half*(cell + lower cell)), grad h by NEWMACGRAD
// with the physical BC applied on the fly, COMPUTEQW (src/AmrHydroF.ChF:137-150)
__device__ __forceinline__ double face_grad(const DV &v, const double *__restrict__ phi, const double *__restrict__ mk,
                                            int i, int j, int dir, int hasMask)
{
    // face (i,j) of direction dir lies between cell (i,j) and (i-1,j) / (i,j-1)
    double hi, lo;
    if (dir == 0) {
        if (i < v.nx) { int idx = cidx(v, i, j); hi = phi[idx]; lo = phiW(v, phi, idx, i, hi, false); }
        else { int idx = cidx(v, v.nx - 1, j); lo = phi[idx]; hi = phiE(v, phi, idx, v.nx - 1, lo, false); }
    } else {
        if (j < v.ny) { int idx = cidx(v, i, j); hi = phi[idx]; lo = phiS(v, phi, idx, j, hi, false); }
        else { int idx = cidx(v, i, v.ny - 1); lo = phi[idx]; hi = phiN(v, phi, idx, v.ny - 1, lo, false); }
    }
    double g = (dir == 0 ? v.fdx : v.fdy) * (hi - lo);
    if (hasMask) {
        int idx = cidx(v, i, j), idm = dir == 0 ? idx - 1 : idx - v.P;
        if (mk[idx] < 1e-6 || mk[idm] < 1e-6) g = 0.0;
    }
    return g;
}
// gradient of the bed on a face: plain stored ghosts (zb is caller data over the ghosted level)
__device__ __forceinline__ double face_grad_zb(const DV &v, const double *__restrict__ zb, const double *__restrict__ mk,
                                               int i, int j, int dir, int hasMask)
{
    int idx = cidx(v, i, j), idm = dir == 0 ? idx - 1 : idx - v.P;
    double g = (dir == 0 ? v.fdx : v.fdy) * (zb[idx] - zb[idm]);
    if (hasMask && (mk[idx] < 1e-6 || mk[idm] < 1e-6)) g = 0.0;
    return g;
}

__device__ __forceinline__ void d_qw_faces(const DV &v, const FP &fp, suhmo_phys_t ph)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > v.nx || j > v.ny) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI], *__restrict__ B = fp.f[SUHMO_F_B], *__restrict__ Re = fp.f[SUHMO_F_RE];
    const double *__restrict__ mk = fp.f[SUHMO_F_MASK];
    int idx = cidx(v, i, j);
    if (j < v.ny) {
        double b = 0.5 * (B[idx] + B[idx - 1]), re = 0.5 * (Re[idx] + Re[idx - 1]);
        double g = face_grad(v, phi, mk, i, j, 0, ph.use_mask_gradients);
        double num_q = -(b * b * b * ph.grav * g);
        double denom_q = 12.0 * ph.nu * (1.0 + ph.omega * re);
        fp.f[SUHMO_F_QWX][idx] = num_q / denom_q;
    }
    if (i < v.nx) {
        double b = 0.5 * (B[idx] + B[idx - v.P]), re = 0.5 * (Re[idx] + Re[idx - v.P]);
        double g = face_grad(v, phi, mk, i, j, 1, ph.use_mask_gradients);
        double num_q = -(b * b * b * ph.grav * g);
        double denom_q = 12.0 * ph.nu * (1.0 + ph.omega * re);
        fp.f[SUHMO_F_QWY][idx] = num_q / denom_q;
    }
}
__global__ __launch_bounds__(256) void k_qw_faces(DV v, FP fp, suhmo_phys_t ph)
{
    d_qw_faces(v, fp, ph);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_qw_faces_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph)
{
    d_qw_faces(vt[blockIdx.z], ft[blockIdx.z], ph);
}

// MODE 0: melt rate + RHS_h (Picard iteration).  MODE 1: melt rate + gap-height RHS + forward Euler.
template <int MODE>
__device__ __forceinline__ void d_melt(const DV &v, const FP &fp, suhmo_phys_t ph, suhmo_model_params_t mp, double dt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI], *__restrict__ zb = fp.f[SUHMO_F_ZB], *__restrict__ mk = fp.f[SUHMO_F_MASK];
    const double *__restrict__ qx = fp.f[SUHMO_F_QWX], *__restrict__ qy = fp.f[SUHMO_F_QWY];
    const int idx = cidx(v, i, j), hm = ph.use_mask_gradients;
    // COMPUTESCAPROD on the four faces of the cell, EdgeToCell
    double t1w = qx[idx] * face_grad(v, phi, mk, i, j, 0, hm), t1e = qx[idx + 1] * face_grad(v, phi, mk, i + 1, j, 0, hm);
    double t1s = qy[idx] * face_grad(v, phi, mk, i, j, 1, hm), t1n = qy[idx + v.P] * face_grad(v, phi, mk, i, j + 1, 1, hm);
    double t2w = qx[idx] * face_grad_zb(v, zb, mk, i, j, 0, hm), t2e = qx[idx + 1] * face_grad_zb(v, zb, mk, i + 1, j, 0, hm);
    double t2s = qy[idx] * face_grad_zb(v, zb, mk, i, j, 1, hm), t2n = qy[idx + v.P] * face_grad_zb(v, zb, mk, i, j + 1, 1, hm);
    double t0 = 0.5 * (t1w + t1e), t1 = 0.5 * (t1s + t1n), u0 = 0.5 * (t2w + t2e), u1 = 0.5 * (t2s + t2n);
    // Calc_meltingRate (src/AmrHydro.cpp:2210-2244)
    const double h = phi[idx], b = fp.f[SUHMO_F_B][idx], Pi = fp.f[SUHMO_F_PI][idx], im = mk[idx];
    double Pw = mp.gravity * mp.rho_w * (h - zb[idx]);
    double sca_prod = 0.0;
    if (mp.basal_friction) sca_prod = 20. * 20. * mp.ub0 * fabs(Pi - Pw) * mp.ub0;
    double abs_QPw = t0 + t1 - (u0 + u1);
    if ((abs_QPw < 0) && (b < 1e-6)) abs_QPw = 0.0;
    double m = mp.G + sca_prod - mp.rho_w * mp.gravity * (t0 + t1) + mp.ct * mp.cw * mp.rho_w * mp.rho_w * mp.gravity * abs_QPw;
    m = m / mp.L;
    m = fmax(m, 0.0);
    if (im < 0.0) m = 0.0;
    fp.f[SUHMO_F_MR][idx] = m;
    fp.f[SUHMO_F_PW][idx] = Pw;
    const double ub_norm = sqrt(mp.ub0 * mp.ub0 + mp.ub1 * mp.ub1);
    if (MODE == 0) {                                           // RHS_h, :3044-3077
        double rho_coef = (1.0 / mp.rho_w - 1.0 / mp.rho_i);
        if (mp.head_melt_off) rho_coef *= 0.0;                 // run-state setting of the reference's committed tables (suhmo_hip.h)
        double r = m * rho_coef;
        if (b < mp.br) r -= ub_norm * (mp.br - b) / mp.lr;
        if (mp.use_moulin_source) r += fp.f[SUHMO_F_MSRC][idx] * mp.ramp + mp.distributed_input;   // :3060-3066
        else r += (im > 0.0) ? mp.distributed_input : 0.0;     // distributed input where there is ice, :2871-2875
        if (mp.diffFactor != 0.0) r -= mp.diffFactor * fp.f[SUHMO_F_DTERM][idx];     // :3071
        if (im < 0.0) r = 0.0;
        fp.f[SUHMO_F_RHS][idx] = r;
    } else {                                                   // CalcRHS_gapHeightFAS :2113-2168 + forward Euler :3406
        double RHS = m * (1.0 / mp.rho_i), RHS_A = RHS, RHS_B = 0.0, cd = 0.0;
        if ((im < 0.0) && mp.use_mask_rhs_b) { RHS = 0.0; if (mp.use_impl_diff) RHS = b; }
        else {
            if (b < mp.br) { RHS += ub_norm * (mp.br - b) / mp.lr; RHS_B = ub_norm * (mp.br - b) / mp.lr; }
            double PimPw = Pi - Pw, AbsPimPw = fabs(PimPw);
            if (ph.cutOffbr > b) RHS -= ph.A * (AbsPimPw * AbsPimPw) * PimPw * b * (1.0 - (ph.cutOffbr - b) / ph.cutOffbr);
            else if (ph.maxOffbr < b) RHS -= ph.A * (AbsPimPw * AbsPimPw) * PimPw * b * (1.0 - (ph.maxOffbr - b) / ph.maxOffbr);
            else RHS -= ph.A * (AbsPimPw * AbsPimPw) * PimPw * b;
            if (!mp.use_impl_diff && mp.diffFactor != 0.0) RHS += mp.diffFactor * fp.f[SUHMO_F_DTERM][idx];   // :2145,:2152,:2159
            cd = RHS_A / (RHS_A + RHS_B);
            if (mp.use_impl_diff) RHS = b + dt * RHS;          // :2165
        }
        fp.f[SUHMO_F_CD][idx] = cd;
        if (mp.use_impl_diff) fp.f[SUHMO_F_RES][idx] = RHS;    // right-hand side of the implicit solve (RES is free here)
        else fp.f[SUHMO_F_B][idx] = RHS * dt + b;              // old b == b: the gap height is untouched during [II]
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_melt(DV v, FP fp, suhmo_phys_t ph, suhmo_model_params_t mp, double dt)
{
    d_melt<MODE>(v, fp, ph, mp, dt);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
template <int MODE>
__global__ __launch_bounds__(256) void k_melt_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph, suhmo_model_params_t mp, double dt)
{
    d_melt<MODE>(vt[blockIdx.z], ft[blockIdx.z], ph, mp, dt);
}

// run-state setting freeze_icefree_gap (suhmo_hip.h): cells without ice keep their gap height through SolveForGap_nl -- the solved
// value of such a cell is replaced by the old one before the solution is copied back (valid cells; the ghosts are refilled afterwards)
__device__ __forceinline__ void d_keep_icefree(const DV &v, const double *__restrict__ mk, const double *__restrict__ bold, double *__restrict__ sol)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    if (mk[idx] < 0.0) sol[idx] = bold[idx];
}
__global__ __launch_bounds__(256) void k_keep_icefree(DV v, const double *__restrict__ mk, const double *__restrict__ bold, double *__restrict__ sol)
{
    d_keep_icefree(v, mk, bold, sol);
}
__global__ __launch_bounds__(256) void k_keep_icefree_m(const DV *__restrict__ vt, const FP *__restrict__ fh, const FP *__restrict__ fg)
{
    d_keep_icefree(vt[blockIdx.z], fh[blockIdx.z].f[SUHMO_F_MASK], fh[blockIdx.z].f[SUHMO_F_B], fg[blockIdx.z].f[SUHMO_F_PHI]);
}
static int keep_icefree(suhmo_level *L, suhmo_level *G, hipStream_t st)
{
    Depth &D = L->d[0];
    hipLaunchKernelGGL(k_keep_icefree, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp.f[SUHMO_F_MASK], D.fp.f[SUHMO_F_B],
                       G->d[0].fp.f[SUHMO_F_PHI]);
    HIPCHK(hipGetLastError());
    return 0;
}

// Picard convergence test, :3169-3185
struct Excl { int i0, j0, i1, j1; };       // local cells [i0, i1) x [j0, j1) do not count (covered by a finer level)
static int exchange1(suhmo_level *L, int f, hipStream_t st) { return suhmo_exchange_list(L, 0, &f, 1, st); }
// Both numbers of the Picard test in one pass and one read-back: max h and max |h_lagged - h|.  The reference's
// max |(h_lagged - h) / maxHead| is the second divided by |maxHead| afterwards: a correctly rounded division by a fixed
// divisor is monotone and sign-symmetric, so the maximum of the quotients is the quotient of the maximum, bit for bit.
__global__ __launch_bounds__(256) void k_picard2_partial(DV v, const double *__restrict__ h, const double *__restrict__ hl,
                                                         double *__restrict__ partial, Excl ex, const double *__restrict__ cover = nullptr)
{
    __shared__ double sm0[256], sm1[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double a0 = -1.0e300, a1 = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) {
            if (i >= ex.i0 && i < ex.i1 && j >= ex.j0 && j < ex.j1) continue;
            int idx = cidx(v, i, j);
            if (cover && cover[idx] != 0.0) continue;          // hierarchy of box unions: SUHMO_F_COVER
            a0 = fmax(a0, h[idx]);
            a1 = fmax(a1, fabs(hl[idx] - h[idx]));
        }
    sm0[tid] = a0; sm1[tid] = a1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sm0[tid] = fmax(sm0[tid], sm0[tid + s]); sm1[tid] = fmax(sm1[tid], sm1[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { int b = blockIdx.y * gridDim.x + blockIdx.x; partial[2 * b] = sm0[0]; partial[2 * b + 1] = sm1[0]; }
}
__global__ void k_max2_final(const double *__restrict__ partial, int n, double *__restrict__ out, HostSlot hs)
{
    __shared__ double sm0[256], sm1[256];
    int tid = threadIdx.x;
    double a0 = -1.0e300, a1 = 0.0;
    for (int k = tid; k < n; k += 256) { a0 = fmax(a0, partial[2 * k]); a1 = fmax(a1, partial[2 * k + 1]); }
    sm0[tid] = a0; sm1[tid] = a1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sm0[tid] = fmax(sm0[tid], sm0[tid + s]); sm1[tid] = fmax(sm1[tid], sm1[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { out[0] = sm0[0]; out[1] = sm1[0]; if (hs.val) hs.val[1] = sm1[0]; suhmo_publish(hs, sm0[0]); }
}
// max h and max |hl - h| over the level's cells (local to the rank)
// over_ranks: the maxima over all ranks of the level's strip partition (computeMax; one MAX all-reduce of both values)
static int picard_maxima(suhmo_level *L, const double *h, const double *hl, double *maxh, double *maxd, hipStream_t st,
                         Excl ex = Excl{0, 0, 0, 0}, const double *cover = nullptr, bool over_ranks = false)
{
    Depth &D = L->d[0];
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    hipLaunchKernelGGL(k_picard2_partial, grd, dim3(64, 4), 0, st, D.v, h, hl, L->scratch + 2, ex, cover);
    hipLaunchKernelGGL(k_max2_final, dim3(1), dim3(256), 0, st, L->scratch + 2, (int)(grd.x * grd.y), L->scratch, over_ranks ? suhmo_reduce_slot(L) : suhmo_host_slot(L));
    if (over_ranks) return suhmo_reduce_finish(L, st, 2, 0, maxh, maxd);
    return suhmo_readback(L, st, maxh, maxd);
}
static inline double picard_quotient(double maxd, double maxHead) { return maxd == 0.0 ? 0.0 : maxd / fabs(maxHead); }

// grad h (cell centred, ghosted) and Re on the ghosted level: reuses the WFlx_level kernels of
// suhmo_bcoef.hip (identical arithmetic: NEWMACGRAD + EdgeToCell + ExtrapGhostCells + COMPUTERE)
int suhmo_grad_re(suhmo_level *L, int depth, hipStream_t st);      // suhmo_bcoef.hip
int suhmo_grad_cc(suhmo_level *L, int depth, hipStream_t st);
int suhmo_re_cells(suhmo_level *L, int depth, hipStream_t st);
int suhmo_bcoef_faces(suhmo_level *L, int depth, hipStream_t st);
int suhmo_copy_ghosts(suhmo_level *L, int depth, int field, hipStream_t st);

static int lagged_chain(suhmo_level *L, hipStream_t st)
{
    Depth &D = L->d[0];
    int rc = suhmo_grad_re(L, 0, st); if (rc) return rc;
    hipLaunchKernelGGL(k_qw_faces, dim3((D.v.nx + 1 + 63) / 64, (D.v.ny + 1 + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- diffusion of the gap height (suhmo.diffFactor != 0)
// ghosts of the melt rate: exchange + ExtrapGhostCells (:2513,:2526); only the edges are read (CellToEdge)
__device__ __forceinline__ void d_extrap_ghosts(const DV &v, double *__restrict__ g)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * v.ny) {
        int side = t / v.ny, j = t % v.ny;
        if (v.cfx[side]) return;                             // coarse-fine side: interpolated
        if (side == 0) { int idx = cidx(v, 0, j); g[idx - 1] = v.per[0] ? g[idx + v.nx - 1] : 2.0 * g[idx] - g[idx + 1]; }
        else { int idx = cidx(v, v.nx - 1, j); g[idx + 1] = v.per[0] ? g[idx - (v.nx - 1)] : 2.0 * g[idx] - g[idx - 1]; }
        return;
    }
    t -= 2 * v.ny;
    if (t < 2 * v.nx) {
        int side = t / v.nx, i = t % v.nx;
        if (v.ext[side]) return;                             // rank boundary: exchanged; coarse-fine side: interpolated
        if (side == 0) { int idx = cidx(v, i, 0); g[idx - v.P] = v.per[1] ? g[idx + (v.ny - 1) * v.P] : 2.0 * g[idx] - g[idx + v.P]; }
        else { int idx = cidx(v, i, v.ny - 1); g[idx + v.P] = v.per[1] ? g[idx - (v.ny - 1) * v.P] : 2.0 * g[idx] - g[idx - v.P]; }
    }
}
__global__ void k_extrap_ghosts(DV v, double *__restrict__ g)
{
    d_extrap_ghosts(v, g);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_extrap_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field)
{
    d_extrap_ghosts(vt[blockIdx.z], ft[blockIdx.z].f[field]);
}
// dCoeff: CellToEdge(mR), CellToEdge(b), setup_iceMask_EC, COMPUTEDCOEFF (src/AmrHydro.cpp:1831-1862, ...F.ChF:241-265)
__device__ __forceinline__ void d_dcoef_faces(const DV &v, const FP &fp, suhmo_phys_t ph, double rho_i)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > v.nx || j > v.ny) return;
    const double *__restrict__ B = fp.f[SUHMO_F_B], *__restrict__ mR = fp.f[SUHMO_F_MR], *__restrict__ mk = fp.f[SUHMO_F_MASK];
    int idx = cidx(v, i, j);
    for (int dir = 0; dir < 2; dir++) {
        if ((dir == 0 && j >= v.ny) || (dir == 1 && i >= v.nx)) continue;
        int im = dir == 0 ? idx - 1 : idx - v.P;
        double m = mk[idx], mm1 = mk[im], mec;
        if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; else mec = 0.0;
        int f = dir == 0 ? i + v.i0 : j + v.j0, fhi = dir == 0 ? v.nxg : v.nyg;   // domain faces (global index on a strip / patch)
        if (f == 0 || f == fhi) mec = 0.0;
        double bec = 0.5 * (B[idx] + B[im]), mrec = 0.5 * (mR[idx] + mR[im]), d;
        if (mec < 0.0 && ph.cutOffB > 0) d = 0.0; else d = fmax(bec * mrec / rho_i, 5.0e-6);
        fp.f[dir == 0 ? SUHMO_F_DCX : SUHMO_F_DCY][idx] = d;
    }
}
__global__ __launch_bounds__(256) void k_dcoef_faces(DV v, FP fp, suhmo_phys_t ph, double rho_i)
{
    d_dcoef_faces(v, fp, ph, rho_i);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_dcoef_faces_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph, double rho_i)
{
    d_dcoef_faces(vt[blockIdx.z], ft[blockIdx.z], ph, rho_i);
}
// COMPUTEDIFTERM2D (src/AmrHydroF.ChF:289-343) of the gap height with its copied ghosts
__device__ __forceinline__ void d_difterm(const DV &v, FP fp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    const double *__restrict__ B = fp.f[SUHMO_F_B], *__restrict__ dx_ = fp.f[SUHMO_F_DCX], *__restrict__ dy_ = fp.f[SUHMO_F_DCY];
    int idx = cidx(v, i, j);
    const double dxinv0 = 1.0 / (v.dx * v.dx), dxinv1 = 1.0 / (v.dy * v.dy);
    fp.f[SUHMO_F_DTERM][idx] =
        (dx_[idx + 1] * (B[idx + 1] - B[idx]) * dxinv0 - dx_[idx] * (B[idx] - B[idx - 1]) * dxinv0
         + dy_[idx + v.P] * (B[idx + v.P] - B[idx]) * dxinv1 - dy_[idx] * (B[idx] - B[idx - v.P]) * dxinv1);
}
__global__ __launch_bounds__(256) void k_difterm(DV v, FP fp)
{
    d_difterm(v, fp);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_difterm_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    d_difterm(vt[blockIdx.z], ft[blockIdx.z]);
}
static int diffusion_terms(suhmo_level *L, const suhmo_model_params_t *mp, hipStream_t st)
{
    Depth &D = L->d[0];
    for (int f : {SUHMO_F_DCX, SUHMO_F_DCY, SUHMO_F_DTERM}) if (!suhmo_field(L, 0, f)) { suhmo_set_error("field allocation failed"); return -2; }
    int n = 2 * D.v.ny + 2 * D.v.nx;
    int rc = exchange1(L, SUHMO_F_MR, st); if (rc) return rc;          // levelmR.exchange() :2513
    hipLaunchKernelGGL(k_extrap_ghosts, dim3((n + 255) / 256), dim3(256), 0, st, D.v, D.fp.f[SUHMO_F_MR]);
    hipLaunchKernelGGL(k_dcoef_faces, dim3((D.v.nx + 1 + 63) / 64, (D.v.ny + 1 + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, L->ph, mp->rho_i);
    hipLaunchKernelGGL(k_difterm, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp);
    HIPCHK(hipGetLastError());
    return 0;
}
// SolveForGap_nl (src/AmrHydro.cpp:593-662): (1 - dt diffFactor div(D grad)) b = RES on a second level handle with the
// linear operator (alpha = 1, aCoef = 1, beta = dt diffFactor, bCoef = D, no nonlinear term), FixedNeumBCFill = Neumann 0.
// [Chombo] VCAMRPoissonOp2 / AMRMultiGrid are not in the reference's tree: the cycle is the FAS cycle of suhmo_fas.hip,
// which for a linear operator converges to the same solution (oracle/time_loop.c:solve_gap_implicit does the same).
// the second handle of a level (created on first use / when dt changes) loaded with this step's data
static int gap_level_prepare(suhmo_level *L, const suhmo_model_params_t *mp, double dt, hipStream_t st)
{
    Depth &D = L->d[0];
    if (L->gap && L->gap_dt != dt) {                       // a new time step size: only beta = dt diffFactor changes
        int rc = suhmo_level_set_alpha_beta(L->gap, 1.0, dt * mp->diffFactor); if (rc) return rc;
        L->gap_dt = dt;
    }
    if (!L->gap) {
        suhmo_level_desc_t d = L->desc;
        d.boxes = L->boxes.data(); d.nbox = (int)(L->boxes.size() / 4);
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { d.bc.type[a][b] = 1; d.bc.value[a][b] = 0.0; }
        d.phys.use_NL = 0; d.alpha = 1.0; d.beta = dt * mp->diffFactor;
        int rc = suhmo_level_create(&L->gap, &d); if (rc) return rc;
        L->gap_dt = dt;
        if ((rc = suhmo_level_set_value(L->gap, 0, SUHMO_F_ACOEF, 1.0, (suhmo_stream_t)st))) return rc;       // aCoeff_GH :1820-1828
    }
    suhmo_level *G = L->gap;
    G->ex = L->ex; G->ar = L->ar; G->ar2 = L->ar2; G->ard = L->ard; G->user = L->user; G->ex_begin = L->ex_begin; G->ex_end = L->ex_end; G->ipc = L->ipc;
    if (G->ag != L->ag || G->ag_user != L->ag_user || G->agg_min_cells != L->agg_min_cells) {                // ... and the same agglomeration
        G->ag = L->ag; G->ag_user = L->ag_user; G->agg_min_cells = L->agg_min_cells;
        int rca = suhmo_agg_setup(G); if (rca) return rca;
    }       // same strip, same neighbours
    Depth &GD = G->d[0];
    if (GD.elems != D.elems) { suhmo_set_error("internal: gap level geometry"); return -4; }
    const size_t bytes = D.elems * sizeof(double);
    HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_PHI], D.fp.f[SUHMO_F_B], bytes, hipMemcpyDeviceToDevice, st));      // initial guess = b :3382-3385
    HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_RHS], D.fp.f[SUHMO_F_RES], bytes, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_BX], D.fp.f[SUHMO_F_DCX], bytes, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_BY], D.fp.f[SUHMO_F_DCY], bytes, hipMemcpyDeviceToDevice, st));
    GD.phi_fresh = 0;
    static const int halo_fields[] = {SUHMO_F_RHS, SUHMO_F_ACOEF, SUHMO_F_BX, SUHMO_F_BY};
    return suhmo_exchange_list(G, 0, halo_fields, 4, st);
}
static void gap_solver_params(suhmo_solver_params_t &sp, int cur_step)
{
    sp.num_smooth = 2; sp.num_bottom = 4; sp.max_iter = 100; sp.iter_min = 2; sp.imin = cur_step < 50 ? 10 : 5;
    sp.eps = 1.0e-7; sp.hang = 1.0e-6; sp.norm_thresh = 1.0e-7; sp.bcoeff_otf = 0; sp.max_depth = -1;
}
static int solve_gap_implicit(suhmo_level *L, const suhmo_model_params_t *mp, double dt, int cur_step, hipStream_t st)
{
    int rc = gap_level_prepare(L, mp, dt, st); if (rc) return rc;
    suhmo_level *G = L->gap;
    rc = suhmo_level_build_mg_coefficients(G, (suhmo_stream_t)st); if (rc) return rc;    // coarse D = average of the fine faces
    suhmo_solver_params_t sp;
    gap_solver_params(sp, cur_step);
    if ((rc = suhmo_level_solve(G, &sp, nullptr, nullptr, (suhmo_stream_t)st))) return rc;
    if (mp->freeze_icefree_gap && (rc = keep_icefree(L, G, st))) return rc;
    HIPCHK(hipMemcpyAsync(L->d[0].fp.f[SUHMO_F_B], G->d[0].fp.f[SUHMO_F_PHI], L->d[0].elems * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int suhmo_level_timestep(suhmo_level_t *L, const suhmo_model_params_t *mp, double dt, int cur_step,
                                    int *picard_iters, int *vcycles, suhmo_stream_t s)
{
    SUHMO_TIME("AmrHydro::timeStepFAS");
    ARG(L && mp); ARG(dt > 0 && cur_step >= 1);
    if (mp->use_impl_diff && mp->diffFactor == 0.0) { suhmo_set_error("use_ImplDiff with diffFactor = 0"); return -1; }
    Depth &D = L->d[0];
    if (L->desc.nx_global > 0 || (D.v.ext[0] && !D.v.rk[0]) || (D.v.ext[1] && !D.v.rk[1])) { suhmo_set_error("timestep on an AMR patch is not built yet"); return -5; }
    if ((D.v.ext[0] || D.v.ext[1]) && !(L->ex && L->ar)) { suhmo_set_error("timestep on a rank strip needs the exchange hooks (suhmo_level_attach_rccl / suhmo_level_set_hooks)"); return -1; }
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    if (mp->use_moulin_source && !L->d[0].fp.f[SUHMO_F_MSRC]) { suhmo_set_error("use_moulin_source without suhmo_level_moulin_source"); return -1; }
    static const int need[] = {SUHMO_F_MR, SUHMO_F_PW, SUHMO_F_QWX, SUHMO_F_QWY, SUHMO_F_HLAG, SUHMO_F_CD,
                               SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_RE};
    for (int f : need) if (!suhmo_field(L, 0, f)) { suhmo_set_error("field allocation failed"); return -2; }
    int rc;
    // [I] ghosts of b (exchange + CopyGhostCells, :2385,:2429); ghosts of h are evaluated on the fly
    if ((rc = suhmo_copy_ghosts(L, 0, SUHMO_F_B, st))) return rc;
    if ((rc = exchange1(L, SUHMO_F_B, st))) return rc;
    // MGnewOp coarsening of B (+ static Pi, zb, mask, aCoef): once per step, b does not change in [II]
    if ((rc = suhmo_build_mg_coefficients(L, false, (hipStream_t)s))) return rc;    // bCoef: re-averaged by every V-cycle (bcoeff_otf)
    suhmo_solver_params_t sp;                                      // SolveForHead_nl, :737-762
    sp.num_smooth = 4; sp.num_bottom = 16; sp.max_iter = 100; sp.iter_min = 2; sp.imin = 5;
    sp.eps = 1.0e-7; sp.hang = 0.01; sp.norm_thresh = 1.0e-7; sp.bcoeff_otf = 1; sp.max_depth = -1;
    if (cur_step < 50) { sp.num_bottom = 10; sp.eps = 1.0e-10; sp.hang = 0.0001; sp.imin = 20; }
    const dim3 blk(64, 4), grd((D.v.nx + 63) / 64, (D.v.ny + 3) / 4);
    bool converged = false;
    int ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {                                           // [II]
        HIPCHK(hipMemcpyAsync(D.fp.f[SUHMO_F_HLAG], D.fp.f[SUHMO_F_PHI], D.elems * sizeof(double), hipMemcpyDeviceToDevice, st));
        if ((rc = lagged_chain(L, st))) return rc;
        if ((rc = suhmo_bcoef_faces(L, 0, st))) return rc;                          // aCoeff_bCoeff :3087-3102
        if (mp->diffFactor != 0.0 && (rc = diffusion_terms(L, mp, st))) return rc;   // lagged melt rate :2548-2551, :2982-2992
        hipLaunchKernelGGL(k_melt<0>, grd, blk, 0, st, D.v, D.fp, L->ph, *mp, dt);
        HIPCHK(hipGetLastError());
        if ((rc = exchange1(L, SUHMO_F_RHS, st))) return rc;        // rank strips relax their halo rows redundantly
        int it = 0;
        if ((rc = suhmo_level_solve(L, &sp, &it, nullptr, s))) return rc;
        nv += it;
        double maxHead = 0.0, maxd = 0.0, res = 0.0;
        if ((rc = picard_maxima(L, D.fp.f[SUHMO_F_PHI], D.fp.f[SUHMO_F_HLAG], &maxHead, &maxd, st, Excl{0, 0, 0, 0}, nullptr, true))) return rc;   // computeMax over all ranks
        res = picard_quotient(maxd, maxHead);
        if (ite_idx > 100) { suhmo_set_error("does not converge (Picard iterations > 100)"); return -6; }   // :3190-3195
        if (cur_step < 2) { if (res < 0.05 && cur_picard > 2) converged = true; }
        else if (cur_step < 50) { if (res < 0.05) converged = true; }
        else { if (res < mp->eps_picard) converged = true; }
        ite_idx++; cur_picard++;
    }
    // [III]
    if ((rc = lagged_chain(L, st))) return rc;
    hipLaunchKernelGGL(k_melt<1>, grd, blk, 0, st, D.v, D.fp, L->ph, *mp, dt);
    HIPCHK(hipGetLastError());
    if (mp->use_impl_diff && (rc = solve_gap_implicit(L, mp, dt, cur_step, st))) return rc;   // :3425-3439
    if ((rc = suhmo_copy_ghosts(L, 0, SUHMO_F_B, st))) return rc;  // :3419-3420 / :3451-3452
    if ((rc = exchange1(L, SUHMO_F_B, st))) return rc;
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles) *vcycles = nv;
    return 0;
}


// ------------------------------------------------------------------ the time step on an AMR hierarchy
// oracle/amr_step.c: every level runs the phases above on its own rectangle; in between PiecewiseLinearFillPatch of the
// coarse-fine ghosts of b, mR, Re (:2373-2380, :2499-2507, :2711-2719), QuadCFInterp of h (inside compGradientMAC) and of
// the cell-centred gradient (:1650-1656), SolveForHead_nl over all levels, CoarseAverage of h (:3138-3141) and the
// Picard test over the cells no finer level covers (:3169-3185); the gap height by forward Euler level by level, or by
// SolveForGap_nl over a second hierarchy of handles (alpha = 1, beta = dt diffFactor, bCoef = D, as solve_gap_implicit).
// Rank strips: a rank holds of every level the rows of its own slab (lv[l] = NULL where the patch does not reach it);
// it still mirrors, on level l-1, the halo demand level l puts there, so that all ranks of a level's communicator issue
// the same sequence of exchanges (as the AMR cycle does, suhmo_amr.hip).
static int amr_chain(suhmo_level_t **lv, int l, hipStream_t st)
{
    suhmo_level *L = lv[l], *C = l > 0 ? lv[l - 1] : nullptr;
    suhmo_stream_t s = (suhmo_stream_t)st;
    int rc;
    if (!L) return C ? suhmo_ensure_phi_halo(C, 0, 1, st) : 0;
    Depth &D = L->d[0];
    if (C && (rc = suhmo_amr2_cf_interp(C, L, SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc;
    if ((rc = suhmo_grad_cc(L, 0, st))) return rc;
    if (C) {
        if ((rc = suhmo_amr2_cf_interp(C, L, SUHMO_F_GRADX, SUHMO_F_GRADX, s))) return rc;
        if ((rc = suhmo_amr2_cf_interp(C, L, SUHMO_F_GRADY, SUHMO_F_GRADY, s))) return rc;
    }
    if ((rc = suhmo_re_cells(L, 0, st))) return rc;
    if (C && (rc = suhmo_amr2_pwl_fill(C, L, SUHMO_F_RE, SUHMO_F_RE, s))) return rc;
    hipLaunchKernelGGL(k_qw_faces, dim3((D.v.nx + 1 + 63) / 64, (D.v.ny + 1 + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
static Excl covered_by(suhmo_level_t **lv, int nlev, int l)
{
    if (l >= nlev - 1 || !lv[l + 1]) return Excl{0, 0, 0, 0};
    const DV &vf = lv[l + 1]->d[0].v, &v = lv[l]->d[0].v;
    return Excl{vf.i0 / 2 - v.i0, vf.j0 / 2 - v.j0, (vf.i0 + vf.nx) / 2 - v.i0, (vf.j0 + vf.ny) / 2 - v.j0};
}
// ghosts of b of every level: PiecewiseLinearFillPatch on coarse-fine sides (from the coarser level's current b, whose halo
// rows were exchanged just before), copies on domain sides, halo rows on rank boundaries
static int amr_gap_ghosts(suhmo_level_t **lv, int nlev, int l, hipStream_t st)
{
    int rc;
    if (!lv[l]) return 0;
    if (l > 0 && (rc = suhmo_amr2_pwl_fill(lv[l - 1], lv[l], SUHMO_F_B, SUHMO_F_B, (suhmo_stream_t)st))) return rc;
    if ((rc = suhmo_copy_ghosts(lv[l], 0, SUHMO_F_B, st))) return rc;
    return exchange1(lv[l], SUHMO_F_B, st);
}
int suhmo_amr_check_hierarchy(suhmo_level_t **lv, int nlev);      // suhmo_amr.hip

extern "C" int suhmo_amr_timestep(suhmo_level_t **lv, int nlev, const suhmo_model_params_t *mp, double dt, int cur_step,
                                  int *picard_iters, int *vcycles, suhmo_stream_t s)
{
    SUHMO_TIME("AmrHydro::timeStepFAS");
    ARG(lv && mp && nlev >= 1 && nlev <= 8 && lv[0]); ARG(dt > 0 && cur_step >= 1);
    if (mp->use_impl_diff && mp->diffFactor == 0.0) { suhmo_set_error("use_ImplDiff with diffFactor = 0"); return -1; }
    int rc = suhmo_amr_check_hierarchy(lv, nlev); if (rc) return rc;
    bool strips = false;
    for (int l = 0; l < nlev; l++) {
        if (!lv[l]) { strips = true; continue; }
        const DV &v = lv[l]->d[0].v;
        if (v.rk[0] || v.rk[1]) {
            strips = true;
            if (!(lv[l]->ex && lv[l]->ar)) { suhmo_set_error("time step on rank strips needs the exchange hooks on every level"); return -1; }
        }
        if (mp->use_moulin_source && !lv[l]->d[0].fp.f[SUHMO_F_MSRC]) { suhmo_set_error("use_moulin_source without a moulin source term (SUHMO_F_MSRC)"); return -1; }
    }
    HIPCHK(hipSetDevice(lv[0]->device));
    hipStream_t st = (hipStream_t)s;
    static const int need[] = {SUHMO_F_MR, SUHMO_F_PW, SUHMO_F_QWX, SUHMO_F_QWY, SUHMO_F_HLAG, SUHMO_F_CD, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_RE};
    for (int l = 0; l < nlev; l++) if (lv[l]) for (int f : need) if (!suhmo_field(lv[l], 0, f)) { suhmo_set_error("field allocation failed"); return -2; }
    // [I]
    for (int l = 0; l < nlev; l++) if ((rc = amr_gap_ghosts(lv, nlev, l, st))) return rc;
    if ((rc = suhmo_build_mg_coefficients(lv[0], false, (hipStream_t)s))) return rc;    // bCoef: re-averaged by every V-cycle (bcoeff_otf)
    suhmo_solver_params_t sp;
    sp.num_smooth = 4; sp.num_bottom = 16; sp.max_iter = 100; sp.iter_min = 2; sp.imin = 5;
    sp.eps = 1.0e-7; sp.hang = 0.01; sp.norm_thresh = 1.0e-7; sp.bcoeff_otf = 1; sp.max_depth = -1;
    if (cur_step < 50) { sp.num_bottom = 10; sp.eps = 1.0e-10; sp.hang = 0.0001; sp.imin = 20; }
    bool converged = false;
    int ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {
        for (int l = 0; l < nlev; l++) {
            if (!lv[l]) continue;
            Depth &D = lv[l]->d[0];
            if ((rc = exchange1(lv[l], SUHMO_F_MR, st))) return rc;                    // levelmR.exchange() :2513 (and the stencil of the fill below)
            if (l > 0 && (rc = suhmo_amr2_pwl_fill(lv[l - 1], lv[l], SUHMO_F_MR, SUHMO_F_MR, s))) return rc;
            if ((rc = amr_gap_ghosts(lv, nlev, l, st))) return rc;
            HIPCHK(hipMemcpyAsync(D.fp.f[SUHMO_F_HLAG], D.fp.f[SUHMO_F_PHI], D.elems * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        for (int l = 0; l < nlev; l++) if ((rc = amr_chain(lv, l, st))) return rc;
        for (int l = 0; l < nlev; l++) {
            if (!lv[l]) continue;
            Depth &D = lv[l]->d[0];
            if ((rc = suhmo_bcoef_faces(lv[l], 0, st))) return rc;                      // aCoeff_bCoeff :3087-3102
            if (mp->diffFactor != 0.0 && (rc = diffusion_terms(lv[l], mp, st))) return rc;
            hipLaunchKernelGGL(k_melt<0>, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, lv[l]->ph, *mp, dt);
            HIPCHK(hipGetLastError());
            if ((rc = exchange1(lv[l], SUHMO_F_RHS, st))) return rc;                   // halo rows relaxed redundantly
        }
        int it = 0;
        if (nlev == 1) rc = suhmo_level_solve(lv[0], &sp, &it, nullptr, s);
        else rc = suhmo_amr_solve(lv, nlev, &sp, &it, nullptr, s);
        if (rc) return rc;
        nv += it;
        for (int l = nlev - 1; l > 0; l--) {                                            // CoarseAverage :3138-3141
            if (lv[l]) { if ((rc = suhmo_amr2_average(lv[l - 1], lv[l], SUHMO_F_PHI, SUHMO_F_PHI, s))) return rc; }
            else if (lv[l - 1]) lv[l - 1]->d[0].phi_fresh = 0;                          // changed on the ranks that hold level l
        }
        double maxHead = -1.0e300, maxd = 0.0, res = 0.0;
        for (int l = 0; l < nlev; l++) {
            if (!lv[l]) continue;
            double m = 0.0, d = 0.0;
            if ((rc = picard_maxima(lv[l], lv[l]->d[0].fp.f[SUHMO_F_PHI], lv[l]->d[0].fp.f[SUHMO_F_HLAG], &m, &d, st, covered_by(lv, nlev, l)))) return rc;
            maxHead = std::max(maxHead, m); maxd = std::max(maxd, d);
        }
        if (strips && lv[0]->ar) {                                  // computeMax over all ranks (level 0 reaches every rank)
            if ((rc = lv[0]->ar(lv[0]->user, &maxHead))) return rc;
            if ((rc = lv[0]->ar(lv[0]->user, &maxd))) return rc;
        }
        res = picard_quotient(maxd, maxHead);
        if (ite_idx > 100) { suhmo_set_error("does not converge (Picard iterations > 100)"); return -6; }
        if (cur_step < 2) { if (res < 0.05 && cur_picard > 2) converged = true; }
        else if (cur_step < 50) { if (res < 0.05) converged = true; }
        else { if (res < mp->eps_picard) converged = true; }
        ite_idx++; cur_picard++;
    }
    // [III] level by level: the coarse gap height is already updated when the fine ghost cells are filled
    for (int l = 0; l < nlev; l++) {
        if ((rc = amr_chain(lv, l, st))) return rc;
        if (!lv[l]) continue;
        Depth &D = lv[l]->d[0];
        hipLaunchKernelGGL(k_melt<1>, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, lv[l]->ph, *mp, dt);
        HIPCHK(hipGetLastError());
        if (mp->use_impl_diff) continue;                               // b stays, RES = b + dt RHS
        if ((rc = amr_gap_ghosts(lv, nlev, l, st))) return rc;
    }
    if (mp->use_impl_diff) {                                           // SolveForGap_nl over the hierarchy :3425-3455
        suhmo_level_t *gaps[8];
        for (int l = 0; l < nlev; l++) {
            gaps[l] = nullptr;
            if (!lv[l]) continue;
            if ((rc = gap_level_prepare(lv[l], mp, dt, st))) return rc;
            gaps[l] = lv[l]->gap;
        }
        if ((rc = suhmo_level_build_mg_coefficients(gaps[0], s))) return rc;
        suhmo_solver_params_t spg;
        gap_solver_params(spg, cur_step);
        if (nlev == 1) rc = suhmo_level_solve(gaps[0], &spg, nullptr, nullptr, s);
        else rc = suhmo_amr_solve(gaps, nlev, &spg, nullptr, nullptr, s);
        if (rc) return rc;
        for (int l = 0; l < nlev; l++) {
            if (!lv[l]) continue;
            Depth &D = lv[l]->d[0];
            if (mp->freeze_icefree_gap && (rc = keep_icefree(lv[l], gaps[l], st))) return rc;
            HIPCHK(hipMemcpyAsync(D.fp.f[SUHMO_F_B], gaps[l]->d[0].fp.f[SUHMO_F_PHI], D.elems * sizeof(double), hipMemcpyDeviceToDevice, st));
            if ((rc = amr_gap_ghosts(lv, nlev, l, st))) return rc;
        }
    }
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles) *vcycles = nv;
    return 0;
}

// ------------------------------------------------------------------ the time step on a hierarchy of box unions
// oracle/amr_step_m.c: suhmo_amr_timestep with every level's rectangle replaced by its boxes; after every fill of data ghosts
// the reference's exchange() is the fine-fine copy between the boxes of the level (suhmo_hier.hip).  Every phase of a level
// >= 1 is ONE launch over all its boxes (blockIdx.z = box, device tables of views and field pointers).
namespace {
__global__ __launch_bounds__(256) void k_picard2_partial_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int use_cover, double *__restrict__ partial)
{
    __shared__ double sm0[256], sm1[256];
    const DV &v = vt[blockIdx.z];
    const double *__restrict__ h = ft[blockIdx.z].f[SUHMO_F_PHI], *__restrict__ hl = ft[blockIdx.z].f[SUHMO_F_HLAG];
    const double *__restrict__ cover = use_cover ? ft[blockIdx.z].f[SUHMO_F_COVER] : nullptr;
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double a0 = -1.0e300, a1 = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) {
            int idx = cidx(v, i, j);
            if (cover && cover[idx] != 0.0) continue;
            a0 = fmax(a0, h[idx]);
            a1 = fmax(a1, fabs(hl[idx] - h[idx]));
        }
    sm0[tid] = a0; sm1[tid] = a1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sm0[tid] = fmax(sm0[tid], sm0[tid + s]); sm1[tid] = fmax(sm1[tid], sm1[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; partial[2 * b] = sm0[0]; partial[2 * b + 1] = sm1[0]; }
}
inline dim3 grid_m(const suhmo_multi &m, int ex = 0, int ey = 0) { return dim3((m.maxnx + ex + 63) / 64, (m.maxny + ey + 3) / 4, m.nbox); }
// a level of the hierarchy as a launch target: level 0 = the base handle, level l >= 1 = all its boxes at once
struct LevT { suhmo_level *base; suhmo_multi m; };
int lev_target(suhmo_hier *H, int l, hipStream_t st, LevT &t)
{
    t.base = nullptr;
    if (l == 0) { t.base = suhmo_hier_boxes_(H, 0)[0]; return 0; }
    return suhmo_hier_multi_(H, l, st, &t.m);
}
int hier_chain(suhmo_hier *H, int l, hipStream_t st)
{
    int rc;
    LevT t;
    if ((rc = lev_target(H, l, st, t))) return rc;
    const suhmo_phys_t &ph = suhmo_hier_boxes_(H, l)[0]->ph;
    if ((rc = suhmo_hier_cf_(H, l, SUHMO_F_PHI, SUHMO_F_PHI, st))) return rc;                  // inside compGradientMAC
    if ((rc = suhmo_hier_ff_(H, l, SUHMO_F_PHI, -1, false, st))) return rc;
    if (t.base) rc = suhmo_grad_cc(t.base, 0, st); else rc = suhmo_multi_grad_cc(t.m, ph.use_mask_gradients, st);
    if (rc) return rc;
    if ((rc = suhmo_hier_cf2_(H, l, SUHMO_F_GRADX, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_GRADY, st))) return rc;   // :1650-1659
    if ((rc = suhmo_hier_ff_(H, l, SUHMO_F_GRADX, SUHMO_F_GRADY, true, st))) return rc;
    if (t.base) rc = suhmo_re_cells(t.base, 0, st); else rc = suhmo_multi_re(t.m, ph, st);
    if (rc) return rc;
    if ((rc = suhmo_hier_pwl_(H, l, SUHMO_F_RE, SUHMO_F_RE, st))) return rc;                   // :2711-2721
    if ((rc = suhmo_hier_ff_(H, l, SUHMO_F_RE, -1, true, st))) return rc;
    if (t.base) { Depth &D = t.base->d[0];
        hipLaunchKernelGGL(k_qw_faces, dim3((D.v.nx + 1 + 63) / 64, (D.v.ny + 1 + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, ph); }
    else if (t.m.nbox > 0) hipLaunchKernelGGL(k_qw_faces_m, grid_m(t.m, 1, 1), dim3(64, 4), 0, st, t.m.dv, t.m.fp, ph);
    HIPCHK(hipGetLastError());
    return 0;
}
// ghosts of b of level l: PiecewiseLinearFillPatch on coarse-fine cells, exchange between the boxes, copies on domain sides
int hier_gap_ghosts(suhmo_hier *H, int l, hipStream_t st)
{
    int rc;
    LevT t;
    if ((rc = lev_target(H, l, st, t))) return rc;
    if ((rc = suhmo_hier_pwl_(H, l, SUHMO_F_B, SUHMO_F_B, st))) return rc;
    if ((rc = suhmo_hier_ff_(H, l, SUHMO_F_B, -1, true, st))) return rc;
    if (t.base) { if ((rc = suhmo_copy_ghosts(t.base, 0, SUHMO_F_B, st))) return rc; return exchange1(t.base, SUHMO_F_B, st); }   // rank strips: halo rows
    return suhmo_multi_coef_ghosts(t.m, SUHMO_F_B, st);
}
// lagged diffusion terms (diffusion_terms of one level) and RHS_h / the gap-height right-hand side of a whole level
int hier_melt(suhmo_hier *H, int l, const suhmo_model_params_t *mp, double dt, int final_, bool diffusion, hipStream_t st)
{
    int rc;
    LevT t;
    if ((rc = lev_target(H, l, st, t))) return rc;
    const suhmo_phys_t &ph = suhmo_hier_boxes_(H, l)[0]->ph;
    if (t.base) {
        suhmo_level *L = t.base;
        Depth &D = L->d[0];
        if (diffusion && (rc = diffusion_terms(L, mp, st))) return rc;
        if (final_) hipLaunchKernelGGL(k_melt<1>, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, ph, *mp, dt);
        else hipLaunchKernelGGL(k_melt<0>, dim3((D.v.nx + 63) / 64, (D.v.ny + 3) / 4), dim3(64, 4), 0, st, D.v, D.fp, ph, *mp, dt);
    } else {
        const suhmo_multi &m = t.m;
        if (diffusion) for (int f : {SUHMO_F_DCX, SUHMO_F_DCY, SUHMO_F_DTERM}) if ((rc = suhmo_hier_ensure_(H, l, f))) return rc;
        if (m.nbox <= 0) return 0;                                               // owner computes: none of this level's boxes is this rank's
        if (diffusion) {
            if ((rc = lev_target(H, l, st, t))) return rc;                       // the tables after the allocation
            int n = 2 * m.maxny + 2 * m.maxnx;
            hipLaunchKernelGGL(k_extrap_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp, (int)SUHMO_F_MR);
            hipLaunchKernelGGL(k_dcoef_faces_m, grid_m(m, 1, 1), dim3(64, 4), 0, st, m.dv, m.fp, ph, mp->rho_i);
            hipLaunchKernelGGL(k_difterm_m, grid_m(m), dim3(64, 4), 0, st, m.dv, m.fp);
        }
        if (final_) hipLaunchKernelGGL(k_melt_m<1>, grid_m(m), dim3(64, 4), 0, st, m.dv, m.fp, ph, *mp, dt);
        else hipLaunchKernelGGL(k_melt_m<0>, grid_m(m), dim3(64, 4), 0, st, m.dv, m.fp, ph, *mp, dt);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// max h and max |h_lagged - h| over the cells of level l no finer level covers
int hier_picard_maxima(suhmo_hier *H, int l, bool covered, double *maxh, double *maxd, hipStream_t st)
{
    int rc;
    LevT t;
    if ((rc = lev_target(H, l, st, t))) return rc;
    suhmo_level *slot = suhmo_hier_boxes_(H, 0)[0];
    if (t.base) {
        suhmo_level *L = t.base;
        if ((rc = picard_maxima(L, L->d[0].fp.f[SUHMO_F_PHI], L->d[0].fp.f[SUHMO_F_HLAG], maxh, maxd, st, Excl{0, 0, 0, 0}, covered ? L->d[0].fp.f[SUHMO_F_COVER] : nullptr))) return rc;
        if (L->ar && (L->d[0].v.rk[0] || L->d[0].v.rk[1])) {                     // computeMax over the ranks of level 0
            if ((rc = L->ar(L->user, maxh))) return rc;
            if ((rc = L->ar(L->user, maxd))) return rc;
        }
        return 0;
    }
    const suhmo_multi &m = t.m;
    *maxh = -1.0e300; *maxd = 0.0;
    if (m.nbox > 0) {
        dim3 grd(std::min((m.maxnx + 63) / 64, 4), std::min((m.maxny + 3) / 4, 8), m.nbox);       // 2 values per block: 64 nbox doubles
        hipLaunchKernelGGL(k_picard2_partial_m, grd, dim3(64, 4), 0, st, m.dv, m.fp, covered ? 1 : 0, m.red);
        hipLaunchKernelGGL(k_max2_final, dim3(1), dim3(256), 0, st, m.red, (int)(grd.x * grd.y * grd.z), slot->scratch, suhmo_host_slot(slot));
        HIPCHK(hipGetLastError());
        if ((rc = suhmo_readback(slot, st, maxh, maxd))) return rc;
    }
    if (suhmo_hier_partitioned_(H)) {                                            // owner computes: computeMax over the ranks
        if ((rc = suhmo_hier_allreduce_max_(H, maxh)) || (rc = suhmo_hier_allreduce_max_(H, maxd))) return rc;
    }
    return 0;
}
}  // namespace

extern "C" int suhmo_hier_timestep(suhmo_hier_t *H, const suhmo_model_params_t *mp, double dt, int cur_step,
                                   int *picard_iters, int *vcycles, suhmo_stream_t s)
{
    SUHMO_TIME("AmrHydro::timeStepFAS");
    ARG(H && mp); ARG(dt > 0 && cur_step >= 1);
    if (mp->use_impl_diff && mp->diffFactor == 0.0) { suhmo_set_error("use_ImplDiff with diffFactor = 0"); return -1; }
    const int nlev = suhmo_hier_nlev_(H);
    HIPCHK(hipSetDevice(suhmo_hier_device_(H)));
    suhmo_hier_invalidate_(H);
    hipStream_t st = (hipStream_t)s;
    int rc;
    static const int need[] = {SUHMO_F_MR, SUHMO_F_PW, SUHMO_F_QWX, SUHMO_F_QWY, SUHMO_F_HLAG, SUHMO_F_CD, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_RE};
    for (int l = 0; l < nlev; l++) {
        for (int f : need) if ((rc = suhmo_hier_ensure_(H, l, f))) return rc;
        if (mp->use_moulin_source) {
            int k0, nk;
            suhmo_hier_owned_(H, l, &k0, &nk);
            for (int k = k0; k < k0 + nk; k++)
                if (!suhmo_hier_boxes_(H, l)[k]->d[0].fp.f[SUHMO_F_MSRC]) { suhmo_set_error("use_moulin_source without a moulin source term (suhmo_hier_moulin_source)"); return -1; }
        }
    }
    suhmo_level *base = suhmo_hier_boxes_(H, 0)[0];
    if ((base->d[0].v.rk[0] || base->d[0].v.rk[1]) && !(base->ex && base->ar)) { suhmo_set_error("time step on rank strips needs the exchange hooks on level 0"); return -1; }
    // [I]
    for (int l = 0; l < nlev; l++) if ((rc = hier_gap_ghosts(H, l, st))) return rc;
    if ((rc = suhmo_build_mg_coefficients(base, false, st))) return rc;         // bCoef: re-averaged by every V-cycle (bcoeff_otf)
    suhmo_solver_params_t sp;
    sp.num_smooth = 4; sp.num_bottom = 16; sp.max_iter = 100; sp.iter_min = 2; sp.imin = 5;
    sp.eps = 1.0e-7; sp.hang = 0.01; sp.norm_thresh = 1.0e-7; sp.bcoeff_otf = 1; sp.max_depth = -1;
    if (cur_step < 50) { sp.num_bottom = 10; sp.eps = 1.0e-10; sp.hang = 0.0001; sp.imin = 20; }
    bool converged = false;
    int ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {
        for (int l = 0; l < nlev; l++) {
            if ((rc = hier_gap_ghosts(H, l, st))) return rc;
            if (l == 0 && (rc = exchange1(base, SUHMO_F_MR, st))) return rc;                    // rank strips: levelmR.exchange() :2513
            if ((rc = suhmo_hier_pwl_(H, l, SUHMO_F_MR, SUHMO_F_MR, st))) return rc;
            if ((rc = suhmo_hier_ff_(H, l, SUHMO_F_MR, -1, true, st))) return rc;               // levelmR.exchange() :2513
            if (l == 0) { Depth &D = base->d[0];
                HIPCHK(hipMemcpyAsync(D.fp.f[SUHMO_F_HLAG], D.fp.f[SUHMO_F_PHI], D.elems * sizeof(double), hipMemcpyDeviceToDevice, st)); }
            else { suhmo_multi m; if ((rc = suhmo_hier_multi_(H, l, st, &m)) || (rc = suhmo_multi_copy(m, SUHMO_F_HLAG, SUHMO_F_PHI, st))) return rc; }
        }
        for (int l = 0; l < nlev; l++) if ((rc = hier_chain(H, l, st))) return rc;
        for (int l = 0; l < nlev; l++) {                                                        // aCoeff_bCoeff :3087-3102
            LevT t;
            if ((rc = lev_target(H, l, st, t))) return rc;
            if (t.base) rc = suhmo_bcoef_faces(t.base, 0, st); else rc = suhmo_multi_bcoef_faces(t.m, suhmo_hier_boxes_(H, l)[0]->ph, st);
            if (rc) return rc;
        }
        for (int l = 0; l < nlev; l++) if ((rc = hier_melt(H, l, mp, dt, 0, mp->diffFactor != 0.0, st))) return rc;
        if ((rc = exchange1(base, SUHMO_F_RHS, st))) return rc;                                 // rank strips: halo rows relaxed redundantly
        int it = 0;
        if ((rc = suhmo_hier_solve(H, &sp, &it, nullptr, s))) return rc;
        nv += it;
        for (int l = nlev - 1; l > 0; l--) if ((rc = suhmo_hier_avg_(H, l, SUHMO_F_PHI, SUHMO_F_PHI, st))) return rc;   // CoarseAverage :3138-3141
        double maxHead = -1.0e300, maxd = 0.0, res = 0.0;
        for (int l = 0; l < nlev; l++) {
            double m = 0.0, d = 0.0;
            if ((rc = hier_picard_maxima(H, l, l < nlev - 1, &m, &d, st))) return rc;
            maxHead = std::max(maxHead, m); maxd = std::max(maxd, d);
        }
        res = picard_quotient(maxd, maxHead);
        if (ite_idx > 100) { suhmo_set_error("does not converge (Picard iterations > 100)"); return -6; }
        if (cur_step < 2) { if (res < 0.05 && cur_picard > 2) converged = true; }
        else if (cur_step < 50) { if (res < 0.05) converged = true; }
        else { if (res < mp->eps_picard) converged = true; }
        ite_idx++; cur_picard++;
    }
    // [III] level by level: the coarse gap height is already updated when the fine ghost cells are filled
    for (int l = 0; l < nlev; l++) {
        if ((rc = hier_chain(H, l, st))) return rc;
        if ((rc = hier_melt(H, l, mp, dt, 1, false, st))) return rc;
        if (mp->use_impl_diff) continue;                               // b stays, RES = b + dt RHS
        if ((rc = hier_gap_ghosts(H, l, st))) return rc;
    }
    if (mp->use_impl_diff) {                                           // SolveForGap_nl over the hierarchy :3425-3455
        suhmo_hier *G = nullptr;
        if ((rc = suhmo_hier_gap_(H, mp, dt, &G))) return rc;
        for (int l = 0; l < nlev; l++) {
            const auto &hb = suhmo_hier_boxes_(H, l), &gb = suhmo_hier_boxes_(G, l);
            if (l > 0) {                                               // all boxes of a level: one launch
                static const int fd[4] = {SUHMO_F_PHI, SUHMO_F_RHS, SUHMO_F_BX, SUHMO_F_BY}, fs[4] = {SUHMO_F_B, SUHMO_F_RES, SUHMO_F_DCX, SUHMO_F_DCY};
                for (int f : fs) if ((rc = suhmo_hier_ensure_(H, l, f))) return rc;
                for (int f : fd) if ((rc = suhmo_hier_ensure_(G, l, f))) return rc;
                suhmo_multi mh, mg;
                if ((rc = suhmo_hier_multi_(H, l, st, &mh)) || (rc = suhmo_hier_multi_(G, l, st, &mg))) return rc;
                if ((rc = suhmo_multi_copy_between(mg, mh, fd, fs, 4, st))) return rc;                                     // initial guess = b :3382-3385
                for (suhmo_level *L : gb) L->d[0].phi_fresh = 0;
                continue;
            }
            for (size_t k = 0; k < hb.size(); k++) {
                Depth &D = hb[k]->d[0], &GD = gb[k]->d[0];
                if (GD.elems != D.elems) { suhmo_set_error("internal: gap hierarchy geometry"); return -4; }
                const size_t bytes = D.elems * sizeof(double);
                HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_PHI], D.fp.f[SUHMO_F_B], bytes, hipMemcpyDeviceToDevice, st));      // initial guess = b :3382-3385
                HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_RHS], D.fp.f[SUHMO_F_RES], bytes, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_BX], D.fp.f[SUHMO_F_DCX], bytes, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(GD.fp.f[SUHMO_F_BY], D.fp.f[SUHMO_F_DCY], bytes, hipMemcpyDeviceToDevice, st));
                GD.phi_fresh = 0;
            }
        }
        {   static const int halo_fields[] = {SUHMO_F_RHS, SUHMO_F_ACOEF, SUHMO_F_BX, SUHMO_F_BY};          // rank strips
            if ((rc = suhmo_exchange_list(suhmo_hier_boxes_(G, 0)[0], 0, halo_fields, 4, st))) return rc; }
        if ((rc = suhmo_level_build_mg_coefficients(suhmo_hier_boxes_(G, 0)[0], s))) return rc;
        suhmo_solver_params_t spg;
        gap_solver_params(spg, cur_step);
        if ((rc = suhmo_hier_solve(G, &spg, nullptr, nullptr, s))) return rc;
        for (int l = 0; l < nlev; l++) {
            const auto &hb = suhmo_hier_boxes_(H, l), &gb = suhmo_hier_boxes_(G, l);
            if (l > 0) {
                static const int fd[1] = {SUHMO_F_B}, fs[1] = {SUHMO_F_PHI};
                suhmo_multi mh, mg;
                if ((rc = suhmo_hier_multi_(H, l, st, &mh)) || (rc = suhmo_hier_multi_(G, l, st, &mg))) return rc;
                if (mp->freeze_icefree_gap && mh.nbox > 0) {
                    hipLaunchKernelGGL(k_keep_icefree_m, grid_m(mh), dim3(64, 4), 0, st, mh.dv, mh.fp, mg.fp);
                    HIPCHK(hipGetLastError());
                }
                if ((rc = suhmo_multi_copy_between(mh, mg, fd, fs, 1, st))) return rc;
            } else
                for (size_t k = 0; k < hb.size(); k++) {
                    if (mp->freeze_icefree_gap && (rc = keep_icefree(hb[k], gb[k], st))) return rc;
                    HIPCHK(hipMemcpyAsync(hb[k]->d[0].fp.f[SUHMO_F_B], gb[k]->d[0].fp.f[SUHMO_F_PHI], hb[k]->d[0].elems * sizeof(double), hipMemcpyDeviceToDevice, st));
                }
            if ((rc = hier_gap_ghosts(H, l, st))) return rc;
        }
    }
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles) *vcycles = nv;
    return 0;
}

// ------------------------------------------------------------------ moulin source term
// Calc_moulin_integral / Calc_moulin_source_term_distributed (src/AmrHydro.cpp:1866-2066).  The n x N array of the
// reference (one component per moulin) is never stored: pass 1 integrates every Gaussian (per-tile partial sums in a
// fixed order, then one block per moulin), pass 2 re-evaluates and normalises.  A Gaussian whose argument exceeds
// 760 underflows to exactly 0 in the reference too, so tiles / cells that far away are skipped without changing a bit.
namespace {
__device__ __forceinline__ double moulin_cell(double xc, double yc, double dx, double dy, double mx, double my, double sg, bool &zero)
{
    const double l[3] = {-0.77459666924 / 2.0, 0.0, 0.77459666924 / 2.0};
    const double v[3] = {0.5555555555, 0.8888888888, 0.5555555555};
    const double k = -1.0 / (2.0 * sg * sg), prefac = 1.0 / (sg * sqrt(2.0 * 3.14));
    double ex[3], ey[3];
    for (int q = 0; q < 3; q++) { ex[q] = (xc + l[q]) * dx - mx; ey[q] = (yc + l[q]) * dy - my; }
    double ax = fmin(fabs(ex[0]), fabs(ex[2])), ay = fmin(fabs(ey[0]), fabs(ey[2]));
    if (ex[0] * ex[2] < 0.0) ax = 0.0;
    if (ey[0] * ey[2] < 0.0) ay = 0.0;
    zero = -k * (ax * ax + ay * ay) > 760.0;
    if (zero) return 0.0;
    double MS[9];
    for (int b = 0; b < 3; b++)
        for (int a = 0; a < 3; a++) { double rad = ex[a] * ex[a] + ey[b] * ey[b]; MS[3 * b + a] = prefac * exp(k * rad); }
    return v[0] * v[0] * MS[0] + v[1] * v[0] * MS[1] + v[2] * v[0] * MS[2]
         + v[0] * v[1] * MS[3] + v[1] * v[1] * MS[4] + v[2] * v[1] * MS[5]
         + v[0] * v[2] * MS[6] + v[1] * v[2] * MS[7] + v[2] * v[2] * MS[8];
}
__global__ __launch_bounds__(256) void k_moulin_partial(DV v, int n, const double *__restrict__ mo, double *__restrict__ partial, Excl ex,
                                                        const double *__restrict__ cover = nullptr)
{
    __shared__ double sm[256];
    const int tid = threadIdx.y * 16 + threadIdx.x;
    const int i = blockIdx.x * 16 + threadIdx.x, j = blockIdx.y * 16 + threadIdx.y;
    bool in = i < v.nx && j < v.ny && !(i >= ex.i0 && i < ex.i1 && j >= ex.j0 && j < ex.j1);   // covered by a finer level: 0
    if (in && cover && cover[cidx(v, i, j)] != 0.0) in = false;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const double tx0 = (v.i0 + blockIdx.x * 16) * v.dx, tx1 = (v.i0 + blockIdx.x * 16 + 16) * v.dx;      // the tile in physical coordinates (a patch / box
    const double ty0 = (v.j0 + blockIdx.y * 16) * v.dy, ty1 = (v.j0 + blockIdx.y * 16 + 16) * v.dy;      // starts at (i0, j0) of its level)
    for (int m = 0; m < n; m++) {
        const double mx = mo[3 * m], my = mo[3 * m + 1], sg = mo[3 * m + 2];
        double ddx = mx < tx0 ? tx0 - mx : (mx > tx1 ? mx - tx1 : 0.0), ddy = my < ty0 ? ty0 - my : (my > ty1 ? my - ty1 : 0.0);
        if ((ddx * ddx + ddy * ddy) / (2.0 * sg * sg) > 760.0) { if (tid == 0) partial[(size_t)blk * n + m] = 0.0; continue; }   // uniform
        bool z;
        double val = in ? moulin_cell(i + 0.5 + v.i0, j + 0.5 + v.j0, v.dx, v.dy, mx, my, sg, z) * v.dx * v.dy : 0.0;
        sm[tid] = val;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = sm[tid] + sm[tid + s]; __syncthreads(); }
        if (tid == 0) partial[(size_t)blk * n + m] = sm[0];
        __syncthreads();
    }
}
__global__ void k_moulin_final(const double *__restrict__ partial, int nblk, int n, double *__restrict__ integ)
{
    __shared__ double sm[256];
    const int m = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int b = tid; b < nblk; b += 256) acc = acc + partial[(size_t)b * n + m];
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = sm[tid] + sm[tid + s]; __syncthreads(); }
    if (tid == 0) integ[m] = sm[0];
}
__global__ __launch_bounds__(256) void k_moulin_src(DV v, int n, const double *__restrict__ mo, const double *__restrict__ flux,
                                                    const double *__restrict__ integ, double tf, double *__restrict__ out, Excl ex,
                                                    const double *__restrict__ cover = nullptr)
{
    const int i = blockIdx.x * 16 + threadIdx.x, j = blockIdx.y * 16 + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    if ((i >= ex.i0 && i < ex.i1 && j >= ex.j0 && j < ex.j1) || (cover && cover[cidx(v, i, j)] != 0.0)) { out[cidx(v, i, j)] = 0.0; return; }   // filled by the average of the finer level
    double sum = 0.0;
    for (int m = 0; m < n; m++) {
        bool z;
        double val = moulin_cell(i + 0.5 + v.i0, j + 0.5 + v.j0, v.dx, v.dy, mo[3 * m], mo[3 * m + 1], mo[3 * m + 2], z);
        if (!z) sum += val * tf / integ[m] * flux[m];
    }
    out[cidx(v, i, j)] = sum;
}
}  // namespace

extern "C" int suhmo_level_moulin_source(suhmo_level_t *L, int n, const double *positions, const double *sigma,
                                         const double *flux, double time_factor, double *integrals, suhmo_stream_t s)
{
    SUHMO_TIME("AmrHydro::Calc_moulin_source_term_distributed");
    ARG(L && n >= 1 && positions && sigma && flux);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[0];
    if (L->desc.nx_global > 0) { suhmo_set_error("moulin source on an AMR patch is not built yet (the integral spans all levels)"); return -5; }
    double *out = suhmo_field(L, 0, SUHMO_F_MSRC);
    if (!out) { suhmo_set_error("field allocation failed"); return -2; }
    std::vector<double> h(4 * (size_t)n);
    for (int m = 0; m < n; m++) {
        ARG(sigma[m] > 0.0);
        h[3 * m] = positions[2 * m]; h[3 * m + 1] = positions[2 * m + 1]; h[3 * m + 2] = sigma[m]; h[3 * (size_t)n + m] = flux[m];
    }
    // rank strip: the integrals run over the whole level on every rank (geometry only), in the single-level order
    DV vg = D.v;
    vg.ny = D.v.nyg; vg.j0 = 0;
    dim3 blk(16, 16), grd((D.v.nx + 15) / 16, (D.v.ny + 15) / 16), grdg((vg.nx + 15) / 16, (vg.ny + 15) / 16);
    const size_t nblk = (size_t)grdg.x * grdg.y;
    double *dev = nullptr;
    HIPCHK(hipMalloc(&dev, (5 * (size_t)n + nblk * n) * sizeof(double)));
    double *mo = dev, *fl = dev + 3 * (size_t)n, *integ = dev + 4 * (size_t)n, *partial = dev + 5 * (size_t)n;
    hipError_t e = hipMemcpyAsync(dev, h.data(), 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_moulin_partial, grdg, blk, 0, st, vg, n, mo, partial, Excl{0, 0, 0, 0});
        hipLaunchKernelGGL(k_moulin_final, dim3(n), dim3(256), 0, st, partial, (int)nblk, n, integ);
        hipLaunchKernelGGL(k_moulin_src, grd, blk, 0, st, D.v, n, mo, fl, integ, time_factor, out, Excl{0, 0, 0, 0});
        e = hipGetLastError();
    }
    if (e == hipSuccess && integrals) e = hipMemcpyAsync(integrals, integ, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) { suhmo_set_error("moulin source: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

// Calc_moulin_integral + Calc_moulin_source_term_distributed on the hierarchy (:1866-2066, :2797-2837): every level samples
// the Gaussians at its own resolution, cells under a finer level do not count in the integrals (finest level first, :1891)
// and receive the average of the finer level's source term afterwards (CoarseAverage :2819-2826).
extern "C" int suhmo_amr_moulin_source(suhmo_level_t **lv, int nlev, const int *patch_boxes, int n, const double *positions,
                                       const double *sigma, const double *flux, double time_factor, double *integrals, suhmo_stream_t s)
{
    ARG(lv && nlev >= 1 && nlev <= 8 && lv[0] && n >= 1 && positions && sigma && flux);
    int rc = suhmo_amr_check_hierarchy(lv, nlev); if (rc) return rc;
    // geometry of every level's WHOLE rectangle: from the boxes (rank strips: a rank may hold a part of a level or none of
    // it, and integrates all of them itself -- analytic integrand, single-process order, no communication) or the handles
    struct Geo { int nx, ny, i0, j0; double dx, dy; } geo[8];
    const DV &b = lv[0]->d[0].v;
    geo[0] = Geo{b.nx, b.nyg, 0, 0, b.dx, b.dy};
    for (int l = 1; l < nlev; l++) {
        if (patch_boxes) {
            const int *q = patch_boxes + 4 * (l - 1);
            ARG(q[2] >= q[0] && q[3] >= q[1]);
            geo[l] = Geo{2 * (q[2] - q[0] + 1), 2 * (q[3] - q[1] + 1), 2 * q[0], 2 * q[1], geo[l - 1].dx / 2.0, geo[l - 1].dy / 2.0};
            if (lv[l]) { const DV &v = lv[l]->d[0].v; const bool part = v.rk[0] || v.rk[1];            // a rank strip holds some of the rows, a whole patch all of them
                if (v.nx != geo[l].nx || v.i0 != geo[l].i0 || v.j0 < geo[l].j0 || v.j0 + v.ny > geo[l].j0 + geo[l].ny
                    || (!part && (v.j0 != geo[l].j0 || v.ny != geo[l].ny))) { suhmo_set_error("moulin source: level %d does not match patch_boxes", l); return -1; } }
        } else {
            if (!lv[l] || lv[l]->d[0].v.rk[0] || lv[l]->d[0].v.rk[1]) { suhmo_set_error("moulin source on rank strips needs patch_boxes"); return -1; }
            const DV &v = lv[l]->d[0].v;
            geo[l] = Geo{v.nx, v.ny, v.i0, v.j0, v.dx, v.dy};
        }
    }
    auto excl_of = [&](int l, int i0, int j0) {                       // the box of level l+1 in cells of level l, relative to (i0, j0)
        if (l >= nlev - 1) return Excl{0, 0, 0, 0};
        return Excl{geo[l + 1].i0 / 2 - i0, geo[l + 1].j0 / 2 - j0, (geo[l + 1].i0 + geo[l + 1].nx) / 2 - i0, (geo[l + 1].j0 + geo[l + 1].ny) / 2 - j0};
    };
    for (int l = 0; l < nlev; l++) if (lv[l] && !suhmo_field(lv[l], 0, SUHMO_F_MSRC)) { suhmo_set_error("field allocation failed"); return -2; }
    HIPCHK(hipSetDevice(lv[0]->device));
    hipStream_t st = (hipStream_t)s;
    std::vector<double> h(4 * (size_t)n), total((size_t)n, 0.0), part((size_t)n);
    for (int m = 0; m < n; m++) {
        ARG(sigma[m] > 0.0);
        h[3 * m] = positions[2 * m]; h[3 * m + 1] = positions[2 * m + 1]; h[3 * m + 2] = sigma[m]; h[3 * (size_t)n + m] = flux[m];
    }
    size_t maxblk = 0;
    for (int l = 0; l < nlev; l++) maxblk = std::max(maxblk, (size_t)((geo[l].nx + 15) / 16) * ((geo[l].ny + 15) / 16));
    double *dev = nullptr;
    HIPCHK(hipMalloc(&dev, (5 * (size_t)n + maxblk * n) * sizeof(double)));
    double *mo = dev, *fl = dev + 3 * (size_t)n, *integ = dev + 4 * (size_t)n, *partial = dev + 5 * (size_t)n;
    hipError_t e = hipMemcpyAsync(dev, h.data(), 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    for (int l = nlev - 1; l >= 0 && e == hipSuccess; l--) {          // finest first (:1891)
        DV vg = b;                                                     // only the geometry below is read by the kernel
        vg.nx = geo[l].nx; vg.ny = geo[l].ny; vg.i0 = geo[l].i0; vg.j0 = geo[l].j0; vg.dx = geo[l].dx; vg.dy = geo[l].dy;
        dim3 blk(16, 16), grd((vg.nx + 15) / 16, (vg.ny + 15) / 16);
        hipLaunchKernelGGL(k_moulin_partial, grd, blk, 0, st, vg, n, mo, partial, excl_of(l, vg.i0, vg.j0));
        hipLaunchKernelGGL(k_moulin_final, dim3(n), dim3(256), 0, st, partial, (int)(grd.x * grd.y), n, integ);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(part.data(), integ, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        for (int m = 0; m < n; m++) total[m] += part[m];
    }
    if (e == hipSuccess) e = hipMemcpyAsync(integ, total.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    for (int l = 0; l < nlev && e == hipSuccess; l++) {
        if (!lv[l]) continue;
        const DV &v = lv[l]->d[0].v;
        dim3 blk(16, 16), grd((v.nx + 15) / 16, (v.ny + 15) / 16);
        hipLaunchKernelGGL(k_moulin_src, grd, blk, 0, st, v, n, mo, fl, integ, time_factor, lv[l]->d[0].fp.f[SUHMO_F_MSRC], excl_of(l, v.i0, v.j0));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) { suhmo_set_error("moulin source: %s", hipGetErrorString(e)); return -2; }
    for (int l = nlev - 1; l > 0; l--) if (lv[l] && (rc = suhmo_amr2_average(lv[l - 1], lv[l], SUHMO_F_MSRC, SUHMO_F_MSRC, s))) return rc;
    if (integrals) for (int m = 0; m < n; m++) integrals[m] = total[m];
    return 0;
}


// suhmo_amr_moulin_source on a hierarchy of box unions (oracle/amr_step_m.c:or_amrm_model_moulin_source): finest level first,
// box after box; cells under a finer level (SUHMO_F_COVER) do not count and get the finer level's average afterwards
extern "C" int suhmo_hier_moulin_source(suhmo_hier_t *H, int n, const double *positions, const double *sigma, const double *flux,
                                        double time_factor, double *integrals, suhmo_stream_t s)
{
    ARG(H && n >= 1 && positions && sigma && flux);
    const int nlev = suhmo_hier_nlev_(H);
    HIPCHK(hipSetDevice(suhmo_hier_device_(H)));
    hipStream_t st = (hipStream_t)s;
    int rc;
    std::vector<double> h(4 * (size_t)n), total((size_t)n, 0.0), part((size_t)n);
    for (int m = 0; m < n; m++) {
        ARG(sigma[m] > 0.0);
        h[3 * m] = positions[2 * m]; h[3 * m + 1] = positions[2 * m + 1]; h[3 * m + 2] = sigma[m]; h[3 * (size_t)n + m] = flux[m];
    }
    // owner computes (levels >= 1 dealt to the ranks): a rank integrates and fills the boxes it owns; the per-box integrals of all ranks are
    // gathered and added up in the single-process order (finest level first, box after box), so every rank gets the same bits
    const bool parted = suhmo_hier_partitioned_(H);
    size_t maxblk = 0, nbt = 0;
    std::vector<size_t> first(nlev + 1, 0);
    for (int l = 0; l < nlev; l++) {
        const auto &bx = suhmo_hier_boxes_(H, l);
        first[l] = nbt; nbt += bx.size();
        int k0, nk;
        suhmo_hier_owned_(H, l, &k0, &nk);
        for (int k = 0; k < (int)bx.size(); k++) {
            suhmo_level *L = bx[k];
            if (k >= k0 && k < k0 + nk && !suhmo_field(L, 0, SUHMO_F_MSRC)) { suhmo_set_error("field allocation failed"); return -2; }
            maxblk = std::max(maxblk, (size_t)((L->d[0].v.nx + 15) / 16) * (((l == 0 ? L->d[0].v.nyg : L->d[0].v.ny) + 15) / 16));
        }
    }
    first[nlev] = nbt;
    double *dev = nullptr;
    HIPCHK(hipMalloc(&dev, (5 * (size_t)n + maxblk * n) * sizeof(double)));
    double *mo = dev, *fl = dev + 3 * (size_t)n, *integ = dev + 4 * (size_t)n, *partial = dev + 5 * (size_t)n;
    hipError_t e = hipMemcpyAsync(dev, h.data(), 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    DV whole;
    const double *whole_cover = suhmo_hier_base_cover_(H, &whole);     // level 0 cut into rank strips: every rank integrates all of it (geometry only)
    std::vector<double> perbox(nbt * (size_t)n, 0.0);                  // the integrals over every box (this rank's; the others' after the gather)
    for (int l = nlev - 1; l >= 0 && e == hipSuccess; l--) {
        const auto &bx = suhmo_hier_boxes_(H, l);
        int k0, nk;
        suhmo_hier_owned_(H, l, &k0, &nk);
        for (int k = k0; k < k0 + nk; k++) {
            suhmo_level *L = bx[k];
            const bool cutbase = l == 0 && (L->d[0].v.rk[0] || L->d[0].v.rk[1]);
            DV v = L->d[0].v;
            if (cutbase) { v.ny = v.nyg; v.j0 = 0; }
            dim3 blk(16, 16), grd((v.nx + 15) / 16, (v.ny + 15) / 16);
            const double *cover = l < nlev - 1 ? (cutbase ? whole_cover : L->d[0].fp.f[SUHMO_F_COVER]) : nullptr;
            hipLaunchKernelGGL(k_moulin_partial, grd, blk, 0, st, v, n, mo, partial, Excl{0, 0, 0, 0}, cover);
            hipLaunchKernelGGL(k_moulin_final, dim3(n), dim3(256), 0, st, partial, (int)(grd.x * grd.y), n, integ);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(part.data(), integ, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
            for (int m = 0; m < n; m++) perbox[(first[l] + k) * (size_t)n + m] = part[m];
        }
    }
    if (e == hipSuccess && parted) {                                   // every rank's integrals over its boxes -> every rank
        const int world = suhmo_hier_world_(H);
        const size_t cnt = nbt * (size_t)n;
        double *gs = nullptr, *gr = nullptr;
        std::vector<double> all(cnt * world);
        e = hipMalloc(&gs, cnt * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&gr, cnt * world * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(gs, perbox.data(), cnt * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && (rc = suhmo_hier_allgather_(H, gs, (long)cnt, gr, st))) { (void)hipFree(gs); (void)hipFree(gr); (void)hipFree(dev); return rc; }
        if (e == hipSuccess) e = hipMemcpyAsync(all.data(), gr, cnt * world * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (gs) (void)hipFree(gs);
        if (gr) (void)hipFree(gr);
        if (e == hipSuccess)
            for (int l = 1; l < nlev; l++)
                for (size_t k = 0; k < first[l + 1] - first[l]; k++) {
                    const int o = suhmo_hier_box_owner(H, l, (int)k, nullptr);
                    if (o >= 0) for (int m = 0; m < n; m++) perbox[(first[l] + k) * (size_t)n + m] = all[(size_t)o * cnt + (first[l] + k) * (size_t)n + m];
                }
    }
    for (int l = nlev - 1; l >= 0; l--)                                // finest first (:1891), box after box
        for (size_t k = first[l]; k < first[l + 1]; k++)
            for (int m = 0; m < n; m++) total[m] += perbox[k * (size_t)n + m];
    if (e == hipSuccess) e = hipMemcpyAsync(integ, total.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    for (int l = 0; l < nlev && e == hipSuccess; l++) {
        const auto &bx = suhmo_hier_boxes_(H, l);
        int k0, nk;
        suhmo_hier_owned_(H, l, &k0, &nk);
        for (int k = k0; k < k0 + nk; k++) {
            suhmo_level *L = bx[k];
            const DV &v = L->d[0].v;
            dim3 blk(16, 16), grd((v.nx + 15) / 16, (v.ny + 15) / 16);
            const double *cover = l < nlev - 1 ? L->d[0].fp.f[SUHMO_F_COVER] : nullptr;
            hipLaunchKernelGGL(k_moulin_src, grd, blk, 0, st, v, n, mo, fl, integ, time_factor, L->d[0].fp.f[SUHMO_F_MSRC], Excl{0, 0, 0, 0}, cover);
            e = hipGetLastError();
            if (e != hipSuccess) break;
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) { suhmo_set_error("moulin source: %s", hipGetErrorString(e)); return -2; }
    for (int l = nlev - 1; l > 0; l--) if ((rc = suhmo_hier_avg_(H, l, SUHMO_F_MSRC, SUHMO_F_MSRC, st))) return rc;
    if (integrals) for (int m = 0; m < n; m++) integrals[m] = total[m];
    return 0;
}

// COMPUTE_TIMEVARYINGRECHARGE (src/AmrHydroF.ChF:346-373) on the ghosted box of the source term
__global__ void k_time_varying_recharge(DV v, const double *__restrict__ zs, double *__restrict__ out, double TK, double background)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x - 1, j = blockIdx.y * blockDim.y + threadIdx.y - 1;
    if (i > v.nx || j > v.ny) return;
    const double ddf = 0.01 / 86400., dT_dZ = -0.0075;
    int idx = cidx(v, i, j);
    out[idx] = fmax(ddf * (TK + zs[idx] * dT_dZ), 0.0) + background;
}
extern "C" int suhmo_level_time_varying_recharge(suhmo_level_t *L, double T_K, double background_input, suhmo_stream_t s)
{
    ARG(L);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[0];
    if (!D.fp.f[SUHMO_F_ZS]) { suhmo_set_error("time-varying recharge: load the ice surface height (SUHMO_F_ZS) first"); return -1; }
    double *out = suhmo_field(L, 0, SUHMO_F_MSRC);
    if (!out) { suhmo_set_error("field allocation failed"); return -2; }
    hipLaunchKernelGGL(k_time_varying_recharge, dim3((D.v.nx + 2 + 63) / 64, (D.v.ny + 2 + 3) / 4), dim3(64, 4), 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_ZS], out, T_K, background_input);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ SHMIP cross-section table
// one thread per cell column, rows summed in ascending j (the order of the reference's BoxIterator per column)
__global__ void k_postproc_columns(DV v, FP fp, suhmo_model_params_t mp, double *__restrict__ out /* 8 x nx */)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.nx) return;
    const double *__restrict__ qx = fp.f[SUHMO_F_QWX], *__restrict__ cd = fp.f[SUHMO_F_CD], *__restrict__ mR = fp.f[SUHMO_F_MR];
    const double *__restrict__ Pw = fp.f[SUHMO_F_PW], *__restrict__ Pi = fp.f[SUHMO_F_PI], *__restrict__ mk = fp.f[SUHMO_F_MASK];
    const double *__restrict__ ms = mp.use_moulin_source ? fp.f[SUHMO_F_MSRC] : nullptr;
    double qt = 0.0, qc = 0.0, qd = 0.0, ext = 0.0, mr = 0.0, yl = 0.0, avp = 0.0, cnt = 0.0;
    for (int j = 0; j < v.ny; j++) {
        int idx = cidx(v, i, j);
        double cdec = 0.5 * (cd[idx] + cd[idx - 1]);                     // CellToEdge(chanDegree), :3696-3697
        double q = qx[idx] * v.dy;
        qt += q; qc += q * cdec; qd += q * (1.0 - cdec);                 // :3734-3738
        bool ice = mk[idx] > 0.0;
        double src = ms ? ms[idx] * mp.ramp + mp.distributed_input : (ice ? mp.distributed_input : 0.0);
        if (ice) { ext += src * v.dy * v.dx; mr += (mR[idx] / mp.rho_w) * v.dy * v.dx; yl += v.dy; }    // :3766-3775
        if (ice && Pi[idx] > 0.0) { avp += Pi[idx] - Pw[idx]; cnt += 1.0; }                             // :3778-3783
    }
    out[0 * v.nx + i] = yl; out[1 * v.nx + i] = qt; out[2 * v.nx + i] = qc; out[3 * v.nx + i] = qd;
    out[4 * v.nx + i] = ext; out[5 * v.nx + i] = mr; out[6 * v.nx + i] = avp; out[7 * v.nx + i] = cnt;
}
// column sums over the rows of this level / strip: 8 x nx = width, Q, Q channelised, Q distributed, external recharge,
// melt recharge, sum of (Pi - Pw), count of its terms
extern "C" int suhmo_level_postproc_partial(suhmo_level_t *L, const suhmo_model_params_t *mp, double *sums, suhmo_stream_t s)
{
    ARG(L && mp && sums);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[0];
    if (L->desc.nx_global > 0) { suhmo_set_error("post-processing table on an AMR patch is not built"); return -5; }
    for (int f : {SUHMO_F_QWX, SUHMO_F_CD, SUHMO_F_MR, SUHMO_F_PW}) if (!D.fp.f[f]) { suhmo_set_error("no time step has run on this level"); return -1; }
    if (mp->use_moulin_source && !D.fp.f[SUHMO_F_MSRC]) { suhmo_set_error("use_moulin_source without suhmo_level_moulin_source"); return -1; }
    const int nx = D.v.nx;
    double *dev = nullptr;
    HIPCHK(hipMalloc(&dev, 8 * (size_t)nx * sizeof(double)));
    hipLaunchKernelGGL(k_postproc_columns, dim3((nx + 63) / 64), dim3(64), 0, st, D.v, D.fp, *mp, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(sums, dev, 8 * (size_t)nx * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) { suhmo_set_error("postproc table: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
// the table from column sums (of the whole level: on rank strips the host adds the strips' sums first)
extern "C" int suhmo_postproc_finish(const double *sums, int nx, double dx, double *table)
{
    ARG(sums && table && nx > 0);
    const double *h = sums;
    double cext = 0.0, cmr = 0.0;
    for (int i = nx - 1; i >= 0; i--) {                    // recharge upstream of the column: cumulative from the upper end
        cext += h[4 * (size_t)nx + i]; cmr += h[5 * (size_t)nx + i];
        double *row = table + 8 * (size_t)i;
        row[0] = (i + 0.5) * dx / 1.0e3; row[1] = h[0 * (size_t)nx + i];
        row[2] = -h[1 * (size_t)nx + i]; row[3] = -h[2 * (size_t)nx + i]; row[4] = -h[3 * (size_t)nx + i];
        row[5] = cext; row[6] = cmr; row[7] = h[6 * (size_t)nx + i] / fmax(h[7 * (size_t)nx + i], 1.0) / 1.0e6;
    }
    return 0;
}
// the "Time(h - d)" lines of the temporal post-processing (src/AmrHydro.cpp:3778-3810, 4040-4053) from the column sums
extern "C" int suhmo_postproc_temporal(const double *sums, int nx, double dx, double *out)
{
    ARG(sums && out && nx > 1);
    const double *h = sums;
    const double lo[3] = {600.0, 3000.0, 5100.0}, hi[3] = {900.0, 3300.0, 5400.0};
    double tot = 0.0, cnt = 0.0, bs[3] = {0.0, 0.0, 0.0}, bc[3] = {0.0, 0.0, 0.0}, rech = 0.0;
    for (int i = 0; i < nx; i++) {
        const double x = (i + 0.5) * dx;
        tot += h[6 * (size_t)nx + i]; cnt += h[7 * (size_t)nx + i];
        for (int b = 0; b < 3; b++) if (x > lo[b] && x < hi[b]) { bs[b] += h[6 * (size_t)nx + i]; bc[b] += h[7 * (size_t)nx + i]; }
        if (i >= 1) rech += h[4 * (size_t)nx + i] + h[5 * (size_t)nx + i];
    }
    out[0] = tot / cnt;
    for (int b = 0; b < 3; b++) out[1 + b] = bs[b] / bc[b];
    out[4] = rech;
    out[5] = -h[1 * (size_t)nx + 1];
    return 0;
}
extern "C" int suhmo_level_postproc_temporal(suhmo_level_t *L, const suhmo_model_params_t *mp, double *out, suhmo_stream_t s)
{
    ARG(L && mp && out);
    Depth &D = L->d[0];
    if (D.v.ext[0] || D.v.ext[1]) { suhmo_set_error("rank strip: add the strips' suhmo_level_postproc_partial sums, then suhmo_postproc_temporal"); return -5; }
    std::vector<double> h(8 * (size_t)D.v.nx);
    int rc = suhmo_level_postproc_partial(L, mp, h.data(), s); if (rc) return rc;
    return suhmo_postproc_temporal(h.data(), D.v.nx, D.v.dx, out);
}
extern "C" int suhmo_level_postproc_table(suhmo_level_t *L, const suhmo_model_params_t *mp, double *table, suhmo_stream_t s)
{
    ARG(L && mp && table);
    Depth &D = L->d[0];
    if (D.v.ext[0] || D.v.ext[1]) { suhmo_set_error("rank strip: add the strips' suhmo_level_postproc_partial sums, then suhmo_postproc_finish"); return -5; }
    std::vector<double> h(8 * (size_t)D.v.nx);
    int rc = suhmo_level_postproc_partial(L, mp, h.data(), s); if (rc) return rc;
    return suhmo_postproc_finish(h.data(), D.v.nx, D.v.dx, table);
}
