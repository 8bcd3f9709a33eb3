/*
 * amr_step.c -- TEST INFRASTRUCTURE ONLY.  One AmrHydro::timeStepFAS (src/AmrHydro.cpp:2254-3460) on a hierarchy of
 * nested levels (level 0 = the domain, level l >= 1 = ONE rectangular patch refined by 2, as oracle/amrn.c): every
 * level runs the phases of oracle/time_loop.c on its own rectangle, with the inter-level steps of the reference in
 * between:
 *   PiecewiseLinearFillPatch of the coarse-fine ghost cells of b, mR (:2373-2380, :2499-2507), Re (:2711-2719,
 *     :3265-3273) [the fills of h and qw are never read: h's ghosts are re-interpolated by compGradientMAC before any
 *     use, qw is a debug variable]
 *   QuadCFInterp of h inside Gradient::compGradientMAC and of the cell-centred gradient (:1650-1656)
 *   SolveForHead_nl over all levels (AMRFASMultiGrid, oracle/amrn.c), CoarseAverage of h (:3138-3141)
 *   computeMax over the cells not covered by a finer level (:3169, :3185)
 * explicit gap-height update or the implicit one (SolveForGap_nl over the hierarchy, with the stand-in operator and cycle of
 * time_loop.c:solve_gap_implicit).  [Chombo] PiecewiseLinearFillPatch is restated
 * from upstream Chombo 3.2's documented algorithm (fillConstantInterp + computeMultiDimSlopes: central differences,
 * one-sided next to the domain boundary, FORT_INTERPLIMIT's multi-dimensional limiter over the 3 x 3 neighbourhood +
 * FORT_INTERPLINEAR) -- the fork is not vendored: UNPINNED, like the rest of the Chombo-side AMR pieces.
 */
#include "time_loop.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* oracle/amrn.c (no header of its own) */
typedef struct OrAmr OrAmr;
OrAmr *or_amr_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                     double alpha, double beta, int nlev, const int *patches);
void or_amr_destroy(OrAmr *A);
void or_amr_patch_io(OrAmr *A, int l, int field, double *g, int ghosted, int set);
int or_amr_solve(OrAmr *A, const OrSolverParams *sp, double *hist);

#define AMAXLEV 8
typedef struct OrAmrModel {
    int nlev;
    OrLevel *base;
    OrAmr *A;
    OrModel *M[AMAXLEV];
    int cur_step;
    /* implicit gap-height solve on the hierarchy (SolveForGap_nl :593-662): a second set of levels with alpha = 1,
     * beta = dt diffFactor, bCoef = D, no nonlinear term, Neumann-0 sides (as time_loop.c:solve_gap_implicit) */
    OrLevel *Gbase; OrAmr *GA; double G_dt;
    int nx0, ny0, max_box, nthreads, patches[4 * AMAXLEV];
    double dx0, dy0;
    OrBC bc; OrPhys ph;
} OrAmrModel;

#define G(M, a, i, j) (a)[(size_t)((j) + 1) * ((M)->nx + 2) + ((i) + 1)]      /* ghosted array of a level, LOCAL indices */

OrAmrModel *or_amr_model_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                                const OrModelParams *mp, int nlev, const int *patches)
{
    OrAmrModel *S = (OrAmrModel *)calloc(1, sizeof(OrAmrModel));
    S->nlev = nlev; S->base = base;
    S->nx0 = nx0; S->ny0 = ny0; S->dx0 = dx0; S->dy0 = dy0; S->bc = *bc; S->ph = *ph; S->max_box = 64; S->nthreads = 1;
    for (int k = 0; k < 4 * (nlev - 1); k++) S->patches[k] = patches[k];
    S->A = or_amr_create(base, nx0, ny0, dx0, dy0, bc, ph, 0.0, -1.0, nlev, patches);
    S->M[0] = or_model_create(base, nx0, ny0, dx0, dy0, bc, ph, mp);
    int nxg = nx0, nyg = ny0;                              /* domain size at the level being built */
    double dx = dx0, dy = dy0;
    for (int l = 1; l < nlev; l++) {
        const int *q = patches + 4 * (l - 1);                /* box of the patch in level l-1 cells */
        nxg *= 2; nyg *= 2; dx /= 2.0; dy /= 2.0;
        int i0 = 2 * q[0], j0 = 2 * q[1], nx = 2 * (q[2] - q[0] + 1), ny = 2 * (q[3] - q[1] + 1);
        S->M[l] = or_model_create(NULL, nx, ny, dx, dy, bc, ph, mp);
        or_model_set_patch(S->M[l], i0, j0, nxg, nyg);
    }
    return S;
}
void or_amr_model_destroy(OrAmrModel *S)
{
    if (!S) return;
    for (int l = 0; l < S->nlev; l++) or_model_destroy(S->M[l]);
    or_amr_destroy(S->A);
    if (S->GA) or_amr_destroy(S->GA);
    if (S->Gbase) or_level_destroy(S->Gbase);
    free(S);
}
OrModel *or_amr_model_level(OrAmrModel *S, int l) { return S->M[l]; }
void or_amr_model_gap_solver_layout(OrAmrModel *S, int max_box, int nthreads) { S->max_box = max_box; S->nthreads = nthreads; }
double *or_amr_model_field(OrAmrModel *S, int l, int id) { return or_model_field(S->M[l], id); }
void or_amr_model_dims(OrAmrModel *S, int l, int *nx, int *ny, int *i0, int *j0)
{ *nx = S->M[l]->nx; *ny = S->M[l]->ny; *i0 = S->M[l]->i0; *j0 = S->M[l]->j0; }

/* ---- [Chombo] PiecewiseLinearFillPatch, ratio 2, one layer of ghost cells (corners included), cells outside the
 * domain are left alone */
void or_pwl_fill(const OrModel *F, const OrModel *C, double *f, const double *c)
{
    for (int j = -1; j <= F->ny; j++)
        for (int i = -1; i <= F->nx; i++) {
            if (i >= 0 && i < F->nx && j >= 0 && j < F->ny) continue;
            int gi = i + F->i0, gj = j + F->j0;
            if (gi < 0 || gi >= F->nxg || gj < 0 || gj >= F->nyg) continue;
            int I = gi >> 1, J = gj >> 1, Ic = I - C->i0, Jc = J - C->j0;
            double c0 = G(C, c, Ic, Jc), s[2];
            for (int d = 0; d < 2; d++) {
                int K = d == 0 ? I : J, nd = d == 0 ? C->nxg : C->nyg, ii = d == 0, jj = d == 1;
                if (K - 1 >= 0 && K + 1 <= nd - 1) s[d] = 0.5 * (G(C, c, Ic + ii, Jc + jj) - G(C, c, Ic - ii, Jc - jj));   /* INTERPCENTRALSLOPE */
                else if (K - 1 < 0) s[d] = G(C, c, Ic + ii, Jc + jj) - c0;                                               /* INTERPHISIDESLOPE */
                else s[d] = c0 - G(C, c, Ic - ii, Jc - jj);                                                               /* INTERPLOSIDESLOPE */
            }
            double smax = c0, smin = c0;                                                                                 /* INTERPLIMIT */
            for (int jj = -1; jj <= 1; jj++)
                for (int ii = -1; ii <= 1; ii++) {
                    int In = I + ii, Jn = J + jj;
                    if (In < 0 || In > C->nxg - 1 || Jn < 0 || Jn > C->nyg - 1) continue;
                    double v = G(C, c, Ic + ii, Jc + jj);
                    smax = fmax(smax, v); smin = fmin(smin, v);
                }
            double deltasum = 0.5 * (fabs(s[0]) + fabs(s[1]));
            if (deltasum > 0.0) {
                double etamax = (smax - c0) / deltasum, etamin = (c0 - smin) / deltasum;
                double eta = fmax(fmin(fmin(etamin, etamax), 1.0), 0.0);
                s[0] = eta * s[0]; s[1] = eta * s[1];
            }
            double v = c0;                                                                                               /* fillConstantInterp */
            v = v + s[0] * ((gi & 1) ? 0.25 : -0.25);                                                                    /* INTERPLINEAR, dir 0 */
            v = v + s[1] * ((gj & 1) ? 0.25 : -0.25);
            G(F, f, i, j) = v;
        }
}
/* ---- [Chombo] QuadCFInterp, ratio 2: the arithmetic of oracle/amrn.c:cf_interp on the levels' own arrays */
void or_quadcf_fill(const OrModel *F, const OrModel *C, double *f, const double *c)
{
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    for (int dir = 0; dir < 2; dir++) {
        int tdir = 1 - dir;
        int ndomf = dir == 0 ? F->nxg : F->nyg, nct = tdir == 0 ? C->nxg : C->nyg;
        for (int side = 0; side < 2; side++) {
            int vlo = dir == 0 ? F->i0 : F->j0, vhi = dir == 0 ? F->i0 + F->nx - 1 : F->j0 + F->ny - 1;
            int g = side == 0 ? vlo - 1 : vhi + 1, inward = side == 0 ? 1 : -1;
            if (g < 0 || g > ndomf - 1) continue;
            int tlo = tdir == 0 ? F->i0 : F->j0, thi = tdir == 0 ? F->i0 + F->nx - 1 : F->j0 + F->ny - 1;
            for (int t = tlo; t <= thi; t++) {
                int icn = g >> 1, ict = t >> 1;
                double xt = (t & 1) ? 0.25 : -0.25;
                int have_lo = ict - 1 >= 0, have_hi = ict + 1 <= nct - 1;
#define CV(o) (dir == 0 ? G(C, c, icn - C->i0, ict + (o) - C->j0) : G(C, c, ict + (o) - C->i0, icn - C->j0))
                double c0 = CV(0), d1 = 0.0, d2 = 0.0;
                if (have_lo && have_hi) { double cm = CV(-1), cp = CV(1); d1 = 0.5 * (cp - cm); d2 = cp - 2.0 * c0 + cm; }
                else if (have_hi) { double cp = CV(1), cpp = CV(2); d1 = 0.5 * (-3.0 * c0 + 4.0 * cp - cpp); d2 = c0 - 2.0 * cp + cpp; }
                else if (have_lo) { double cm = CV(-1), cmm = CV(-2); d1 = 0.5 * (3.0 * c0 - 4.0 * cm + cmm); d2 = c0 - 2.0 * cm + cmm; }
#undef CV
                double phistar = c0 + xt * d1 + (0.5 * xt * xt) * d2;
                int ig = (dir == 0 ? g : t) - F->i0, jg = (dir == 0 ? t : g) - F->j0;
                int di = dir == 0 ? inward : 0, dj = dir == 0 ? 0 : inward;
                G(F, f, ig, jg) = c_s * phistar + c_b * G(F, f, ig + di, jg + dj) + c_a * G(F, f, ig + 2 * di, jg + 2 * dj);
            }
        }
    }
}
/* ---- [Chombo] CoarseAverage / FORT_AVERAGE: covered coarse cell = (sum of its 4 fine cells, i fastest) / 4 */
static void average_down(const OrModel *F, OrModel *C, const double *f, double *c)
{
    for (int J = 0; J < F->ny / 2; J++)
        for (int I = 0; I < F->nx / 2; I++) {
            double s = 0.0;
            for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + G(F, f, 2 * I + ii, 2 * J + jj);
            G(C, c, I + F->i0 / 2 - C->i0, J + F->j0 / 2 - C->j0) = s * 0.25;
        }
}
static int covered(const OrAmrModel *S, int l, int i, int j)        /* cell (i,j) (local) of level l lies under level l+1 */
{
    if (l >= S->nlev - 1) return 0;
    const OrModel *F = S->M[l + 1], *M = S->M[l];
    int gi = i + M->i0, gj = j + M->j0;
    return gi >= F->i0 / 2 && gi < (F->i0 + F->nx) / 2 && gj >= F->j0 / 2 && gj < (F->j0 + F->ny) / 2;
}

/* grad h, Re, Qw of level l with the inter-level fills (compute_grad_head :1610-1674, evaluate_Re_quadratic, :2703-2760) */
static void chain(OrAmrModel *S, int l)
{
    OrModel *M = S->M[l], *C = l > 0 ? S->M[l - 1] : NULL;
    if (C) or_quadcf_fill(M, C, M->c[OM_H], C->c[OM_H]);          /* inside compGradientMAC (util/Gradient.cpp) */
    or_model_grad(M);
    if (C) { or_quadcf_fill(M, C, M->c[OM_GRADX], C->c[OM_GRADX]); or_quadcf_fill(M, C, M->c[OM_GRADY], C->c[OM_GRADY]); }
    or_model_re(M);
    if (C) or_pwl_fill(M, C, M->c[OM_RE], C->c[OM_RE]);
    or_model_qw(M);
}

/* Calc_moulin_integral + Calc_moulin_source_term_distributed on the hierarchy (:1866-2066, :2797-2837): the Gaussians
 * are sampled on every level, cells under a finer level do not count in the integral and get the average of the finer
 * level's source term afterwards.  Result in OM_MSRC of every level (valid cells); integ: nm integrals. */
void or_amr_model_moulin_source(OrAmrModel *S, int nm, const double *pos, const double *sigma, const double *flux,
                                double time_factor, double *integ)
{
    const double v[3] = {0.5555555555, 0.8888888888, 0.5555555555};
    const double lq[3] = {-0.77459666924 / 2.0, 0.0, 0.77459666924 / 2.0};
    double *ms[AMAXLEV];
    for (int m = 0; m < nm; m++) integ[m] = 0.0;
    for (int l = S->nlev - 1; l >= 0; l--) {                          /* finest first (:1891) */
        OrModel *M = S->M[l];
        ms[l] = (double *)calloc((size_t)M->nx * M->ny * nm, sizeof(double));
        for (int j = 0; j < M->ny; j++)
            for (int i = 0; i < M->nx; i++) {
                if (covered(S, l, i, j)) continue;                    /* setVal(0.0, overlayBox) */
                double xl[3], yl[3];
                for (int k = 0; k < 3; k++) { xl[k] = (i + M->i0 + 0.5 + lq[k]) * M->dx; yl[k] = (j + M->j0 + 0.5 + lq[k]) * M->dy; }
                for (int m = 0; m < nm; m++) {
                    double prefac = 1.0 / (sigma[m] * sqrt(2.0 * 3.14));
                    double MS[9];
                    for (int b = 0; b < 3; b++)
                        for (int a = 0; a < 3; a++) {
                            double ex = xl[a] - pos[2 * m], ey = yl[b] - pos[2 * m + 1];
                            double rad = ex * ex + ey * ey;
                            MS[3 * b + a] = prefac * exp(-1.0 / (2.0 * sigma[m] * sigma[m]) * rad);
                        }
                    ms[l][((size_t)j * M->nx + i) * nm + m] =
                        v[0] * v[0] * MS[0] + v[1] * v[0] * MS[1] + v[2] * v[0] * MS[2] + v[0] * v[1] * MS[3] + v[1] * v[1] * MS[4]
                        + v[2] * v[1] * MS[5] + v[0] * v[2] * MS[6] + v[1] * v[2] * MS[7] + v[2] * v[2] * MS[8];
                }
            }
        for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) for (int m = 0; m < nm; m++)
            integ[m] += ms[l][((size_t)j * M->nx + i) * nm + m] * M->dx * M->dy;
    }
    for (int l = 0; l < S->nlev; l++) {
        OrModel *M = S->M[l];
        double *out = M->c[OM_MSRC];
        for (int j = 0; j < M->ny; j++)
            for (int i = 0; i < M->nx; i++) {
                double sum = 0.0;
                for (int m = 0; m < nm; m++) sum += ms[l][((size_t)j * M->nx + i) * nm + m] * time_factor / integ[m] * flux[m];
                G(M, out, i, j) = sum;
            }
        free(ms[l]);
    }
    for (int l = S->nlev - 1; l > 0; l--) average_down(S->M[l], S->M[l - 1], S->M[l]->c[OM_MSRC], S->M[l - 1]->c[OM_MSRC]);   /* :2819-2826 */
}

/* one step of the hierarchy; returns 0, -1 if the Picard loop exceeds 100 iterations, -2 for what is not restated */
int or_amr_model_timestep(OrAmrModel *S, double dt, int *picard_iters, int *vcycles_total)
{
    const int n = S->nlev;
    const int impl = S->M[0]->mp.use_impl_diff;
    double *tmp[AMAXLEV];
    for (int l = 0; l < n; l++) tmp[l] = (double *)malloc(sizeof(double) * (size_t)S->M[l]->nx * S->M[l]->ny);
    for (int l = 0; l < n; l++) {                                     /* static fields of the solver's levels (factory define) */
        OrModel *M = S->M[l];
        const int om[3] = {OM_PI, OM_ZB, OM_MASK}, of[3] = {OR_F_PI, OR_F_ZB, OR_F_MASK};
        for (int k = 0; k < 3; k++) {
            if (l == 0) or_level_set(S->base, 0, of[k], M->c[om[k]], 1);
            else or_amr_patch_io(S->A, l, of[k], M->c[om[k]], 1, 1);
        }
    }
    /* [I] */
    for (int l = 0; l < n; l++) {
        if (l > 0) or_pwl_fill(S->M[l], S->M[l - 1], S->M[l]->c[OM_B], S->M[l - 1]->c[OM_B]);
        or_model_begin_step(S->M[l]);
    }
    S->cur_step = S->M[0]->cur_step;
    OrSolverParams sp;
    or_model_solver_params(S->M[0], &sp);
    int converged = 0, ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {
        for (int l = 0; l < n; l++) {                                 /* :2482-2532 */
            OrModel *M = S->M[l];
            if (l > 0) { or_pwl_fill(M, S->M[l - 1], M->c[OM_B], S->M[l - 1]->c[OM_B]); or_pwl_fill(M, S->M[l - 1], M->c[OM_MR], S->M[l - 1]->c[OM_MR]); }
            or_model_begin_iteration(M);
        }
        for (int l = 0; l < n; l++) chain(S, l);
        for (int l = 0; l < n; l++) or_model_rhs_h(S->M[l]);
        /* SolveForHead_nl over the hierarchy */
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            for (int pass = 0; pass < 2; pass++) {
                const double *src = M->c[pass == 0 ? OM_H : OM_RHSH];
                for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) tmp[l][(size_t)j * M->nx + i] = G(M, src, i, j);
                if (l == 0) or_level_set(S->base, 0, pass == 0 ? OR_F_PHI : OR_F_RHS, tmp[l], 0);
                else or_amr_patch_io(S->A, l, pass == 0 ? OR_F_PHI : OR_F_RHS, tmp[l], 0, 1);
            }
            double *bx = (double *)malloc(sizeof(double) * (size_t)(M->nx + 1) * M->ny), *by = (double *)malloc(sizeof(double) * (size_t)M->nx * (M->ny + 1));
            or_model_bcoef(M, bx, by);                                /* aCoeff_bCoeff :3087-3102 */
            if (l == 0) { or_level_set(S->base, 0, OR_F_B, M->c[OM_B], 1); or_level_set(S->base, 0, OR_F_BX, bx, 0); or_level_set(S->base, 0, OR_F_BY, by, 0);
                          or_level_build_mg_coefficients(S->base); }
            else { or_amr_patch_io(S->A, l, OR_F_B, M->c[OM_B], 1, 1); or_amr_patch_io(S->A, l, OR_F_BX, bx, 0, 1); or_amr_patch_io(S->A, l, OR_F_BY, by, 0, 1); }
            free(bx); free(by);
        }
        nv += or_amr_solve(S->A, &sp, NULL);
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            if (l == 0) or_level_get(S->base, 0, OR_F_PHI, tmp[l], 0); else or_amr_patch_io(S->A, l, OR_F_PHI, tmp[l], 0, 0);
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) G(M, M->c[OM_H], i, j) = tmp[l][(size_t)j * M->nx + i];
        }
        for (int l = n - 1; l > 0; l--) average_down(S->M[l], S->M[l - 1], S->M[l]->c[OM_H], S->M[l - 1]->c[OM_H]);      /* :3138-3141 */
        for (int l = 0; l < n; l++) or_model_head_ghosts(S->M[l], S->M[l]->c[OM_H]);
        double maxHead = -1e300, res = 0.0;                           /* computeMax over the uncovered cells :3169-3185 */
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++)
                if (!covered(S, l, i, j) && G(M, M->c[OM_H], i, j) > maxHead) maxHead = G(M, M->c[OM_H], i, j);
        }
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) {
                if (covered(S, l, i, j)) continue;
                double d = fabs((G(M, M->c[OM_HLAG], i, j) - G(M, M->c[OM_H], i, j)) / maxHead);
                if (d > res) res = d;
            }
        }
        if (ite_idx > 100) { for (int l = 0; l < n; l++) free(tmp[l]); return -1; }
        converged = or_model_picard_converged(S->M[0], res, cur_picard);
        ite_idx++; cur_picard++;
    }
    /* [III] level by level: the coarse gap height is already updated when the fine ghosts are filled (:3252-3421) */
    double *rhs_b[AMAXLEV] = {0};
    for (int l = 0; l < n; l++) {
        OrModel *M = S->M[l];
        chain(S, l);
        if (impl) { rhs_b[l] = (double *)malloc(sizeof(double) * (size_t)M->nx * M->ny); or_model_gap_rhs(M, dt, rhs_b[l]); }
        else {
            or_model_gap_update(M, dt);
            if (l > 0) { or_pwl_fill(M, S->M[l - 1], M->c[OM_B], S->M[l - 1]->c[OM_B]); or_model_copy_ghosts(M, M->c[OM_B]); }
        }
        M->time += dt;
    }
    if (impl) {                                                       /* SolveForGap_nl over the hierarchy :3425-3455 */
        const OrModelParams *p = &S->M[0]->mp;
        if (!S->GA || S->G_dt != dt) {
            if (S->GA) { or_amr_destroy(S->GA); or_level_destroy(S->Gbase); }
            OrBC nb = S->bc;
            for (int d = 0; d < 2; d++) for (int sd = 0; sd < 2; sd++) { nb.type[d][sd] = 1; nb.value[d][sd] = 0.0; }
            OrPhys lp = S->ph; lp.use_NL = 0;
            S->Gbase = or_level_create(S->nx0, S->ny0, S->dx0, S->dy0, S->max_box, &nb, &lp, 1.0, dt * p->diffFactor, S->nthreads);
            S->GA = or_amr_create(S->Gbase, S->nx0, S->ny0, S->dx0, S->dy0, &nb, &lp, 1.0, dt * p->diffFactor, n, S->patches);
            S->G_dt = dt;
            for (int l = 0; l < n; l++) {                             /* aCoeff_GH = 1 :1820-1828 */
                OrModel *M = S->M[l];
                for (size_t k = 0; k < (size_t)M->nx * M->ny; k++) tmp[l][k] = 1.0;
                if (l == 0) { or_level_set(S->Gbase, 0, OR_F_ACOEF, tmp[l], 0); or_level_set(S->Gbase, 0, OR_F_MASK, M->c[OM_MASK], 1); }
                else { or_amr_patch_io(S->GA, l, OR_F_ACOEF, tmp[l], 0, 1); or_amr_patch_io(S->GA, l, OR_F_MASK, M->c[OM_MASK], 1, 1); }
            }
        }
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) tmp[l][(size_t)j * M->nx + i] = G(M, M->c[OM_B], i, j);
            if (l == 0) {
                or_level_set(S->Gbase, 0, OR_F_PHI, tmp[l], 0); or_level_set(S->Gbase, 0, OR_F_RHS, rhs_b[l], 0);
                or_level_set(S->Gbase, 0, OR_F_BX, or_model_dcoef(M, 0), 0); or_level_set(S->Gbase, 0, OR_F_BY, or_model_dcoef(M, 1), 0);
                or_level_build_mg_coefficients(S->Gbase);
            } else {
                or_amr_patch_io(S->GA, l, OR_F_PHI, tmp[l], 0, 1); or_amr_patch_io(S->GA, l, OR_F_RHS, rhs_b[l], 0, 1);
                or_amr_patch_io(S->GA, l, OR_F_BX, (double *)or_model_dcoef(M, 0), 0, 1); or_amr_patch_io(S->GA, l, OR_F_BY, (double *)or_model_dcoef(M, 1), 0, 1);
            }
        }
        OrSolverParams spg;
        spg.num_smooth = 2; spg.num_bottom = 4; spg.max_iter = 100; spg.iter_min = 2; spg.imin = S->M[0]->cur_step < 50 ? 10 : 5;
        spg.eps = 1.0e-7; spg.hang = 1.0e-6; spg.norm_thresh = 1.0e-7; spg.bcoeff_otf = 0; spg.max_depth = -1;
        (void)or_amr_solve(S->GA, &spg, NULL);
        for (int l = 0; l < n; l++) {
            OrModel *M = S->M[l];
            if (l == 0) or_level_get(S->Gbase, 0, OR_F_PHI, tmp[l], 0); else or_amr_patch_io(S->GA, l, OR_F_PHI, tmp[l], 0, 0);
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) G(M, M->c[OM_B], i, j) = tmp[l][(size_t)j * M->nx + i];
            if (l > 0) or_pwl_fill(M, S->M[l - 1], M->c[OM_B], S->M[l - 1]->c[OM_B]);
            or_model_copy_ghosts(M, M->c[OM_B]);
            free(rhs_b[l]);
        }
    }
    for (int l = 0; l < n; l++) free(tmp[l]);
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles_total) *vcycles_total = nv;
    return 0;
}
