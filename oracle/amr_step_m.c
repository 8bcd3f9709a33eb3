/*
 * amr_step_m.c -- TEST INFRASTRUCTURE ONLY.  One AmrHydro::timeStepFAS (src/AmrHydro.cpp:2254-3460) on a hierarchy whose
 * levels >= 1 are UNIONS OF BOXES (oracle/amrm.c): amr_step.c with every level's rectangle replaced by its list of boxes.
 * Every box is an OrModel of its own (time_loop.c) whose four sides inside the domain carry data ghosts; after every fill
 * of such ghosts the reference's exchange() follows here as the fine-fine copy between the boxes of the level:
 *   PiecewiseLinearFillPatch of b, mR (:2373-2380, :2499-2507), Re (:2711-2719) + exchange (:2385, :2513, :2721)
 *   QuadCFInterp of h (inside compGradientMAC, coverage-aware stencils of amrm.c) and of the cell-centred gradient
 *     (:1650-1656) + exchange (:1659)
 *   SolveForHead_nl over the hierarchy = or_amrm_solve, CoarseAverage of h (:3138-3141)
 *   computeMax over the cells no finer level covers (:3169, :3185)
 * With one box per level this file IS amr_step.c bit for bit (tests/test_oracle_amr_step_m.py).  [Chombo] pieces as there:
 * unpinned.
 */
#include "time_loop.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* oracle/amrm.c */
typedef struct OrAmrM OrAmrM;
OrAmrM *or_amrm_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                       double alpha, double beta, int nlev, const int *nbox, const int *boxes);
void or_amrm_destroy(OrAmrM *A);
void or_amrm_box_io(OrAmrM *A, int l, int k, int field, double *g, int ghosted, int set);
int or_amrm_solve(OrAmrM *A, const OrSolverParams *sp, double *hist);
int or_amrm_owner(const OrAmrM *A, int l, int i, int j);
void or_amrm_exchange_fabs(OrAmrM *A, int l, OrFab **fabs, int corners);
void or_amrm_cf_interp_fabs(OrAmrM *A, int l, OrFab **fabs, int comp, const double *coarse);
/* oracle/amr_step.c */
void or_pwl_fill(const OrModel *F, const OrModel *C, double *f, const double *c);

#define AMAXLEV 8
typedef struct OrAmrMModel {
    int nlev;
    OrLevel *base;
    OrAmrM *A;
    int nbox[AMAXLEV];
    OrModel **M[AMAXLEV];                 /* M[l][k] */
    int *boxes; int nboxes_total;
    int nxd[AMAXLEV], nyd[AMAXLEV];
    OrLevel *Gbase; OrAmrM *GA; double G_dt;
    int nx0, ny0, max_box, nthreads;
    double dx0, dy0;
    OrBC bc; OrPhys ph;
} OrAmrMModel;

#define G(M, a, i, j) (a)[(size_t)((j) + 1) * ((M)->nx + 2) + ((i) + 1)]      /* ghosted array of a box, LOCAL indices */

OrAmrMModel *or_amrm_model_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                                  const OrModelParams *mp, int nlev, const int *nbox, const int *boxes)
{
    OrAmrMModel *S = (OrAmrMModel *)calloc(1, sizeof(OrAmrMModel));
    S->nlev = nlev; S->base = base;
    S->nx0 = nx0; S->ny0 = ny0; S->dx0 = dx0; S->dy0 = dy0; S->bc = *bc; S->ph = *ph; S->max_box = 64; S->nthreads = 1;
    S->A = or_amrm_create(base, nx0, ny0, dx0, dy0, bc, ph, 0.0, -1.0, nlev, nbox, boxes);
    if (!S->A) { free(S); return NULL; }
    int tot = 0;
    for (int l = 1; l < nlev; l++) tot += nbox[l];
    S->nboxes_total = tot;
    S->boxes = (int *)malloc(sizeof(int) * 4 * (size_t)(tot > 0 ? tot : 1));
    memcpy(S->boxes, boxes, sizeof(int) * 4 * (size_t)tot);
    S->nbox[0] = 1; S->nxd[0] = nx0; S->nyd[0] = ny0;
    S->M[0] = (OrModel **)malloc(sizeof(OrModel *));
    S->M[0][0] = or_model_create(base, nx0, ny0, dx0, dy0, bc, ph, mp);
    double dx = dx0, dy = dy0;
    const int *q = boxes;
    for (int l = 1; l < nlev; l++) {
        S->nbox[l] = nbox[l]; S->nxd[l] = 2 * S->nxd[l - 1]; S->nyd[l] = 2 * S->nyd[l - 1];
        dx /= 2.0; dy /= 2.0;
        S->M[l] = (OrModel **)malloc(sizeof(OrModel *) * (size_t)nbox[l]);
        for (int k = 0; k < nbox[l]; k++, q += 4) {
            S->M[l][k] = or_model_create(NULL, q[2] - q[0] + 1, q[3] - q[1] + 1, dx, dy, bc, ph, mp);
            or_model_set_patch(S->M[l][k], q[0], q[1], S->nxd[l], S->nyd[l]);
        }
    }
    return S;
}
void or_amrm_model_destroy(OrAmrMModel *S)
{
    if (!S) return;
    for (int l = 0; l < S->nlev; l++) { for (int k = 0; k < S->nbox[l]; k++) or_model_destroy(S->M[l][k]); free(S->M[l]); }
    or_amrm_destroy(S->A);
    if (S->GA) or_amrm_destroy(S->GA);
    if (S->Gbase) or_level_destroy(S->Gbase);
    free(S->boxes);
    free(S);
}
OrModel *or_amrm_model_box(OrAmrMModel *S, int l, int k) { return S->M[l][k]; }
double *or_amrm_model_field(OrAmrMModel *S, int l, int k, int id) { return or_model_field(S->M[l][k], id); }
void or_amrm_model_gap_solver_layout(OrAmrMModel *S, int max_box, int nthreads) { S->max_box = max_box; S->nthreads = nthreads; }

/* ---- level-wide helpers ---- */
/* valid cells of a field of level l over its DOMAIN (zero where the level has no box) */
static double *dom_valid(OrAmrMModel *S, int l, int fid)
{
    int nx = S->nxd[l], ny = S->nyd[l];
    double *a = (double *)calloc((size_t)nx * ny, sizeof(double));
    for (int k = 0; k < S->nbox[l]; k++) {
        OrModel *M = S->M[l][k];
        const double *c = M->c[fid];
        for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) a[(size_t)(j + M->j0) * nx + (i + M->i0)] = G(M, c, i, j);
    }
    return a;
}
/* the same with a (zero) ghost ring and a stand-in model that spans the level's domain: the coarse side of or_pwl_fill */
static double *dom_ghosted(OrAmrMModel *S, int l, int fid, OrModel *stub)
{
    int nx = S->nxd[l], ny = S->nyd[l];
    memset(stub, 0, sizeof(*stub));
    stub->nx = nx; stub->ny = ny; stub->i0 = 0; stub->j0 = 0; stub->nxg = nx; stub->nyg = ny;
    double *a = (double *)calloc((size_t)(nx + 2) * (ny + 2), sizeof(double));
    for (int k = 0; k < S->nbox[l]; k++) {
        OrModel *M = S->M[l][k];
        const double *c = M->c[fid];
        for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) a[(size_t)(j + M->j0 + 1) * (nx + 2) + (i + M->i0 + 1)] = G(M, c, i, j);
    }
    return a;
}
/* the boxes' ghosted arrays of one field as fabs over the level's index space */
static OrFab **alias_fabs(OrAmrMModel *S, int l, int fid)
{
    int n = S->nbox[l];
    OrFab *store = (OrFab *)malloc(sizeof(OrFab) * (size_t)n);
    OrFab **p = (OrFab **)malloc(sizeof(OrFab *) * (size_t)(n + 1));
    for (int k = 0; k < n; k++) {
        OrModel *M = S->M[l][k];
        store[k].p = M->c[fid]; store[k].ncomp = 1;
        store[k].lo0 = M->i0 - 1; store[k].lo1 = M->j0 - 1; store[k].hi0 = M->i0 + M->nx; store[k].hi1 = M->j0 + M->ny;
        p[k] = &store[k];
    }
    p[n] = store;                                        /* kept for free_fabs */
    return p;
}
static void free_fabs(OrAmrMModel *S, int l, OrFab **p) { free(p[S->nbox[l]]); free(p); }
/* exchange(): fine-fine ghost cells of a field of level l */
static void mm_ff(OrAmrMModel *S, int l, int fid, int corners)
{
    if (l == 0) return;
    OrFab **f = alias_fabs(S, l, fid);
    or_amrm_exchange_fabs(S->A, l, f, corners);
    free_fabs(S, l, f);
}
/* QuadCFInterp of a field of level l from level l-1, then the exchange */
static void mm_quadcf(OrAmrMModel *S, int l, int fid, int corners)
{
    if (l == 0) return;
    double *c = dom_valid(S, l - 1, fid);
    OrFab **f = alias_fabs(S, l, fid);
    or_amrm_cf_interp_fabs(S->A, l, f, 0, c);
    or_amrm_exchange_fabs(S->A, l, f, corners);
    free_fabs(S, l, f);
    free(c);
}
/* PiecewiseLinearFillPatch of a field of level l from level l-1, then the exchange (which overwrites the fine-fine cells) */
static void mm_pwl(OrAmrMModel *S, int l, int fid)
{
    if (l == 0) return;
    OrModel stub;
    double *c = dom_ghosted(S, l - 1, fid, &stub);
    for (int k = 0; k < S->nbox[l]; k++) or_pwl_fill(S->M[l][k], &stub, S->M[l][k]->c[fid], c);
    free(c);
    mm_ff(S, l, fid, 1);
}
/* [Chombo] CoarseAverage: covered cells of level l-1 <- average of the 4 fine cells */
static void mm_average_down(OrAmrMModel *S, int l, int fid)
{
    for (int k = 0; k < S->nbox[l]; k++) {
        OrModel *F = S->M[l][k];
        const double *f = F->c[fid];
        for (int J = 0; J < F->ny / 2; J++)
            for (int I = 0; I < F->nx / 2; I++) {
                double s = 0.0;
                for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + G(F, f, 2 * I + ii, 2 * J + jj);
                int gi = I + F->i0 / 2, gj = J + F->j0 / 2;
                int o = l - 1 == 0 ? 0 : or_amrm_owner(S->A, l - 1, gi, gj);
                OrModel *C = S->M[l - 1][o];
                G(C, C->c[fid], gi - C->i0, gj - C->j0) = s * 0.25;
            }
    }
}
static int covered(const OrAmrMModel *S, int l, const OrModel *M, int i, int j)   /* cell (i,j) (local) of a box of level l lies under level l+1 */
{
    if (l >= S->nlev - 1) return 0;
    return or_amrm_owner(S->A, l + 1, 2 * (i + M->i0), 2 * (j + M->j0)) >= 0;
}

/* grad h, Re, Qw of level l with the inter-level fills and the exchanges between its boxes */
static void chain(OrAmrMModel *S, int l)
{
    mm_quadcf(S, l, OM_H, 0);                                   /* inside compGradientMAC */
    for (int k = 0; k < S->nbox[l]; k++) or_model_grad(S->M[l][k]);
    mm_quadcf(S, l, OM_GRADX, 1); mm_quadcf(S, l, OM_GRADY, 1);  /* :1650-1659 */
    for (int k = 0; k < S->nbox[l]; k++) or_model_re(S->M[l][k]);
    mm_pwl(S, l, OM_RE);                                        /* :2711-2721 */
    for (int k = 0; k < S->nbox[l]; k++) or_model_qw(S->M[l][k]);
}

/* Calc_moulin_integral + Calc_moulin_source_term_distributed on the hierarchy (:1866-2066, :2797-2837), as amr_step.c */
void or_amrm_model_moulin_source(OrAmrMModel *S, int nm, const double *pos, const double *sigma, const double *flux,
                                 double time_factor, double *integ)
{
    const double v[3] = {0.5555555555, 0.8888888888, 0.5555555555};
    const double lq[3] = {-0.77459666924 / 2.0, 0.0, 0.77459666924 / 2.0};
    double **ms[AMAXLEV];
    for (int m = 0; m < nm; m++) integ[m] = 0.0;
    for (int l = S->nlev - 1; l >= 0; l--) {                          /* finest first (:1891) */
        ms[l] = (double **)malloc(sizeof(double *) * (size_t)S->nbox[l]);
        for (int k = 0; k < S->nbox[l]; k++) {
            OrModel *M = S->M[l][k];
            ms[l][k] = (double *)calloc((size_t)M->nx * M->ny * nm, sizeof(double));
            for (int j = 0; j < M->ny; j++)
                for (int i = 0; i < M->nx; i++) {
                    if (covered(S, l, M, i, j)) continue;                 /* setVal(0.0, overlayBox) */
                    double xl[3], yl[3];
                    for (int a = 0; a < 3; a++) { xl[a] = (i + M->i0 + 0.5 + lq[a]) * M->dx; yl[a] = (j + M->j0 + 0.5 + lq[a]) * M->dy; }
                    for (int m = 0; m < nm; m++) {
                        double prefac = 1.0 / (sigma[m] * sqrt(2.0 * 3.14));
                        double MS[9];
                        for (int b = 0; b < 3; b++)
                            for (int a = 0; a < 3; a++) {
                                double ex = xl[a] - pos[2 * m], ey = yl[b] - pos[2 * m + 1];
                                double rad = ex * ex + ey * ey;
                                MS[3 * b + a] = prefac * exp(-1.0 / (2.0 * sigma[m] * sigma[m]) * rad);
                            }
                        ms[l][k][((size_t)j * M->nx + i) * nm + m] =
                            v[0] * v[0] * MS[0] + v[1] * v[0] * MS[1] + v[2] * v[0] * MS[2] + v[0] * v[1] * MS[3] + v[1] * v[1] * MS[4]
                            + v[2] * v[1] * MS[5] + v[0] * v[2] * MS[6] + v[1] * v[2] * MS[7] + v[2] * v[2] * MS[8];
                    }
                }
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) for (int m = 0; m < nm; m++)
                integ[m] += ms[l][k][((size_t)j * M->nx + i) * nm + m] * M->dx * M->dy;
        }
    }
    for (int l = 0; l < S->nlev; l++) {
        for (int k = 0; k < S->nbox[l]; k++) {
            OrModel *M = S->M[l][k];
            double *out = M->c[OM_MSRC];
            for (int j = 0; j < M->ny; j++)
                for (int i = 0; i < M->nx; i++) {
                    double sum = 0.0;
                    for (int m = 0; m < nm; m++) sum += ms[l][k][((size_t)j * M->nx + i) * nm + m] * time_factor / integ[m] * flux[m];
                    G(M, out, i, j) = sum;
                }
            free(ms[l][k]);
        }
        free(ms[l]);
    }
    for (int l = S->nlev - 1; l > 0; l--) mm_average_down(S, l, OM_MSRC);   /* :2819-2826 */
}

static void to_solver(OrAmrMModel *S, OrAmrM *A, OrLevel *base, int l, int k, int field, double *g, int ghosted)
{
    if (l == 0) or_level_set(base, 0, field, g, ghosted); else or_amrm_box_io(A, l, k, field, g, ghosted, 1);
}
static void from_solver(OrAmrMModel *S, OrAmrM *A, OrLevel *base, int l, int k, int field, double *g)
{
    if (l == 0) or_level_get(base, 0, field, g, 0); else or_amrm_box_io(A, l, k, field, g, 0, 0);
}

/* one step of the hierarchy; returns 0, -1 if the Picard loop exceeds 100 iterations */
int or_amrm_model_timestep(OrAmrMModel *S, double dt, int *picard_iters, int *vcycles_total)
{
    const int n = S->nlev;
    OrModel *M0 = S->M[0][0];
    const int impl = M0->mp.use_impl_diff;
    size_t maxc = 0;
    for (int l = 0; l < n; l++) for (int k = 0; k < S->nbox[l]; k++) { size_t c = (size_t)S->M[l][k]->nx * S->M[l][k]->ny; if (c > maxc) maxc = c; }
    double *tmp = (double *)malloc(sizeof(double) * maxc);
#define EACH(l, k, M) for (int k = 0; k < S->nbox[l]; k++) for (OrModel *M = S->M[l][k]; M; M = NULL)
    for (int l = 0; l < n; l++) EACH(l, k, M) {                       /* static fields of the solver's levels (factory define) */
        const int om[3] = {OM_PI, OM_ZB, OM_MASK}, of[3] = {OR_F_PI, OR_F_ZB, OR_F_MASK};
        for (int q = 0; q < 3; q++) to_solver(S, S->A, S->base, l, k, of[q], M->c[om[q]], 1);
    }
    /* [I] */
    for (int l = 0; l < n; l++) {
        mm_pwl(S, l, OM_B);
        EACH(l, k, M) or_model_begin_step(M);
    }
    OrSolverParams sp;
    or_model_solver_params(M0, &sp);
    int converged = 0, ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {
        for (int l = 0; l < n; l++) {                                 /* :2482-2532 */
            mm_pwl(S, l, OM_B); mm_pwl(S, l, OM_MR);
            EACH(l, k, M) or_model_begin_iteration(M);
        }
        for (int l = 0; l < n; l++) chain(S, l);
        for (int l = 0; l < n; l++) EACH(l, k, M) or_model_rhs_h(M);
        /* SolveForHead_nl over the hierarchy */
        for (int l = 0; l < n; l++) EACH(l, k, M) {
            for (int pass = 0; pass < 2; pass++) {
                const double *src = M->c[pass == 0 ? OM_H : OM_RHSH];
                for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) tmp[(size_t)j * M->nx + i] = G(M, src, i, j);
                to_solver(S, S->A, S->base, l, k, pass == 0 ? OR_F_PHI : OR_F_RHS, tmp, 0);
            }
            to_solver(S, S->A, S->base, l, k, OR_F_B, M->c[OM_B], 1);
            {
                double *bx = (double *)malloc(sizeof(double) * (size_t)(M->nx + 1) * M->ny), *by = (double *)malloc(sizeof(double) * (size_t)M->nx * (M->ny + 1));
                or_model_bcoef(M, bx, by);                            /* aCoeff_bCoeff :3087-3102 */
                to_solver(S, S->A, S->base, l, k, OR_F_BX, bx, 0); to_solver(S, S->A, S->base, l, k, OR_F_BY, by, 0);
                free(bx); free(by);
            }
            if (l == 0) or_level_build_mg_coefficients(S->base);
        }
        nv += or_amrm_solve(S->A, &sp, NULL);
        for (int l = 0; l < n; l++) EACH(l, k, M) {
            from_solver(S, S->A, S->base, l, k, OR_F_PHI, tmp);
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) G(M, M->c[OM_H], i, j) = tmp[(size_t)j * M->nx + i];
        }
        for (int l = n - 1; l > 0; l--) mm_average_down(S, l, OM_H);   /* :3138-3141 */
        for (int l = 0; l < n; l++) EACH(l, k, M) or_model_head_ghosts(M, M->c[OM_H]);
        double maxHead = -1e300, res = 0.0;                           /* computeMax over the uncovered cells :3169-3185 */
        for (int l = 0; l < n; l++) EACH(l, k, M)
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++)
                if (!covered(S, l, M, i, j) && G(M, M->c[OM_H], i, j) > maxHead) maxHead = G(M, M->c[OM_H], i, j);
        for (int l = 0; l < n; l++) EACH(l, k, M)
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) {
                if (covered(S, l, M, i, j)) continue;
                double d = fabs((G(M, M->c[OM_HLAG], i, j) - G(M, M->c[OM_H], i, j)) / maxHead);
                if (d > res) res = d;
            }
        if (ite_idx > 100) { free(tmp); return -1; }
        converged = or_model_picard_converged(M0, res, cur_picard);
        ite_idx++; cur_picard++;
    }
    /* [III] level by level: the coarse gap height is already updated when the fine ghosts are filled (:3252-3421) */
    double **rhs_b[AMAXLEV] = {0};
    for (int l = 0; l < n; l++) {
        chain(S, l);
        if (impl) {
            rhs_b[l] = (double **)malloc(sizeof(double *) * (size_t)S->nbox[l]);
            EACH(l, k, M) { rhs_b[l][k] = (double *)malloc(sizeof(double) * (size_t)M->nx * M->ny); or_model_gap_rhs(M, dt, rhs_b[l][k]); }
        } else {
            EACH(l, k, M) or_model_gap_update(M, dt);
            if (l > 0) { mm_pwl(S, l, OM_B); EACH(l, k, M) or_model_copy_ghosts(M, M->c[OM_B]); }
        }
        EACH(l, k, M) M->time += dt;
    }
    if (impl) {                                                       /* SolveForGap_nl over the hierarchy :3425-3455 */
        const OrModelParams *p = &M0->mp;
        if (!S->GA || S->G_dt != dt) {
            if (S->GA) { or_amrm_destroy(S->GA); or_level_destroy(S->Gbase); }
            OrBC nb = S->bc;
            for (int d = 0; d < 2; d++) for (int sd = 0; sd < 2; sd++) { nb.type[d][sd] = 1; nb.value[d][sd] = 0.0; }
            OrPhys lp = S->ph; lp.use_NL = 0;
            S->Gbase = or_level_create(S->nx0, S->ny0, S->dx0, S->dy0, S->max_box, &nb, &lp, 1.0, dt * p->diffFactor, S->nthreads);
            S->GA = or_amrm_create(S->Gbase, S->nx0, S->ny0, S->dx0, S->dy0, &nb, &lp, 1.0, dt * p->diffFactor, n, S->nbox, S->boxes);
            S->G_dt = dt;
            for (int l = 0; l < n; l++) EACH(l, k, M) {                /* aCoeff_GH = 1 :1820-1828 */
                for (size_t q = 0; q < (size_t)M->nx * M->ny; q++) tmp[q] = 1.0;
                to_solver(S, S->GA, S->Gbase, l, k, OR_F_ACOEF, tmp, 0);
                to_solver(S, S->GA, S->Gbase, l, k, OR_F_MASK, M->c[OM_MASK], 1);
            }
        }
        for (int l = 0; l < n; l++) EACH(l, k, M) {
            for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) tmp[(size_t)j * M->nx + i] = G(M, M->c[OM_B], i, j);
            to_solver(S, S->GA, S->Gbase, l, k, OR_F_PHI, tmp, 0);
            to_solver(S, S->GA, S->Gbase, l, k, OR_F_RHS, rhs_b[l][k], 0);
            to_solver(S, S->GA, S->Gbase, l, k, OR_F_BX, (double *)or_model_dcoef(M, 0), 0);
            to_solver(S, S->GA, S->Gbase, l, k, OR_F_BY, (double *)or_model_dcoef(M, 1), 0);
            if (l == 0) or_level_build_mg_coefficients(S->Gbase);
        }
        OrSolverParams spg;
        spg.num_smooth = 2; spg.num_bottom = 4; spg.max_iter = 100; spg.iter_min = 2; spg.imin = M0->cur_step < 50 ? 10 : 5;
        spg.eps = 1.0e-7; spg.hang = 1.0e-6; spg.norm_thresh = 1.0e-7; spg.bcoeff_otf = 0; spg.max_depth = -1;
        (void)or_amrm_solve(S->GA, &spg, NULL);
        for (int l = 0; l < n; l++) {
            EACH(l, k, M) {
                from_solver(S, S->GA, S->Gbase, l, k, OR_F_PHI, tmp);
                for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) G(M, M->c[OM_B], i, j) = tmp[(size_t)j * M->nx + i];
                free(rhs_b[l][k]);
            }
            mm_pwl(S, l, OM_B);
            EACH(l, k, M) or_model_copy_ghosts(M, M->c[OM_B]);
            free(rhs_b[l]);
        }
    }
#undef EACH
    free(tmp);
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles_total) *vcycles_total = nv;
    return 0;
}
