/*
 * amr2.c -- TEST INFRASTRUCTURE ONLY.  Two-level AMR head solve: base level (an OrLevel with its
 * multigrid depths, level_shim.c) + ONE rectangular fine patch refined by 2 (cfg3 of BASELINE.json:
 * exec/0_convergence_channelized/2lev_base, a fixed refined box around the moulin).
 *
 * What is restated from the reference's own source:
 *   relaxNF / AMRResidualNF / AMROperator / AMRRestrictS / AMRProlongS_2 / AMRNorm
 *                                              src/AMRNonLinearPoissonOp.cpp:690-704, 889-1069, 1143-1264
 *   reflux + getFlux                           src/VCAMRNonLinearPoissonOp.cpp:555-652, 792-841
 *   UpdateOperator / WFlx_level with a coarser level   src/VCAMRNonLinearPoissonOp.cpp:34-64,
 *                                              src/AmrHydro.cpp:1415-1539 (coarse branch :1455-1488)
 * What is NOT in the reference's tree and is restated from upstream Chombo's documented semantics
 * ("parity unpinned"; every such piece is marked [Chombo] below):
 *   QuadCFInterp (quadratic coarse-fine ghost interpolation), LevelFluxRegister (reflux bookkeeping),
 *   FORT_AVERAGE, the copyTo + CornerCopier of AMRProlongS_2, and the AMR FAS cycle ordering
 *   (VCycleAMR).
 * The fine patch is ONE box (the box decomposition never changes a bit, see level_shim.h), so every
 * fine-level method is the per-box kernel of suhmo_oracle.c applied to that box.
 */
#include "level_shim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(f, i, j, n) (*or_at((f), (i), (j), (n)))

typedef struct OrAmr2 {
    OrLevel *C;                       /* base level, nxc x nyc */
    int nxc, nyc, nxf, nyf;           /* coarse / fine DOMAIN sizes */
    double dxc[2], dxf[2];
    OrBC bc; OrPhys ph; double alpha, beta;
    int ci0, cj0, ci1, cj1;           /* patch in coarse indices (inclusive) */
    OrBox fb;                         /* patch in fine indices */
    OrFab phi, rhs, acoef, B, Pi, zb, mask, bx, by, lam, nl, dnl, res, lphi, gradH, Re;
    int lambda_dirty;
    /* coarse work arrays, valid cells nyc x nxc */
    double *phic, *rhs0, *lphic, *resc, *phiold, *corr, *gxc, *gyc;
    long cf_interps, refluxes;
} OrAmr2;

static OrFab fab_alloc(OrBox b, int g, int ncomp)
{
    OrFab f;
    f.lo0 = b.lo0 - g; f.lo1 = b.lo1 - g; f.hi0 = b.hi0 + g; f.hi1 = b.hi1 + g; f.ncomp = ncomp;
    f.p = (double *)calloc((size_t)(f.hi0 - f.lo0 + 1) * (size_t)(f.hi1 - f.lo1 + 1) * (size_t)ncomp, sizeof(double));
    return f;
}
#define CC(a, i, j) (a)[(size_t)(j) * A->nxc + (i)]       /* coarse valid-cell arrays */

OrAmr2 *or_amr2_create(OrLevel *coarse, int nxc, int nyc, double dxc, double dyc, const OrBC *bc, const OrPhys *ph,
                       double alpha, double beta, int ci0, int cj0, int ci1, int cj1)
{
    OrAmr2 *A = (OrAmr2 *)calloc(1, sizeof(OrAmr2));
    A->C = coarse; A->nxc = nxc; A->nyc = nyc; A->nxf = 2 * nxc; A->nyf = 2 * nyc;
    A->dxc[0] = dxc; A->dxc[1] = dyc; A->dxf[0] = dxc / 2.0; A->dxf[1] = dyc / 2.0;   /* refRatio 2 */
    A->bc = *bc; A->ph = *ph; A->alpha = alpha; A->beta = beta;
    A->ci0 = ci0; A->cj0 = cj0; A->ci1 = ci1; A->cj1 = cj1;
    A->fb.lo0 = 2 * ci0; A->fb.lo1 = 2 * cj0; A->fb.hi0 = 2 * ci1 + 1; A->fb.hi1 = 2 * cj1 + 1;
    OrBox v = A->fb, fx = v, fy = v; fx.hi0 += 1; fy.hi1 += 1;
    A->phi = fab_alloc(v, 1, 1); A->rhs = fab_alloc(v, 0, 1); A->acoef = fab_alloc(v, 0, 1);
    A->B = fab_alloc(v, 1, 1); A->Pi = fab_alloc(v, 1, 1); A->zb = fab_alloc(v, 1, 1); A->mask = fab_alloc(v, 1, 1);
    A->bx = fab_alloc(fx, 0, 1); A->by = fab_alloc(fy, 0, 1);
    A->lam = fab_alloc(v, 0, 1); A->nl = fab_alloc(v, 0, 1); A->dnl = fab_alloc(v, 0, 1);
    A->res = fab_alloc(v, 0, 1); A->lphi = fab_alloc(v, 0, 1);
    A->gradH = fab_alloc(v, 1, 2); A->Re = fab_alloc(v, 1, 1);
    A->lambda_dirty = 1;
    size_t nc = (size_t)nxc * nyc;
    double **w[] = {&A->phic, &A->rhs0, &A->lphic, &A->resc, &A->phiold, &A->corr, &A->gxc, &A->gyc};
    for (int k = 0; k < 8; k++) *w[k] = (double *)calloc(k >= 6 ? (size_t)(nxc + 2) * (nyc + 2) : nc, sizeof(double));
    return A;
}
void or_amr2_destroy(OrAmr2 *A)
{
    if (!A) return;
    OrFab *f[] = {&A->phi, &A->rhs, &A->acoef, &A->B, &A->Pi, &A->zb, &A->mask, &A->bx, &A->by, &A->lam, &A->nl,
                  &A->dnl, &A->res, &A->lphi, &A->gradH, &A->Re};
    for (int k = 0; k < 16; k++) free(f[k]->p);
    free(A->phic); free(A->rhs0); free(A->lphic); free(A->resc); free(A->phiold); free(A->corr); free(A->gxc); free(A->gyc);
    free(A);
}

/* fine-patch fields <-> arrays of the patch's size: (nyp x nxp), ghosted (nyp+2) x (nxp+2), BX nyp x (nxp+1), BY */
static OrFab *fine_field(OrAmr2 *A, int field)
{
    switch (field) {
    case OR_F_PHI: return &A->phi; case OR_F_RHS: return &A->rhs; case OR_F_ACOEF: return &A->acoef;
    case OR_F_B: return &A->B; case OR_F_PI: return &A->Pi; case OR_F_ZB: return &A->zb; case OR_F_MASK: return &A->mask;
    case OR_F_BX: return &A->bx; case OR_F_BY: return &A->by; case OR_F_LAMBDA: return &A->lam;
    case OR_F_RES: return &A->res; case OR_F_LPHI: return &A->lphi; case OR_F_NL: return &A->nl; case OR_F_DNL: return &A->dnl;
    }
    return NULL;
}
void or_amr2_fine_io(OrAmr2 *A, int field, double *g, int ghosted, int set)
{
    OrFab *f = fine_field(A, field);
    int gf = (ghosted && f->lo0 < A->fb.lo0) ? 1 : 0;
    int lo0 = (field == OR_F_BX || field == OR_F_BY) ? f->lo0 : A->fb.lo0 - gf, hi0 = (field == OR_F_BX || field == OR_F_BY) ? f->hi0 : A->fb.hi0 + gf;
    int lo1 = (field == OR_F_BX || field == OR_F_BY) ? f->lo1 : A->fb.lo1 - gf, hi1 = (field == OR_F_BX || field == OR_F_BY) ? f->hi1 : A->fb.hi1 + gf;
    long pitch = hi0 - lo0 + 1;
    for (int j = lo1; j <= hi1; j++)
        for (int i = lo0; i <= hi0; i++) {
            double *q = &g[(long)(j - lo1) * pitch + (i - lo0)];
            if (set) AT(f, i, j, 0) = *q; else *q = AT(f, i, j, 0);
        }
    if (set && (field == OR_F_ACOEF || field == OR_F_BX || field == OR_F_BY)) A->lambda_dirty = 1;
}

/* ---------------- fine-patch operator methods (one box) ---------------- */
/* mixBCValues on the sides of the patch that lie on the domain boundary (src/AmrHydro.cpp:248-309) */
static void fine_bc(const OrAmr2 *A, OrFab *state, int homogeneous, const double dx[2], int ndx, int ndy, OrBox valid)
{
    for (int dir = 0; dir < 2; dir++) {
        if (A->bc.periodic[dir]) continue;
        int ndom = dir == 0 ? ndx : ndy;
        for (int side = 0; side < 2; side++) {
            int vlo = dir == 0 ? valid.lo0 : valid.lo1, vhi = dir == 0 ? valid.hi0 : valid.hi1;
            int g = side == 0 ? vlo - 1 : vhi + 1;
            if (g >= 0 && g <= ndom - 1) continue;
            int isign = side == 0 ? -1 : 1, type = A->bc.type[dir][side];
            double value = homogeneous ? 0.0 : A->bc.value[dir][side];
            int tlo = dir == 0 ? valid.lo1 : valid.lo0, thi = dir == 0 ? valid.hi1 : valid.hi0;
            for (int t = tlo; t <= thi; t++) {
                int ig = dir == 0 ? g : t, jg = dir == 0 ? t : g;
                int in = dir == 0 ? g - isign : t, jn = dir == 0 ? t : g - isign;
                double nearVal = AT(state, in, jn, 0);
                if (type == 0) AT(state, ig, jg, 0) = 2.0 * value - nearVal;
                else { double gv = nearVal; if (!homogeneous) gv += (double)isign * dx[dir] * value; AT(state, ig, jg, 0) = gv; }
            }
        }
    }
}
static void fine_nonlinear(OrAmr2 *A)
{
    if (!A->ph.use_NL) { memset(A->nl.p, 0, sizeof(double) * (size_t)(A->fb.hi0 - A->fb.lo0 + 1) * (A->fb.hi1 - A->fb.lo1 + 1));
                         memset(A->dnl.p, 0, sizeof(double) * (size_t)(A->fb.hi0 - A->fb.lo0 + 1) * (A->fb.hi1 - A->fb.lo1 + 1)); return; }
    or_computenonlinearterms(&A->phi, &A->B, &A->mask, &A->Pi, &A->zb, A->fb, &A->nl, &A->dnl, &A->ph);
}
static void fine_reset_lambda(OrAmr2 *A)
{
    if (!A->lambda_dirty) return;
    for (int j = A->fb.lo1; j <= A->fb.hi1; j++)
        for (int i = A->fb.lo0; i <= A->fb.hi0; i++) AT(&A->lam, i, j, 0) = AT(&A->acoef, i, j, 0) * A->alpha;
    for (int dir = 0; dir < 2; dir++)
        or_sumfacesnl(&A->lam, A->beta, dir == 0 ? &A->bx : &A->by, A->fb, dir, 1.0 / (A->dxf[dir] * A->dxf[dir]));
    A->lambda_dirty = 0;
}
/* relax(): levelGSRB x sweeps; coarse-fine ghosts keep the values of the last coarseFineInterp */
void or_amr2_fine_gsrb(OrAmr2 *A, int sweeps)
{
    for (int it = 0; it < sweeps; it++) {
        fine_reset_lambda(A);
        for (int pass = 0; pass <= 1; pass++) {
            fine_bc(A, &A->phi, 0, A->dxf, A->nxf, A->nyf, A->fb);
            fine_nonlinear(A);
            or_gsrbhelmholtzvcnl2d(&A->phi, &A->rhs, A->fb, A->dxf, A->alpha, &A->acoef, A->beta, &A->bx, &A->by,
                                   &A->nl, &A->dnl, &A->lam, pass);
        }
        fine_bc(A, &A->phi, 1, A->dxf, A->nxf, A->nyf, A->fb);
    }
}
void or_amr2_fine_apply_op(OrAmr2 *A, int homogeneous)
{
    fine_bc(A, &A->phi, homogeneous, A->dxf, A->nxf, A->nyf, A->fb);
    fine_nonlinear(A);
    or_vcnlcomputeop2d(&A->lphi, &A->phi, A->alpha, &A->acoef, A->beta, &A->bx, &A->by, &A->nl, A->fb, A->dxf);
}
void or_amr2_fine_residual(OrAmr2 *A)
{
    fine_bc(A, &A->phi, 0, A->dxf, A->nxf, A->nyf, A->fb);
    fine_nonlinear(A);
    or_vcnlcomputeres2d(&A->res, &A->phi, &A->rhs, A->alpha, &A->acoef, A->beta, &A->bx, &A->by, &A->nl, A->fb, A->dxf);
}

/* ---------------- [Chombo] QuadCFInterp::coarseFineInterp, refinement ratio 2 ----------------
 * For every fine ghost cell outside a side of the patch that is not on the domain boundary:
 *  (1) the coarse field is interpolated along the interface to the ghost cell's tangential position with a
 *      quadratic through the coarse cell containing it and its two tangential neighbours (centred first and
 *      second differences; one-sided second-order differences next to a non-periodic domain boundary);
 *  (2) a quadratic in the normal direction through that value (at the coarse cell centre, 1.5 fine cells beyond
 *      the first interior fine cell) and the two fine cells inside the patch gives the ghost value:
 *      ghost = 8/15 phistar + 2/3 near - 1/5 far.
 * Corner ghost cells are not filled (the 5-point operator never reads them). */
static double coarse_at(const OrAmr2 *A, const double *c, int ghosted, int i, int j)
{
    if (A->bc.periodic[0]) { if (i < 0) i += A->nxc; else if (i >= A->nxc) i -= A->nxc; }
    if (A->bc.periodic[1]) { if (j < 0) j += A->nyc; else if (j >= A->nyc) j -= A->nyc; }
    return ghosted ? c[(size_t)(j + 1) * (A->nxc + 2) + (i + 1)] : c[(size_t)j * A->nxc + i];
}
static void cf_interp(OrAmr2 *A, OrFab *f, int comp, const double *coarse, int ghosted)
{
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    for (int dir = 0; dir < 2; dir++) {
        int tdir = 1 - dir;
        int ndomf = dir == 0 ? A->nxf : A->nyf, nct = tdir == 0 ? A->nxc : A->nyc;
        for (int side = 0; side < 2; side++) {
            int vlo = dir == 0 ? A->fb.lo0 : A->fb.lo1, vhi = dir == 0 ? A->fb.hi0 : A->fb.hi1;
            int g = side == 0 ? vlo - 1 : vhi + 1, inward = side == 0 ? 1 : -1;
            if (g < 0 || g > ndomf - 1) continue;                    /* physical boundary: m_bc fills it */
            int tlo = tdir == 0 ? A->fb.lo0 : A->fb.lo1, thi = tdir == 0 ? A->fb.hi0 : A->fb.hi1;
            for (int t = tlo; t <= thi; t++) {
                int icn = g >> 1, ict = t >> 1;
                double xt = (t & 1) ? 0.25 : -0.25;
                int per = A->bc.periodic[tdir];
                int have_lo = per || ict - 1 >= 0, have_hi = per || ict + 1 <= nct - 1;
#define CV(o) (dir == 0 ? coarse_at(A, coarse, ghosted, icn, ict + (o)) : coarse_at(A, coarse, ghosted, ict + (o), icn))
                double c0 = CV(0), d1 = 0.0, d2 = 0.0;
                if (have_lo && have_hi) { double cm = CV(-1), cp = CV(1); d1 = 0.5 * (cp - cm); d2 = cp - 2.0 * c0 + cm; }
                else if (have_hi) { double cp = CV(1), cpp = CV(2); d1 = 0.5 * (-3.0 * c0 + 4.0 * cp - cpp); d2 = c0 - 2.0 * cp + cpp; }
                else if (have_lo) { double cm = CV(-1), cmm = CV(-2); d1 = 0.5 * (3.0 * c0 - 4.0 * cm + cmm); d2 = c0 - 2.0 * cm + cmm; }
#undef CV
                double phistar = c0 + xt * d1 + (0.5 * xt * xt) * d2;
                int ig = dir == 0 ? g : t, jg = dir == 0 ? t : g;
                int i1 = dir == 0 ? g + inward : t, j1 = dir == 0 ? t : g + inward;
                int i2 = dir == 0 ? g + 2 * inward : t, j2 = dir == 0 ? t : g + 2 * inward;
                AT(f, ig, jg, comp) = c_s * phistar + c_b * AT(f, i1, j1, comp) + c_a * AT(f, i2, j2, comp);
            }
        }
    }
    A->cf_interps++;
}
void or_amr2_cf_interp_phi(OrAmr2 *A)
{
    or_level_get(A->C, 0, OR_F_PHI, A->phic, 0);
    cf_interp(A, &A->phi, 0, A->phic, 0);
}

/* ---------------- [Chombo] FORT_AVERAGE: coarse = (sum of the 4 fine cells, i fastest) * 1/4 ---------------- */
static void average_to_coarse(const OrAmr2 *A, const OrFab *fine, double *coarse)
{
    for (int J = A->cj0; J <= A->cj1; J++)
        for (int I = A->ci0; I <= A->ci1; I++) {
            double s = 0.0;
            for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + AT(fine, 2 * I + ii, 2 * J + jj, 0);
            CC(coarse, I, J) = s * 0.25;
        }
}

/* ---------------- UpdateOperator on the fine level with a coarser level ---------------- */
/* coarse cell-centred gradient, ghosted: compGradientCC + exchange + ExtrapGhostCells (src/AmrHydro.cpp:1466-1480).
 * The coarse head enters with its ghost cells as mixBCValues leaves them (inhomogeneous). */
static void coarse_gradient(OrAmr2 *A)
{
    int nx = A->nxc, ny = A->nyc, P = nx + 2;
    double *h = (double *)calloc((size_t)P * (ny + 2), sizeof(double)), *m = (double *)calloc((size_t)P * (ny + 2), sizeof(double));
    or_level_bc(A->C, 0, OR_F_PHI, 0);
    or_level_get(A->C, 0, OR_F_PHI, h, 1);
    or_level_get(A->C, 0, OR_F_MASK, m, 1);
#define G(a, i, j) (a)[(size_t)((j) + 1) * P + ((i) + 1)]
    if (A->bc.periodic[0]) for (int j = 0; j < ny; j++) { G(h, -1, j) = G(h, nx - 1, j); G(h, nx, j) = G(h, 0, j); G(m, -1, j) = G(m, nx - 1, j); G(m, nx, j) = G(m, 0, j); }
    if (A->bc.periodic[1]) for (int i = 0; i < nx; i++) { G(h, i, -1) = G(h, i, ny - 1); G(h, i, ny) = G(h, i, 0); G(m, i, -1) = G(m, i, ny - 1); G(m, i, ny) = G(m, i, 0); }
    int hm = A->ph.use_mask_gradients;
    double f0 = 1.0 / A->dxc[0], f1 = 1.0 / A->dxc[1];
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double gW = f0 * (G(h, i, j) - G(h, i - 1, j)), gE = f0 * (G(h, i + 1, j) - G(h, i, j));
            double gS = f1 * (G(h, i, j) - G(h, i, j - 1)), gN = f1 * (G(h, i, j + 1) - G(h, i, j));
            if (hm) {
                int mc = G(m, i, j) < 1e-6;
                if (mc || G(m, i - 1, j) < 1e-6) gW = 0.0;
                if (mc || G(m, i + 1, j) < 1e-6) gE = 0.0;
                if (mc || G(m, i, j - 1) < 1e-6) gS = 0.0;
                if (mc || G(m, i, j + 1) < 1e-6) gN = 0.0;
            }
            G(A->gxc, i, j) = 0.5 * (gW + gE); G(A->gyc, i, j) = 0.5 * (gS + gN);
        }
    double *gg[2] = {A->gxc, A->gyc};
    for (int c = 0; c < 2; c++) {
        double *a = gg[c];
        if (A->bc.periodic[0]) for (int j = 0; j < ny; j++) { G(a, -1, j) = G(a, nx - 1, j); G(a, nx, j) = G(a, 0, j); }
        else for (int j = 0; j < ny; j++) { G(a, -1, j) = 2.0 * G(a, 0, j) - G(a, 1, j); G(a, nx, j) = 2.0 * G(a, nx - 1, j) - G(a, nx - 2, j); }
        if (A->bc.periodic[1]) for (int i = 0; i < nx; i++) { G(a, i, -1) = G(a, i, ny - 1); G(a, i, ny) = G(a, i, 0); }
        else for (int i = 0; i < nx; i++) { G(a, i, -1) = 2.0 * G(a, i, 0) - G(a, i, 1); G(a, i, ny) = 2.0 * G(a, i, ny - 1) - G(a, i, ny - 2); }
    }
#undef G
    free(h); free(m);
}
/* ExtrapGhostCells on the sides of the patch that lie on the domain boundary (util/ExtrapGhostCells.cpp:94-180) */
static void fine_extrap(const OrAmr2 *A, OrFab *f)
{
    for (int dir = 0; dir < 2; dir++) {
        if (A->bc.periodic[dir]) continue;
        int ndom = dir == 0 ? A->nxf : A->nyf;
        for (int hiLo = 0; hiLo < 2; hiLo++) {
            int g = hiLo == 0 ? -1 : ndom;
            OrBox s;
            if (dir == 0) { s.lo0 = s.hi0 = g; s.lo1 = f->lo1; s.hi1 = f->hi1; }
            else { s.lo1 = s.hi1 = g; s.lo0 = f->lo0; s.hi0 = f->hi0; }
            if (s.lo0 < f->lo0 || s.hi0 > f->hi0 || s.lo1 < f->lo1 || s.hi1 > f->hi1) continue;
            or_simpleextrapbc(f, s, dir, hiLo);
        }
    }
}
void or_amr2_fine_update_operator(OrAmr2 *A)
{
    /* UpdateOperator :47-53: exchange (one box: nothing) + physical BC; coarse-fine ghosts as they are */
    fine_bc(A, &A->phi, 0, A->dxf, A->nxf, A->nyf, A->fb);
    int hasMask = A->ph.use_mask_gradients;
    OrBox v = A->fb;
    memset(A->gradH.p, 0, sizeof(double) * 2 * (size_t)(A->gradH.hi0 - A->gradH.lo0 + 1) * (A->gradH.hi1 - A->gradH.lo1 + 1));
    for (int dir = 0; dir < 2; dir++) {
        OrBox eb = v; if (dir == 0) eb.hi0 += 1; else eb.hi1 += 1;
        OrFab eg = fab_alloc(eb, 0, 1);
        or_newmacgrad(&eg, &A->mask, &A->phi, eb, A->dxf, dir, hasMask);
        int ii = dir == 0, jj = dir == 1;
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++) AT(&A->gradH, i, j, dir) = 0.5 * (AT(&eg, i, j, 0) + AT(&eg, i + ii, j + jj, 0));
        free(eg.p);
    }
    coarse_gradient(A);                                   /* :1455-1480 */
    cf_interp(A, &A->gradH, 0, A->gxc, 1);                /* QuadCFInterp of the 2-component gradient :1482-1487 */
    cf_interp(A, &A->gradH, 1, A->gyc, 1);
    fine_extrap(A, &A->gradH);                            /* :1490-1491 */
    OrBox region = {A->Re.lo0, A->Re.lo1, A->Re.hi0, A->Re.hi1};
    or_computere(&A->B, &A->gradH, region, &A->Re, &A->ph);
    for (int dir = 0; dir < 2; dir++) {
        OrFab *bC = dir == 0 ? &A->bx : &A->by;
        OrBox fb = {bC->lo0, bC->lo1, bC->hi0, bC->hi1};
        OrFab B_ec = fab_alloc(fb, 0, 1), Re_ec = fab_alloc(fb, 0, 1), IM_ec = fab_alloc(fb, 0, 1);
        int ii = dir == 0, jj = dir == 1, face_hi = dir == 0 ? A->nxf : A->nyf;
        for (int j = fb.lo1; j <= fb.hi1; j++)
            for (int i = fb.lo0; i <= fb.hi0; i++) {
                AT(&Re_ec, i, j, 0) = 0.5 * (AT(&A->Re, i, j, 0) + AT(&A->Re, i - ii, j - jj, 0));
                AT(&B_ec, i, j, 0) = 0.5 * (AT(&A->B, i, j, 0) + AT(&A->B, i - ii, j - jj, 0));
                double m = AT(&A->mask, i, j, 0), mm1 = AT(&A->mask, i - ii, j - jj, 0), mec;
                if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; else mec = 0.0;
                int idx = dir == 0 ? i : j;
                if (idx == 0 || idx == face_hi) mec = 0.0;
                AT(&IM_ec, i, j, 0) = mec;
            }
        or_computebcoeff(&B_ec, &Re_ec, fb, bC, &IM_ec, &A->ph);
        free(B_ec.p); free(Re_ec.p); free(IM_ec.p);
    }
    A->lambda_dirty = 1;
    fine_reset_lambda(A);
}

/* ---------------- coarse composite operator: applyOpI + reflux ---------------- */
/* [Chombo] LevelFluxRegister: on every coarse-fine face the coarse flux is replaced by the average of the two
 * fine fluxes.  reg = -(dt*Fc) + (dt*Ff0)/2 + (dt*Ff1)/2 (dt = transverse coarse cell size), then
 * L(phi) of the coarse cell OUTSIDE the patch += sign * reg / (dx*dy), sign = +1 when the face is the cell's
 * high face.  Fluxes: getFlux (src/VCAMRNonLinearPoissonOp.cpp:792-841). */
static void reflux(OrAmr2 *A, const double *bxc, const double *byc, double *lofphi)
{
    const double rscale = 1.0 / (A->dxc[0] * A->dxc[1]);
    for (int dir = 0; dir < 2; dir++) {
        int ndomc = dir == 0 ? A->nxc : A->nyc;
        double tsize = A->dxc[1 - dir];
        double cs = A->beta * 1 / A->dxc[dir], fs = A->beta * 2 / A->dxc[dir];
        const OrFab *bf = dir == 0 ? &A->bx : &A->by;
        for (int side = 0; side < 2; side++) {
            int F = dir == 0 ? (side == 0 ? A->ci0 : A->ci1 + 1) : (side == 0 ? A->cj0 : A->cj1 + 1);   /* coarse face index */
            int outside = side == 0 ? F - 1 : F;
            if (outside < 0 || outside > ndomc - 1) continue;         /* patch side on the domain boundary */
            double sign = side == 0 ? 1.0 : -1.0;
            int tlo = dir == 0 ? A->cj0 : A->ci0, thi = dir == 0 ? A->cj1 : A->ci1;
            for (int T = tlo; T <= thi; T++) {
                double phihi, philo, bc_;
                if (dir == 0) { phihi = CC(A->phic, F, T); philo = CC(A->phic, F - 1, T); bc_ = bxc[(size_t)T * (A->nxc + 1) + F]; }
                else { phihi = CC(A->phic, T, F); philo = CC(A->phic, T, F - 1); bc_ = byc[(size_t)F * A->nxc + T]; }
                double Fc = -bc_ * ((phihi - philo) * cs);
                double reg = -(tsize * Fc);
                for (int k = 0; k < 2; k++) {
                    int fi = dir == 0 ? 2 * F : 2 * T + k, fj = dir == 0 ? 2 * T + k : 2 * F;
                    double ph_hi = AT(&A->phi, fi, fj, 0), ph_lo = dir == 0 ? AT(&A->phi, fi - 1, fj, 0) : AT(&A->phi, fi, fj - 1, 0);
                    double Ff = -AT(bf, fi, fj, 0) * ((ph_hi - ph_lo) * fs);
                    reg = reg + (tsize * Ff) * 0.5;
                }
                if (dir == 0) CC(lofphi, outside, T) = CC(lofphi, outside, T) + sign * rscale * reg;
                else CC(lofphi, T, outside) = CC(lofphi, T, outside) + sign * rscale * reg;
            }
        }
    }
    A->refluxes++;
}
/* AMROperator on the base level (:942-967): L0(phi0) by applyOpI, + reflux with the fine level (whose
 * coarse-fine ghosts are interpolated first, VCAMR...cpp:602).  Result in A->lphic. */
static void coarse_composite_operator(OrAmr2 *A)
{
    or_level_apply_op(A->C, 0, 0);
    or_level_get(A->C, 0, OR_F_LPHI, A->lphic, 0);
    or_level_get(A->C, 0, OR_F_PHI, A->phic, 0);
    double *bxc = (double *)malloc(sizeof(double) * (size_t)(A->nxc + 1) * A->nyc), *byc = (double *)malloc(sizeof(double) * (size_t)A->nxc * (A->nyc + 1));
    or_level_get(A->C, 0, OR_F_BX, bxc, 0); or_level_get(A->C, 0, OR_F_BY, byc, 0);
    cf_interp(A, &A->phi, 0, A->phic, 0);
    fine_bc(A, &A->phi, 0, A->dxf, A->nxf, A->nyf, A->fb);
    reflux(A, bxc, byc, A->lphic);
    free(bxc); free(byc);
}

/* composite residual: fine res = rhs1 - L1(phi1) (AMRResidualNF), coarse res = rhs0 - L0comp (AMRResidual);
 * returns the composite max norm (AMRNorm: coarse cells under the patch do not count, :1222-1264) */
double or_amr2_residual(OrAmr2 *A)
{
    or_amr2_cf_interp_phi(A);
    or_amr2_fine_residual(A);
    coarse_composite_operator(A);
    or_level_get(A->C, 0, OR_F_RHS, A->rhs0, 0);
    double nrm = 0.0;
    for (int J = 0; J < A->nyc; J++)
        for (int I = 0; I < A->nxc; I++) {
            double r = -1.0 * CC(A->lphic, I, J) + 1.0 * CC(A->rhs0, I, J);          /* axby(res, res, rhs, -1, 1) */
            CC(A->resc, I, J) = r;
            int covered = I >= A->ci0 && I <= A->ci1 && J >= A->cj0 && J <= A->cj1;
            if (!covered && fabs(r) > nrm) nrm = fabs(r);
        }
    for (int j = A->fb.lo1; j <= A->fb.hi1; j++)
        for (int i = A->fb.lo0; i <= A->fb.hi0; i++) { double r = fabs(AT(&A->res, i, j, 0)); if (r > nrm) nrm = r; }
    return nrm;
}

/* base-level part of the last composite residual (nxc x nyc, row-major; covered cells hold rhs - L too) */
const double *or_amr2_coarse_residual(const OrAmr2 *A) { return A->resc; }

/* one AMR FAS V-cycle (VCycleAMR; reconstruction) */
void or_amr2_vcycle(OrAmr2 *A, const OrSolverParams *sp)
{
    size_t nc = (size_t)A->nxc * A->nyc;
    or_level_get(A->C, 0, OR_F_RHS, A->rhs0, 0);
    /* operator of the fine level from the current head (bcoeff_otf) */
    or_amr2_cf_interp_phi(A);
    if (sp->bcoeff_otf) or_amr2_fine_update_operator(A);
    /* relaxNF(phi1, phi0, rhs1, pre): coarseFineInterp + levelGSRB x pre */
    or_amr2_fine_gsrb(A, sp->num_smooth);
    /* AMRRestrictS(skip_res = true): phi0 under the patch <- average(phi1) */
    or_level_get(A->C, 0, OR_F_PHI, A->phic, 0);
    average_to_coarse(A, &A->phi, A->phic);
    or_level_set(A->C, 0, OR_F_PHI, A->phic, 0);
    /* residuals: fine (AMRResidualNF), coarse composite (AMROperator + reflux); covered cells <- average(res1) */
    (void)or_amr2_residual(A);
    average_to_coarse(A, &A->res, A->resc);
    /* FAS right-hand side of the base level: res0' + L0(phi0)  (L0 = applyOpI without reflux: the level's own operator) */
    double *rhsp = (double *)malloc(sizeof(double) * nc);
    or_level_get(A->C, 0, OR_F_LPHI, A->lphic, 0);
    for (size_t k = 0; k < nc; k++) rhsp[k] = A->resc[k] + A->lphic[k];
    or_level_set(A->C, 0, OR_F_RHS, rhsp, 0);
    or_level_get(A->C, 0, OR_F_PHI, A->phiold, 0);
    or_level_vcycle(A->C, sp);                                   /* MGCycle of the base level */
    or_level_set(A->C, 0, OR_F_RHS, A->rhs0, 0);
    free(rhsp);
    /* AMRProlongS_2: phi1 += PROLONG_2_NL(phi0 - phi0_old), coarse correction with its BC ghosts (inhomogeneous in
     * FAS mode :1163-1165) and, inside the domain, the neighbouring coarse cells (copyTo + CornerCopier) */
    or_level_get(A->C, 0, OR_F_PHI, A->phic, 0);
    for (size_t k = 0; k < nc; k++) A->corr[k] = A->phic[k] - A->phiold[k];
    {
        OrBox cb = {A->ci0, A->cj0, A->ci1, A->cj1};
        OrFab ct = fab_alloc(cb, 1, 1);
        for (int J = ct.lo1; J <= ct.hi1; J++)
            for (int I = ct.lo0; I <= ct.hi0; I++) {
                int i = I, j = J;
                if (A->bc.periodic[0]) { if (i < 0) i += A->nxc; else if (i >= A->nxc) i -= A->nxc; }
                if (A->bc.periodic[1]) { if (j < 0) j += A->nyc; else if (j >= A->nyc) j -= A->nyc; }
                if (i >= 0 && i < A->nxc && j >= 0 && j < A->nyc) AT(&ct, I, J, 0) = CC(A->corr, i, j);
            }
        fine_bc(A, &ct, 0, A->dxc, A->nxc, A->nyc, cb);
        or_prolong_2_nl(&A->phi, &ct, A->fb, 2);
        free(ct.p);
    }
    /* relaxNF(phi1, phi0, rhs1, post) */
    or_amr2_cf_interp_phi(A);
    or_amr2_fine_gsrb(A, sp->num_smooth);
}

/* AMRMultiGrid::solveNoInit stopping rule on the composite residual */
int or_amr2_solve(OrAmr2 *A, const OrSolverParams *sp, double *hist)
{
    double initial_rnorm = or_amr2_residual(A);
    double rnorm = initial_rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    int goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    int goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        or_amr2_vcycle(A, sp);
        rnorm = or_amr2_residual(A);
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    return iter;
}
