/*
 * amrn.c -- TEST INFRASTRUCTURE ONLY.  AMR head solve on N nested levels: level 0 = the base level (an OrLevel
 * with its multigrid depths, level_shim.c), level l >= 1 = ONE rectangular patch refined by 2 and properly nested
 * (>= 2 cells of level l-1 between its boundary and the boundary of the level l-1 patch, AmrHydro.nestingRadius;
 * cfg4 / cfg5 of BASELINE.json are 3-level hierarchies).
 *
 * Same provenance as amr2.c (which is the two-level special case and must agree with this file bit for bit):
 * the operator methods follow src/AMRNonLinearPoissonOp.cpp:690-704, 889-1264 and
 * src/VCAMRNonLinearPoissonOp.cpp:34-64, 555-652, 792-841, src/AmrHydro.cpp:1415-1539; QuadCFInterp,
 * LevelFluxRegister, FORT_AVERAGE, the ghosted coarse copy of AMRProlongS_2 and the AMR FAS cycle order are
 * restated from upstream Chombo's documented semantics ([Chombo], unpinned).
 * Every patch is ONE box; data of a level is handed between levels as arrays over that level's whole DOMAIN
 * (zero outside the patch), which keeps the inter-level stencils identical for a base level and for a patch.
 */
#include "level_shim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(f, i, j, n) (*or_at((f), (i), (j), (n)))
#define MAXLEV 8

typedef struct Lv {
    int l, nxd, nyd;                  /* level index, DOMAIN size at this level */
    double dx[2];
    OrLevel *base;                    /* l == 0 */
    OrBox vb;                         /* valid box (l == 0: the domain) */
    OrFab phi, rhs, acoef, B, Pi, zb, mask, bx, by, lam, nl, dnl, res, lphi, gradH, Re;   /* l >= 1 */
    int lambda_dirty;
} Lv;

typedef struct OrAmr {
    int nlev;
    Lv lv[MAXLEV];
    OrBC bc; OrPhys ph; double alpha, beta;
} OrAmr;

static OrFab fab_alloc(OrBox b, int g, int ncomp)
{
    OrFab f;
    f.lo0 = b.lo0 - g; f.lo1 = b.lo1 - g; f.hi0 = b.hi0 + g; f.hi1 = b.hi1 + g; f.ncomp = ncomp;
    f.p = (double *)calloc((size_t)(f.hi0 - f.lo0 + 1) * (size_t)(f.hi1 - f.lo1 + 1) * (size_t)ncomp, sizeof(double));
    return f;
}

/* patches: nlev-1 boxes in the index space of the level BELOW each patch (coarse cells ci0,cj0,ci1,cj1) */
OrAmr *or_amr_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                     double alpha, double beta, int nlev, const int *patches)
{
    OrAmr *A = (OrAmr *)calloc(1, sizeof(OrAmr));
    A->nlev = nlev; A->bc = *bc; A->ph = *ph; A->alpha = alpha; A->beta = beta;
    Lv *L0 = &A->lv[0];
    L0->l = 0; L0->nxd = nx0; L0->nyd = ny0; L0->dx[0] = dx0; L0->dx[1] = dy0; L0->base = base;
    L0->vb.lo0 = 0; L0->vb.lo1 = 0; L0->vb.hi0 = nx0 - 1; L0->vb.hi1 = ny0 - 1;
    for (int l = 1; l < nlev; l++) {
        Lv *P = &A->lv[l], *C = &A->lv[l - 1];
        const int *q = patches + 4 * (l - 1);
        P->l = l; P->nxd = 2 * C->nxd; P->nyd = 2 * C->nyd; P->dx[0] = C->dx[0] / 2.0; P->dx[1] = C->dx[1] / 2.0;
        P->vb.lo0 = 2 * q[0]; P->vb.lo1 = 2 * q[1]; P->vb.hi0 = 2 * q[2] + 1; P->vb.hi1 = 2 * q[3] + 1;
        OrBox v = P->vb, fx = v, fy = v; fx.hi0 += 1; fy.hi1 += 1;
        P->phi = fab_alloc(v, 1, 1); P->rhs = fab_alloc(v, 0, 1); P->acoef = fab_alloc(v, 0, 1);
        P->B = fab_alloc(v, 1, 1); P->Pi = fab_alloc(v, 1, 1); P->zb = fab_alloc(v, 1, 1); P->mask = fab_alloc(v, 1, 1);
        P->bx = fab_alloc(fx, 0, 1); P->by = fab_alloc(fy, 0, 1);
        P->lam = fab_alloc(v, 0, 1); P->nl = fab_alloc(v, 0, 1); P->dnl = fab_alloc(v, 0, 1);
        P->res = fab_alloc(v, 0, 1); P->lphi = fab_alloc(v, 0, 1);
        P->gradH = fab_alloc(v, 1, 2); P->Re = fab_alloc(v, 1, 1);
        P->lambda_dirty = 1;
    }
    return A;
}
void or_amr_destroy(OrAmr *A)
{
    if (!A) return;
    for (int l = 1; l < A->nlev; l++) {
        Lv *P = &A->lv[l];
        OrFab *f[] = {&P->phi, &P->rhs, &P->acoef, &P->B, &P->Pi, &P->zb, &P->mask, &P->bx, &P->by, &P->lam, &P->nl,
                      &P->dnl, &P->res, &P->lphi, &P->gradH, &P->Re};
        for (int k = 0; k < 16; k++) free(f[k]->p);
    }
    free(A);
}
static OrFab *patch_field(Lv *P, int field)
{
    switch (field) {
    case OR_F_PHI: return &P->phi; case OR_F_RHS: return &P->rhs; case OR_F_ACOEF: return &P->acoef;
    case OR_F_B: return &P->B; case OR_F_PI: return &P->Pi; case OR_F_ZB: return &P->zb; case OR_F_MASK: return &P->mask;
    case OR_F_BX: return &P->bx; case OR_F_BY: return &P->by; case OR_F_LAMBDA: return &P->lam;
    case OR_F_RES: return &P->res; case OR_F_LPHI: return &P->lphi; case OR_F_NL: return &P->nl; case OR_F_DNL: return &P->dnl;
    }
    return NULL;
}
/* patch-sized arrays <-> fields of level l >= 1 (same convention as or_amr2_fine_io) */
void or_amr_patch_io(OrAmr *A, int l, int field, double *g, int ghosted, int set)
{
    Lv *P = &A->lv[l];
    OrFab *f = patch_field(P, field);
    int face = field == OR_F_BX || field == OR_F_BY;
    int gf = (!face && ghosted && f->lo0 < P->vb.lo0) ? 1 : 0;
    int lo0 = face ? f->lo0 : P->vb.lo0 - gf, hi0 = face ? f->hi0 : P->vb.hi0 + gf;
    int lo1 = face ? f->lo1 : P->vb.lo1 - gf, hi1 = face ? f->hi1 : P->vb.hi1 + gf;
    long pitch = hi0 - lo0 + 1;
    for (int j = lo1; j <= hi1; j++)
        for (int i = lo0; i <= hi0; i++) {
            double *q = &g[(long)(j - lo1) * pitch + (i - lo0)];
            if (set) AT(f, i, j, 0) = *q; else *q = AT(f, i, j, 0);
        }
    if (set && (field == OR_F_ACOEF || face)) P->lambda_dirty = 1;
}

/* ---------------- a level's field as an array over its DOMAIN ---------------- */
/* cells: nyd x nxd (or ghosted (nyd+2) x (nxd+2)); BX nyd x (nxd+1); BY (nyd+1) x nxd; zero outside a patch */
static double *dom_get(OrAmr *A, int l, int field, int ghosted)
{
    Lv *V = &A->lv[l];
    int gg = ghosted ? 1 : 0;
    size_t n = field == OR_F_BX ? (size_t)(V->nxd + 1) * V->nyd : field == OR_F_BY ? (size_t)V->nxd * (V->nyd + 1)
                                : (size_t)(V->nxd + 2 * gg) * (V->nyd + 2 * gg);
    double *a = (double *)calloc(n, sizeof(double));
    if (l == 0) { or_level_get(V->base, 0, field, a, ghosted); return a; }
    const OrFab *f = patch_field(V, field);
    long pitch = field == OR_F_BX ? V->nxd + 1 : field == OR_F_BY ? V->nxd : V->nxd + 2 * gg;
    int lo0 = f->lo0, hi0 = f->hi0, lo1 = f->lo1, hi1 = f->hi1;
    if (!(field == OR_F_BX || field == OR_F_BY) && !ghosted) { lo0 = V->vb.lo0; hi0 = V->vb.hi0; lo1 = V->vb.lo1; hi1 = V->vb.hi1; }
    for (int j = lo1; j <= hi1; j++)
        for (int i = lo0; i <= hi0; i++) a[(size_t)(j + gg) * pitch + (i + gg)] = AT(f, i, j, 0);
    return a;
}
/* write the cells of `region` (level-l indices) from a valid-cell domain array back into the level */
static void dom_put(OrAmr *A, int l, int field, const double *a, OrBox region)
{
    Lv *V = &A->lv[l];
    if (l == 0) {
        double *full = dom_get(A, 0, field, 0);
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) full[(size_t)j * V->nxd + i] = a[(size_t)j * V->nxd + i];
        or_level_set(V->base, 0, field, full, 0);
        free(full);
        return;
    }
    OrFab *f = patch_field(V, field);
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) AT(f, i, j, 0) = a[(size_t)j * V->nxd + i];
}

/* ---------------- patch operator methods (one box), as amr2.c ---------------- */
static void box_bc(const OrAmr *A, OrFab *state, int homogeneous, const double dx[2], int ndx, int ndy, OrBox valid)
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
static void patch_nonlinear(OrAmr *A, Lv *P)
{
    size_t n = (size_t)(P->vb.hi0 - P->vb.lo0 + 1) * (P->vb.hi1 - P->vb.lo1 + 1);
    if (!A->ph.use_NL) { memset(P->nl.p, 0, sizeof(double) * n); memset(P->dnl.p, 0, sizeof(double) * n); return; }
    or_computenonlinearterms(&P->phi, &P->B, &P->mask, &P->Pi, &P->zb, P->vb, &P->nl, &P->dnl, &A->ph);
}
static void patch_reset_lambda(OrAmr *A, Lv *P)
{
    if (!P->lambda_dirty) return;
    for (int j = P->vb.lo1; j <= P->vb.hi1; j++)
        for (int i = P->vb.lo0; i <= P->vb.hi0; i++) AT(&P->lam, i, j, 0) = AT(&P->acoef, i, j, 0) * A->alpha;
    for (int dir = 0; dir < 2; dir++)
        or_sumfacesnl(&P->lam, A->beta, dir == 0 ? &P->bx : &P->by, P->vb, dir, 1.0 / (P->dx[dir] * P->dx[dir]));
    P->lambda_dirty = 0;
}
static void lv_gsrb(OrAmr *A, int l, int sweeps)
{
    Lv *P = &A->lv[l];
    if (l == 0) { or_level_gsrb(P->base, 0, sweeps); return; }
    for (int it = 0; it < sweeps; it++) {
        patch_reset_lambda(A, P);
        for (int pass = 0; pass <= 1; pass++) {
            box_bc(A, &P->phi, 0, P->dx, P->nxd, P->nyd, P->vb);
            patch_nonlinear(A, P);
            or_gsrbhelmholtzvcnl2d(&P->phi, &P->rhs, P->vb, P->dx, A->alpha, &P->acoef, A->beta, &P->bx, &P->by,
                                   &P->nl, &P->dnl, &P->lam, pass);
        }
        box_bc(A, &P->phi, 1, P->dx, P->nxd, P->nyd, P->vb);
    }
}
static void lv_apply_op(OrAmr *A, int l)            /* applyOpI, inhomogeneous: LPHI of the level */
{
    Lv *P = &A->lv[l];
    if (l == 0) { or_level_apply_op(P->base, 0, 0); return; }
    box_bc(A, &P->phi, 0, P->dx, P->nxd, P->nyd, P->vb);
    patch_nonlinear(A, P);
    or_vcnlcomputeop2d(&P->lphi, &P->phi, A->alpha, &P->acoef, A->beta, &P->bx, &P->by, &P->nl, P->vb, P->dx);
}
static void lv_residual(OrAmr *A, int l)            /* residualI: RES of the level */
{
    Lv *P = &A->lv[l];
    if (l == 0) { or_level_residual(P->base, 0); return; }
    box_bc(A, &P->phi, 0, P->dx, P->nxd, P->nyd, P->vb);
    patch_nonlinear(A, P);
    or_vcnlcomputeres2d(&P->res, &P->phi, &P->rhs, A->alpha, &P->acoef, A->beta, &P->bx, &P->by, &P->nl, P->vb, P->dx);
}

/* ---------------- [Chombo] QuadCFInterp, ratio 2 (see amr2.c:cf_interp) ---------------- */
static double cdom(const OrAmr *A, const Lv *C, const double *c, int ghosted, int i, int j)
{
    if (A->bc.periodic[0]) { if (i < 0) i += C->nxd; else if (i >= C->nxd) i -= C->nxd; }
    if (A->bc.periodic[1]) { if (j < 0) j += C->nyd; else if (j >= C->nyd) j -= C->nyd; }
    return ghosted ? c[(size_t)(j + 1) * (C->nxd + 2) + (i + 1)] : c[(size_t)j * C->nxd + i];
}
static void cf_interp(OrAmr *A, int l, OrFab *f, int comp, const double *coarse, int ghosted)
{
    const Lv *F = &A->lv[l], *C = &A->lv[l - 1];
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    for (int dir = 0; dir < 2; dir++) {
        int tdir = 1 - dir;
        int ndomf = dir == 0 ? F->nxd : F->nyd, nct = tdir == 0 ? C->nxd : C->nyd;
        for (int side = 0; side < 2; side++) {
            int vlo = dir == 0 ? F->vb.lo0 : F->vb.lo1, vhi = dir == 0 ? F->vb.hi0 : F->vb.hi1;
            int g = side == 0 ? vlo - 1 : vhi + 1, inward = side == 0 ? 1 : -1;
            if (g < 0 || g > ndomf - 1) continue;
            int tlo = tdir == 0 ? F->vb.lo0 : F->vb.lo1, thi = tdir == 0 ? F->vb.hi0 : F->vb.hi1;
            for (int t = tlo; t <= thi; t++) {
                int icn = g >> 1, ict = t >> 1;
                double xt = (t & 1) ? 0.25 : -0.25;
                int per = A->bc.periodic[tdir];
                int have_lo = per || ict - 1 >= 0, have_hi = per || ict + 1 <= nct - 1;
#define CV(o) (dir == 0 ? cdom(A, C, coarse, ghosted, icn, ict + (o)) : cdom(A, C, coarse, ghosted, ict + (o), icn))
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
}
/* head of level l: coarse-fine ghosts from level l-1 (no-op on the base level) */
static void cf_interp_phi(OrAmr *A, int l)
{
    if (l == 0) return;
    double *c = dom_get(A, l - 1, OR_F_PHI, 0);
    cf_interp(A, l, &A->lv[l].phi, 0, c, 0);
    free(c);
}

/* [Chombo] FORT_AVERAGE of a field of level l into the covered cells of level l-1 */
static void average_down(OrAmr *A, int l, int field)
{
    Lv *F = &A->lv[l], *C = &A->lv[l - 1];
    const OrFab *fine = patch_field(F, field);
    double *c = (double *)calloc((size_t)C->nxd * C->nyd, sizeof(double));
    OrBox cov = {F->vb.lo0 / 2, F->vb.lo1 / 2, (F->vb.hi0 - 1) / 2, (F->vb.hi1 - 1) / 2};
    for (int J = cov.lo1; J <= cov.hi1; J++)
        for (int I = cov.lo0; I <= cov.hi0; I++) {
            double s = 0.0;
            for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + AT(fine, 2 * I + ii, 2 * J + jj, 0);
            c[(size_t)J * C->nxd + I] = s * 0.25;
        }
    dom_put(A, l - 1, field, c, cov);
    free(c);
}

/* ---------------- UpdateOperator of level l >= 1 with its coarser level ---------------- */
/* cell-centred gradient of level l over its valid box, ghosted domain arrays: compGradientCC + exchange +
 * ExtrapGhostCells on the domain sides the level touches (src/AmrHydro.cpp:1443-1451, 1466-1480) */
static void level_gradient(OrAmr *A, int l, double *gx, double *gy)
{
    Lv *V = &A->lv[l];
    int nx = V->nxd, ny = V->nyd, P = nx + 2;
    if (l == 0) or_level_bc(V->base, 0, OR_F_PHI, 0); else box_bc(A, &V->phi, 0, V->dx, V->nxd, V->nyd, V->vb);
    double *h = dom_get(A, l, OR_F_PHI, 1), *m = dom_get(A, l, OR_F_MASK, 1);
#define G(a, i, j) (a)[(size_t)((j) + 1) * P + ((i) + 1)]
    if (l == 0) {
        if (A->bc.periodic[0]) for (int j = 0; j < ny; j++) { G(h, -1, j) = G(h, nx - 1, j); G(h, nx, j) = G(h, 0, j); G(m, -1, j) = G(m, nx - 1, j); G(m, nx, j) = G(m, 0, j); }
        if (A->bc.periodic[1]) for (int i = 0; i < nx; i++) { G(h, i, -1) = G(h, i, ny - 1); G(h, i, ny) = G(h, i, 0); G(m, i, -1) = G(m, i, ny - 1); G(m, i, ny) = G(m, i, 0); }
    }
    int hm = A->ph.use_mask_gradients;
    double f0 = 1.0 / V->dx[0], f1 = 1.0 / V->dx[1];
    for (int j = V->vb.lo1; j <= V->vb.hi1; j++)
        for (int i = V->vb.lo0; i <= V->vb.hi0; i++) {
            double gW = f0 * (G(h, i, j) - G(h, i - 1, j)), gE = f0 * (G(h, i + 1, j) - G(h, i, j));
            double gS = f1 * (G(h, i, j) - G(h, i, j - 1)), gN = f1 * (G(h, i, j + 1) - G(h, i, j));
            if (hm) {
                int mc = G(m, i, j) < 1e-6;
                if (mc || G(m, i - 1, j) < 1e-6) gW = 0.0;
                if (mc || G(m, i + 1, j) < 1e-6) gE = 0.0;
                if (mc || G(m, i, j - 1) < 1e-6) gS = 0.0;
                if (mc || G(m, i, j + 1) < 1e-6) gN = 0.0;
            }
            G(gx, i, j) = 0.5 * (gW + gE); G(gy, i, j) = 0.5 * (gS + gN);
        }
    double *gg[2] = {gx, gy};
    for (int c = 0; c < 2; c++) {
        double *a = gg[c];
        if (l == 0 && A->bc.periodic[0]) for (int j = 0; j < ny; j++) { G(a, -1, j) = G(a, nx - 1, j); G(a, nx, j) = G(a, 0, j); }
        else if (!A->bc.periodic[0]) {
            if (V->vb.lo0 == 0) for (int j = V->vb.lo1; j <= V->vb.hi1; j++) G(a, -1, j) = 2.0 * G(a, 0, j) - G(a, 1, j);
            if (V->vb.hi0 == nx - 1) for (int j = V->vb.lo1; j <= V->vb.hi1; j++) G(a, nx, j) = 2.0 * G(a, nx - 1, j) - G(a, nx - 2, j);
        }
        if (l == 0 && A->bc.periodic[1]) for (int i = 0; i < nx; i++) { G(a, i, -1) = G(a, i, ny - 1); G(a, i, ny) = G(a, i, 0); }
        else if (!A->bc.periodic[1]) {
            if (V->vb.lo1 == 0) for (int i = V->vb.lo0; i <= V->vb.hi0; i++) G(a, i, -1) = 2.0 * G(a, i, 0) - G(a, i, 1);
            if (V->vb.hi1 == ny - 1) for (int i = V->vb.lo0; i <= V->vb.hi0; i++) G(a, i, ny) = 2.0 * G(a, i, ny - 1) - G(a, i, ny - 2);
        }
    }
#undef G
    free(h); free(m);
}
static void patch_extrap(const OrAmr *A, const Lv *P, OrFab *f)
{
    for (int dir = 0; dir < 2; dir++) {
        if (A->bc.periodic[dir]) continue;
        int ndom = dir == 0 ? P->nxd : P->nyd;
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
static void patch_update_operator(OrAmr *A, int l)
{
    Lv *P = &A->lv[l], *C = &A->lv[l - 1];
    cf_interp_phi(A, l - 1);                              /* the coarser level's own coarse-fine ghosts (its gradient reads them) */
    box_bc(A, &P->phi, 0, P->dx, P->nxd, P->nyd, P->vb);
    int hasMask = A->ph.use_mask_gradients;
    OrBox v = P->vb;
    memset(P->gradH.p, 0, sizeof(double) * 2 * (size_t)(P->gradH.hi0 - P->gradH.lo0 + 1) * (P->gradH.hi1 - P->gradH.lo1 + 1));
    for (int dir = 0; dir < 2; dir++) {
        OrBox eb = v; if (dir == 0) eb.hi0 += 1; else eb.hi1 += 1;
        OrFab eg = fab_alloc(eb, 0, 1);
        or_newmacgrad(&eg, &P->mask, &P->phi, eb, P->dx, dir, hasMask);
        int ii = dir == 0, jj = dir == 1;
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++) AT(&P->gradH, i, j, dir) = 0.5 * (AT(&eg, i, j, 0) + AT(&eg, i + ii, j + jj, 0));
        free(eg.p);
    }
    size_t ng = (size_t)(C->nxd + 2) * (C->nyd + 2);
    double *gxc = (double *)calloc(ng, sizeof(double)), *gyc = (double *)calloc(ng, sizeof(double));
    level_gradient(A, l - 1, gxc, gyc);
    cf_interp(A, l, &P->gradH, 0, gxc, 1);
    cf_interp(A, l, &P->gradH, 1, gyc, 1);
    free(gxc); free(gyc);
    patch_extrap(A, P, &P->gradH);
    OrBox region = {P->Re.lo0, P->Re.lo1, P->Re.hi0, P->Re.hi1};
    or_computere(&P->B, &P->gradH, region, &P->Re, &A->ph);
    for (int dir = 0; dir < 2; dir++) {
        OrFab *bC = dir == 0 ? &P->bx : &P->by;
        OrBox fb = {bC->lo0, bC->lo1, bC->hi0, bC->hi1};
        OrFab B_ec = fab_alloc(fb, 0, 1), Re_ec = fab_alloc(fb, 0, 1), IM_ec = fab_alloc(fb, 0, 1);
        int ii = dir == 0, jj = dir == 1, face_hi = dir == 0 ? P->nxd : P->nyd;
        for (int j = fb.lo1; j <= fb.hi1; j++)
            for (int i = fb.lo0; i <= fb.hi0; i++) {
                AT(&Re_ec, i, j, 0) = 0.5 * (AT(&P->Re, i, j, 0) + AT(&P->Re, i - ii, j - jj, 0));
                AT(&B_ec, i, j, 0) = 0.5 * (AT(&P->B, i, j, 0) + AT(&P->B, i - ii, j - jj, 0));
                double m = AT(&P->mask, i, j, 0), mm1 = AT(&P->mask, i - ii, j - jj, 0), mec;
                if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; else mec = 0.0;
                int idx = dir == 0 ? i : j;
                if (idx == 0 || idx == face_hi) mec = 0.0;
                AT(&IM_ec, i, j, 0) = mec;
            }
        or_computebcoeff(&B_ec, &Re_ec, fb, bC, &IM_ec, &A->ph);
        free(B_ec.p); free(Re_ec.p); free(IM_ec.p);
    }
    P->lambda_dirty = 1;
    patch_reset_lambda(A, P);
}

/* ---------------- [Chombo] LevelFluxRegister: reflux of level l's fluxes into L(phi) of level l-1 ---------------- */
static void reflux(OrAmr *A, int l, double *lofphi /* level l-1, domain array */)
{
    Lv *F = &A->lv[l], *C = &A->lv[l - 1];
    double *phic = dom_get(A, l - 1, OR_F_PHI, 0), *bxc = dom_get(A, l - 1, OR_F_BX, 0), *byc = dom_get(A, l - 1, OR_F_BY, 0);
    const int ci0 = F->vb.lo0 / 2, cj0 = F->vb.lo1 / 2, ci1 = (F->vb.hi0 - 1) / 2, cj1 = (F->vb.hi1 - 1) / 2;
    const double rscale = 1.0 / (C->dx[0] * C->dx[1]);
#define CC(a, i, j) (a)[(size_t)(j) * C->nxd + (i)]
    for (int dir = 0; dir < 2; dir++) {
        int ndomc = dir == 0 ? C->nxd : C->nyd;
        double tsize = C->dx[1 - dir];
        double cs = A->beta * 1 / C->dx[dir], fs = A->beta * 2 / C->dx[dir];
        const OrFab *bf = dir == 0 ? &F->bx : &F->by;
        for (int side = 0; side < 2; side++) {
            int Fc_ = dir == 0 ? (side == 0 ? ci0 : ci1 + 1) : (side == 0 ? cj0 : cj1 + 1);
            int outside = side == 0 ? Fc_ - 1 : Fc_;
            if (outside < 0 || outside > ndomc - 1) continue;
            double sign = side == 0 ? 1.0 : -1.0;
            int tlo = dir == 0 ? cj0 : ci0, thi = dir == 0 ? cj1 : ci1;
            for (int T = tlo; T <= thi; T++) {
                double phihi, philo, bc_;
                if (dir == 0) { phihi = CC(phic, Fc_, T); philo = CC(phic, Fc_ - 1, T); bc_ = bxc[(size_t)T * (C->nxd + 1) + Fc_]; }
                else { phihi = CC(phic, T, Fc_); philo = CC(phic, T, Fc_ - 1); bc_ = byc[(size_t)Fc_ * C->nxd + T]; }
                double Fcoarse = -bc_ * ((phihi - philo) * cs);
                double reg = -(tsize * Fcoarse);
                for (int k = 0; k < 2; k++) {
                    int fi = dir == 0 ? 2 * Fc_ : 2 * T + k, fj = dir == 0 ? 2 * T + k : 2 * Fc_;
                    double ph_hi = AT(&F->phi, fi, fj, 0), ph_lo = dir == 0 ? AT(&F->phi, fi - 1, fj, 0) : AT(&F->phi, fi, fj - 1, 0);
                    double Ff = -AT(bf, fi, fj, 0) * ((ph_hi - ph_lo) * fs);
                    reg = reg + (tsize * Ff) * 0.5;
                }
                if (dir == 0) CC(lofphi, outside, T) = CC(lofphi, outside, T) + sign * rscale * reg;
                else CC(lofphi, T, outside) = CC(lofphi, T, outside) + sign * rscale * reg;
            }
        }
    }
#undef CC
    free(phic); free(bxc); free(byc);
}

/* RES of level l-1 = rhs - [applyOpI(phi) + reflux from level l] (AMRResidual / AMROperator :889-967); the level's own
 * coarse-fine ghosts (from level l-2) and level l's ghosts are interpolated first.  LPHI keeps the plain L(phi). */
static void composite_residual(OrAmr *A, int l /* the FINE level of the pair */)
{
    Lv *C = &A->lv[l - 1];
    cf_interp_phi(A, l - 1);
    lv_apply_op(A, l - 1);
    double *lphi = dom_get(A, l - 1, OR_F_LPHI, 0), *rhs = dom_get(A, l - 1, OR_F_RHS, 0);
    cf_interp_phi(A, l);
    box_bc(A, &A->lv[l].phi, 0, A->lv[l].dx, A->lv[l].nxd, A->lv[l].nyd, A->lv[l].vb);
    reflux(A, l, lphi);
    size_t n = (size_t)C->nxd * C->nyd;
    for (size_t k = 0; k < n; k++) lphi[k] = -1.0 * lphi[k] + 1.0 * rhs[k];
    dom_put(A, l - 1, OR_F_RES, lphi, C->vb);
    free(lphi); free(rhs);
}

static double max_abs_excluding(OrAmr *A, int l, int has_finer)
{
    Lv *V = &A->lv[l];
    double *r = dom_get(A, l, OR_F_RES, 0), nrm = 0.0;
    OrBox cov = {1, 1, 0, 0};
    if (has_finer) { Lv *F = &A->lv[l + 1]; cov.lo0 = F->vb.lo0 / 2; cov.lo1 = F->vb.lo1 / 2; cov.hi0 = (F->vb.hi0 - 1) / 2; cov.hi1 = (F->vb.hi1 - 1) / 2; }
    for (int j = V->vb.lo1; j <= V->vb.hi1; j++)
        for (int i = V->vb.lo0; i <= V->vb.hi0; i++) {
            if (has_finer && i >= cov.lo0 && i <= cov.hi0 && j >= cov.lo1 && j <= cov.hi1) continue;
            double a = fabs(r[(size_t)j * V->nxd + i]);
            if (a > nrm) nrm = a;
        }
    free(r);
    return nrm;
}
/* composite residual of the hierarchy and its max norm (AMRNorm: covered cells do not count) */
double or_amr_residual(OrAmr *A)
{
    int top = A->nlev - 1;
    if (top == 0) { lv_residual(A, 0); return max_abs_excluding(A, 0, 0); }
    cf_interp_phi(A, top);
    lv_residual(A, top);                                   /* AMRResidualNF on the finest level */
    for (int l = top; l >= 1; l--) composite_residual(A, l);
    double nrm = 0.0;
    for (int l = 0; l <= top; l++) { double a = max_abs_excluding(A, l, l < top); if (a > nrm) nrm = a; }
    return nrm;
}

/* VCycleAMR(l): the rhs currently stored on level l is the one to relax against */
static void vcycle_amr(OrAmr *A, int l, const OrSolverParams *sp)
{
    if (l == 0) { or_level_vcycle(A->lv[0].base, sp); return; }
    Lv *F = &A->lv[l], *C = &A->lv[l - 1];
    size_t nc = (size_t)C->nxd * C->nyd;
    cf_interp_phi(A, l);
    if (sp->bcoeff_otf) patch_update_operator(A, l);
    lv_gsrb(A, l, sp->num_smooth);                          /* relaxNF */
    average_down(A, l, OR_F_PHI);                           /* AMRRestrictS(skip_res) */
    cf_interp_phi(A, l);
    lv_residual(A, l);                                      /* res_l = rhs_l - L_l(phi_l) (no reflux: a FAS rhs already holds the finer levels) */
    composite_residual(A, l);                               /* RES_{l-1} = rhs_{l-1} - [L + reflux], LPHI_{l-1} = L */
    average_down(A, l, OR_F_RES);                           /* covered cells <- average(res_l) */
    double *rhs_save = dom_get(A, l - 1, OR_F_RHS, 0), *res = dom_get(A, l - 1, OR_F_RES, 0), *lphi = dom_get(A, l - 1, OR_F_LPHI, 0);
    double *phiold = dom_get(A, l - 1, OR_F_PHI, 0);
    double *rhsp = (double *)malloc(sizeof(double) * nc);
    for (size_t k = 0; k < nc; k++) rhsp[k] = res[k] + lphi[k];
    dom_put(A, l - 1, OR_F_RHS, rhsp, C->vb);
    vcycle_amr(A, l - 1, sp);
    dom_put(A, l - 1, OR_F_RHS, rhs_save, C->vb);
    /* AMRProlongS_2 */
    double *phic = dom_get(A, l - 1, OR_F_PHI, 0);
    {
        OrBox cb = {F->vb.lo0 / 2, F->vb.lo1 / 2, (F->vb.hi0 - 1) / 2, (F->vb.hi1 - 1) / 2};
        OrFab ct = fab_alloc(cb, 1, 1);
        for (int J = ct.lo1; J <= ct.hi1; J++)
            for (int I = ct.lo0; I <= ct.hi0; I++) {
                int i = I, j = J;
                if (A->bc.periodic[0]) { if (i < 0) i += C->nxd; else if (i >= C->nxd) i -= C->nxd; }
                if (A->bc.periodic[1]) { if (j < 0) j += C->nyd; else if (j >= C->nyd) j -= C->nyd; }
                if (i >= 0 && i < C->nxd && j >= 0 && j < C->nyd) AT(&ct, I, J, 0) = phic[(size_t)j * C->nxd + i] - phiold[(size_t)j * C->nxd + i];
            }
        box_bc(A, &ct, 0, C->dx, C->nxd, C->nyd, cb);
        or_prolong_2_nl(&F->phi, &ct, F->vb, 2);
        free(ct.p);
    }
    free(rhs_save); free(res); free(lphi); free(phiold); free(rhsp); free(phic);
    cf_interp_phi(A, l);
    lv_gsrb(A, l, sp->num_smooth);
}
void or_amr_vcycle(OrAmr *A, const OrSolverParams *sp) { vcycle_amr(A, A->nlev - 1, sp); }

int or_amr_solve(OrAmr *A, const OrSolverParams *sp, double *hist)
{
    double initial_rnorm = or_amr_residual(A);
    double rnorm = initial_rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    int goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    int goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        or_amr_vcycle(A, sp);
        rnorm = or_amr_residual(A);
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    return iter;
}
