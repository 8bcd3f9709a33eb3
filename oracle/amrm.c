/*
 * amrm.c -- TEST INFRASTRUCTURE ONLY.  AMR head solve on N levels whose levels >= 1 are UNIONS OF BOXES, the way the
 * reference grids them (BRMeshRefine with fill_ratio < 1 and block_factor 2, src/AmrHydro.cpp:4176-4604, gives several
 * abutting and disjoint boxes per level; exec/AMR_multiMoulins/run_C_3lev/input.hydro:37,64-83).  Level 0 = the base
 * level (an OrLevel with its multigrid depths, level_shim.c); level l >= 1 = nbox[l] disjoint, coarse-aligned boxes
 * refined by 2 whose union is properly nested in level l-1 (coarsen(box) grown by 2 lies in the union of level l-1 or
 * outside the domain).
 *
 * Every box owns its fabs with ONE ghost layer, as a Chombo LevelData<FArrayBox> does.  A ghost cell of a box is
 *   - a fine-fine cell   when another box of the level (or a periodic image) holds it: filled by exchange()
 *                        (Copier::exchange, src/VCAMRNonLinearPoissonOp.cpp:47,124,304,405,692,751, built at :912-913),
 *   - a coarse-fine cell when it lies inside the domain and no box of the level holds it: filled by QuadCFInterp
 *                        (src/AMRNonLinearPoissonOp.cpp:698-700, 933, 956, 1003; src/VCAMRNonLinearPoissonOp.cpp:602),
 *   - a domain ghost     otherwise: mixBCValues (src/AmrHydro.cpp:248-309).
 * At a re-entrant corner of the union the same index is the x-ghost of one box and the y-ghost of another, with
 * different interpolated values: per-box storage keeps both, as the reference does.
 *
 * Same provenance as amrn.c, which is the one-box-per-level special case and must agree with this file BIT FOR BIT
 * (tests/test_oracle_amrm.py): operator methods follow src/AMRNonLinearPoissonOp.cpp:690-704, 889-1264,
 * src/VCAMRNonLinearPoissonOp.cpp:34-64, 555-652, 792-841, src/AmrHydro.cpp:1415-1539.  [Chombo] = restated from upstream
 * Chombo 3.2 semantics because the fork is not vendored (UNPINNED):
 *   QuadCFInterp / QuadCFStencil   tangential derivatives on the coarse level use only "good" coarse cells = inside the
 *                                  domain (or periodic images) and NOT covered by the fine level: centred when both
 *                                  neighbours are good, second-order one-sided when two good cells lie on one side,
 *                                  "dropped order" (first-order difference, no second derivative) when only one does,
 *                                  zero derivatives when none does; then the normal quadratic 8/15, 2/3, -1/5.
 *                                  With one rectangular box per level the covered test never fires.
 *   LevelFluxRegister              a coarse cell next to several coarse-fine faces takes their increments one after
 *                                  the other in the order (fine box, direction, side).
 *   FORT_AVERAGE, the ghosted coarse copy of AMRProlongS_2, the AMR FAS cycle order.
 */
#include "level_shim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(f, i, j, n) (*or_at((f), (i), (j), (n)))
#define MAXLEV 8
enum { ORM_F_GRADH = 100, ORM_F_RE = 101 };

typedef struct Bx {
    OrBox vb;
    OrFab phi, rhs, acoef, B, Pi, zb, mask, bx, by, lam, nl, dnl, res, lphi, gradH, Re;
    int lambda_dirty;
} Bx;
typedef struct LvM {
    int l, nxd, nyd;                  /* level index, DOMAIN size at this level */
    double dx[2];
    OrLevel *base;                    /* l == 0 */
    int nbox; Bx *b;                  /* l >= 1 */
    int *owner;                       /* l >= 1: nyd x nxd, index of the box holding the cell, -1 = none */
} LvM;
typedef struct OrAmrM {
    int nlev;
    LvM lv[MAXLEV];
    OrBC bc; OrPhys ph; double alpha, beta;
} OrAmrM;

static OrFab fab_alloc(OrBox b, int g, int ncomp)
{
    OrFab f;
    f.lo0 = b.lo0 - g; f.lo1 = b.lo1 - g; f.hi0 = b.hi0 + g; f.hi1 = b.hi1 + g; f.ncomp = ncomp;
    f.p = (double *)calloc((size_t)(f.hi0 - f.lo0 + 1) * (size_t)(f.hi1 - f.lo1 + 1) * (size_t)ncomp, sizeof(double));
    return f;
}
/* periodic wrap of a cell index of level V; returns 0 when the cell lies outside the domain */
static int wrap_cell(const OrAmrM *A, const LvM *V, int *i, int *j)
{
    if (A->bc.periodic[0]) { if (*i < 0) *i += V->nxd; else if (*i >= V->nxd) *i -= V->nxd; }
    if (A->bc.periodic[1]) { if (*j < 0) *j += V->nyd; else if (*j >= V->nyd) *j -= V->nyd; }
    return *i >= 0 && *i < V->nxd && *j >= 0 && *j < V->nyd;
}
/* box of level V (l >= 1) holding cell (i,j) after the periodic wrap, -1 = none / outside; level 0 holds everything */
static int owner_of(const OrAmrM *A, const LvM *V, int i, int j)
{
    if (!wrap_cell(A, V, &i, &j)) return -1;
    if (V->l == 0) return 0;
    return V->owner[(size_t)j * V->nxd + i];
}

/* nbox[l], l = 1 .. nlev-1 (nbox[0] is ignored); boxes: all levels' boxes one after the other, 4 ints each
 * (lo0, lo1, hi0, hi1) in the index space of THEIR OWN level.  Returns NULL if a box is misaligned, boxes overlap
 * or the nesting fails. */
OrAmrM *or_amrm_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                       double alpha, double beta, int nlev, const int *nbox, const int *boxes)
{
    OrAmrM *A = (OrAmrM *)calloc(1, sizeof(OrAmrM));
    A->nlev = nlev; A->bc = *bc; A->ph = *ph; A->alpha = alpha; A->beta = beta;
    LvM *L0 = &A->lv[0];
    L0->l = 0; L0->nxd = nx0; L0->nyd = ny0; L0->dx[0] = dx0; L0->dx[1] = dy0; L0->base = base;
    const int *q = boxes;
    for (int l = 1; l < nlev; l++) {
        LvM *P = &A->lv[l], *C = &A->lv[l - 1];
        P->l = l; P->nxd = 2 * C->nxd; P->nyd = 2 * C->nyd; P->dx[0] = C->dx[0] / 2.0; P->dx[1] = C->dx[1] / 2.0;
        P->nbox = nbox[l];
        P->b = (Bx *)calloc((size_t)P->nbox, sizeof(Bx));
        P->owner = (int *)malloc(sizeof(int) * (size_t)P->nxd * P->nyd);
        for (size_t k = 0; k < (size_t)P->nxd * P->nyd; k++) P->owner[k] = -1;
        int bad = 0;
        for (int k = 0; k < P->nbox; k++, q += 4) {
            Bx *B = &P->b[k];
            B->vb.lo0 = q[0]; B->vb.lo1 = q[1]; B->vb.hi0 = q[2]; B->vb.hi1 = q[3];
            if ((q[0] & 1) || (q[1] & 1) || !(q[2] & 1) || !(q[3] & 1) || q[0] < 0 || q[1] < 0 || q[2] >= P->nxd || q[3] >= P->nyd || q[2] < q[0] || q[3] < q[1]) { bad = 1; continue; }
            for (int j = q[1]; j <= q[3]; j++)
                for (int i = q[0]; i <= q[2]; i++) {
                    if (P->owner[(size_t)j * P->nxd + i] >= 0) bad = 1;
                    P->owner[(size_t)j * P->nxd + i] = k;
                }
            OrBox v = B->vb, fx = v, fy = v; fx.hi0 += 1; fy.hi1 += 1;
            B->phi = fab_alloc(v, 1, 1); B->rhs = fab_alloc(v, 0, 1); B->acoef = fab_alloc(v, 0, 1);
            B->B = fab_alloc(v, 1, 1); B->Pi = fab_alloc(v, 1, 1); B->zb = fab_alloc(v, 1, 1); B->mask = fab_alloc(v, 1, 1);
            B->bx = fab_alloc(fx, 0, 1); B->by = fab_alloc(fy, 0, 1);
            B->lam = fab_alloc(v, 0, 1); B->nl = fab_alloc(v, 0, 1); B->dnl = fab_alloc(v, 0, 1);
            B->res = fab_alloc(v, 0, 1); B->lphi = fab_alloc(v, 0, 1);
            B->gradH = fab_alloc(v, 1, 2); B->Re = fab_alloc(v, 1, 1);
            B->lambda_dirty = 1;
        }
        /* proper nesting: coarsen(box) grown by 2 lies in level l-1 (or outside a non-periodic domain) */
        for (int k = 0; k < P->nbox && !bad; k++) {
            const OrBox *v = &P->b[k].vb;
            for (int J = v->lo1 / 2 - 2; J <= v->hi1 / 2 + 2 && !bad; J++)
                for (int I = v->lo0 / 2 - 2; I <= v->hi0 / 2 + 2; I++) {
                    int i = I, j = J;
                    if (!wrap_cell(A, C, &i, &j)) continue;
                    if (l - 1 > 0 && C->owner[(size_t)j * C->nxd + i] < 0) { bad = 1; break; }
                }
        }
        if (bad) { A->nlev = l + 1; void or_amrm_destroy(OrAmrM *); or_amrm_destroy(A); return NULL; }
    }
    return A;
}
void or_amrm_destroy(OrAmrM *A)
{
    if (!A) return;
    for (int l = 1; l < A->nlev; l++) {
        LvM *P = &A->lv[l];
        for (int k = 0; k < P->nbox; k++) {
            Bx *B = &P->b[k];
            OrFab *f[] = {&B->phi, &B->rhs, &B->acoef, &B->B, &B->Pi, &B->zb, &B->mask, &B->bx, &B->by, &B->lam, &B->nl,
                          &B->dnl, &B->res, &B->lphi, &B->gradH, &B->Re};
            for (int m = 0; m < 16; m++) free(f[m]->p);
        }
        free(P->b); free(P->owner);
    }
    free(A);
}
int or_amrm_owner(const OrAmrM *A, int l, int i, int j) { return owner_of(A, &A->lv[l], i, j); }   /* with the periodic wrap */
int or_amrm_num_boxes(const OrAmrM *A, int l) { return l == 0 ? 1 : A->lv[l].nbox; }
void or_amrm_box(const OrAmrM *A, int l, int k, int *b4)
{
    const OrBox *v = &A->lv[l].b[k].vb;
    b4[0] = v->lo0; b4[1] = v->lo1; b4[2] = v->hi0; b4[3] = v->hi1;
}
static OrFab *bx_field(Bx *P, int field)
{
    switch (field) {
    case OR_F_PHI: return &P->phi; case OR_F_RHS: return &P->rhs; case OR_F_ACOEF: return &P->acoef;
    case OR_F_B: return &P->B; case OR_F_PI: return &P->Pi; case OR_F_ZB: return &P->zb; case OR_F_MASK: return &P->mask;
    case OR_F_BX: return &P->bx; case OR_F_BY: return &P->by; case OR_F_LAMBDA: return &P->lam;
    case OR_F_RES: return &P->res; case OR_F_LPHI: return &P->lphi; case OR_F_NL: return &P->nl; case OR_F_DNL: return &P->dnl;
    case ORM_F_GRADH: return &P->gradH; case ORM_F_RE: return &P->Re;
    }
    return NULL;
}
/* box-sized arrays <-> fields of box k of level l >= 1: cells ny x nx (ghosted: (ny+2) x (nx+2)), BX ny x (nx+1), BY (ny+1) x nx */
void or_amrm_box_io(OrAmrM *A, int l, int k, int field, double *g, int ghosted, int set)
{
    Bx *P = &A->lv[l].b[k];
    OrFab *f = bx_field(P, field);
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

/* ---------------- a level's field as an array over its DOMAIN (valid cells / faces; zero where the level has none) ---- */
static double *dom_get(OrAmrM *A, int l, int field)
{
    LvM *V = &A->lv[l];
    size_t n = field == OR_F_BX ? (size_t)(V->nxd + 1) * V->nyd : field == OR_F_BY ? (size_t)V->nxd * (V->nyd + 1) : (size_t)V->nxd * V->nyd;
    double *a = (double *)calloc(n, sizeof(double));
    if (l == 0) { or_level_get(V->base, 0, field, a, 0); return a; }
    long pitch = field == OR_F_BX ? V->nxd + 1 : V->nxd;
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        const OrFab *f = bx_field(P, field);
        int face = field == OR_F_BX || field == OR_F_BY;
        int lo0 = face ? f->lo0 : P->vb.lo0, hi0 = face ? f->hi0 : P->vb.hi0, lo1 = face ? f->lo1 : P->vb.lo1, hi1 = face ? f->hi1 : P->vb.hi1;
        for (int j = lo1; j <= hi1; j++)
            for (int i = lo0; i <= hi0; i++) a[(size_t)j * pitch + i] = AT(f, i, j, 0);
    }
    return a;
}
/* write the cells of `region` (level-l indices) that the level holds from a valid-cell domain array back into the level */
static void dom_put(OrAmrM *A, int l, int field, const double *a, OrBox region)
{
    LvM *V = &A->lv[l];
    if (l == 0) {
        double *full = dom_get(A, 0, field);
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) full[(size_t)j * V->nxd + i] = a[(size_t)j * V->nxd + i];
        or_level_set(V->base, 0, field, full, 0);
        free(full);
        return;
    }
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            int o = V->owner[(size_t)j * V->nxd + i];
            if (o >= 0) AT(bx_field(&V->b[o], field), i, j, 0) = a[(size_t)j * V->nxd + i];
        }
}
static OrBox whole_domain(const LvM *V) { OrBox b = {0, 0, V->nxd - 1, V->nyd - 1}; return b; }

/* ---------------- Copier::exchange of a 1-ghost cell field of level l >= 1 ---------------- */
/* every ghost cell of every box that another box of the level (or a periodic image) holds takes that box's value;
 * corners = 0: the operator's copier (exchangeDefine + trimEdges, :912-913: no corner cells) */
void or_amrm_exchange_fabs(OrAmrM *A, int l, OrFab **fabs, int corners)
{
    LvM *V = &A->lv[l];
    if (l == 0) return;
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        OrFab *f = fabs[k];
        for (int j = P->vb.lo1 - 1; j <= P->vb.hi1 + 1; j++)
            for (int i = P->vb.lo0 - 1; i <= P->vb.hi0 + 1; i++) {
                int gx = i < P->vb.lo0 || i > P->vb.hi0, gy = j < P->vb.lo1 || j > P->vb.hi1;
                if (!gx && !gy) continue;
                if (gx && gy && !corners) continue;
                int iw = i, jw = j;
                if (!wrap_cell(A, V, &iw, &jw)) continue;
                int o = V->owner[(size_t)jw * V->nxd + iw];
                if (o < 0) continue;
                const OrFab *s = fabs[o];
                for (int c = 0; c < f->ncomp; c++) AT(f, i, j, c) = AT(s, iw, jw, c);
            }
    }
}
static void exchange(OrAmrM *A, int l, int field, int corners)
{
    LvM *V = &A->lv[l];
    if (l == 0) return;
    OrFab **fabs = (OrFab **)malloc(sizeof(OrFab *) * (size_t)V->nbox);
    for (int k = 0; k < V->nbox; k++) fabs[k] = bx_field(&V->b[k], field);
    or_amrm_exchange_fabs(A, l, fabs, corners);
    free(fabs);
}

/* ---------------- box operator methods, as amrn.c ---------------- */
static void box_bc(const OrAmrM *A, OrFab *state, int homogeneous, const double dx[2], int ndx, int ndy, OrBox valid)
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
static void level_bc(OrAmrM *A, int l, int homogeneous)
{
    LvM *V = &A->lv[l];
    if (l == 0) { or_level_bc(V->base, 0, OR_F_PHI, homogeneous); return; }
    for (int k = 0; k < V->nbox; k++) box_bc(A, &V->b[k].phi, homogeneous, V->dx, V->nxd, V->nyd, V->b[k].vb);
}
static void bx_nonlinear(OrAmrM *A, Bx *P)
{
    size_t n = (size_t)(P->vb.hi0 - P->vb.lo0 + 1) * (P->vb.hi1 - P->vb.lo1 + 1);
    if (!A->ph.use_NL) { memset(P->nl.p, 0, sizeof(double) * n); memset(P->dnl.p, 0, sizeof(double) * n); return; }
    or_computenonlinearterms(&P->phi, &P->B, &P->mask, &P->Pi, &P->zb, P->vb, &P->nl, &P->dnl, &A->ph);
}
static void bx_reset_lambda(OrAmrM *A, const LvM *V, Bx *P)
{
    if (!P->lambda_dirty) return;
    for (int j = P->vb.lo1; j <= P->vb.hi1; j++)
        for (int i = P->vb.lo0; i <= P->vb.hi0; i++) AT(&P->lam, i, j, 0) = AT(&P->acoef, i, j, 0) * A->alpha;
    for (int dir = 0; dir < 2; dir++)
        or_sumfacesnl(&P->lam, A->beta, dir == 0 ? &P->bx : &P->by, P->vb, dir, 1.0 / (V->dx[dir] * V->dx[dir]));
    P->lambda_dirty = 0;
}
/* levelGSRB x sweeps (src/VCAMRNonLinearPoissonOp.cpp:654-760): per colour pass exchange, BC, NL, kernel on every box */
static void lv_gsrb(OrAmrM *A, int l, int sweeps)
{
    LvM *V = &A->lv[l];
    if (l == 0) { or_level_gsrb(V->base, 0, sweeps); return; }
    for (int it = 0; it < sweeps; it++) {
        for (int k = 0; k < V->nbox; k++) bx_reset_lambda(A, V, &V->b[k]);
        for (int pass = 0; pass <= 1; pass++) {
            exchange(A, l, OR_F_PHI, 0);
            level_bc(A, l, 0);
            for (int k = 0; k < V->nbox; k++) {
                Bx *P = &V->b[k];
                bx_nonlinear(A, P);
                or_gsrbhelmholtzvcnl2d(&P->phi, &P->rhs, P->vb, V->dx, A->alpha, &P->acoef, A->beta, &P->bx, &P->by,
                                       &P->nl, &P->dnl, &P->lam, pass);
            }
        }
        exchange(A, l, OR_F_PHI, 0);
        level_bc(A, l, 1);
    }
}
static void lv_apply_op(OrAmrM *A, int l)            /* applyOpI, inhomogeneous: LPHI of the level */
{
    LvM *V = &A->lv[l];
    if (l == 0) { or_level_apply_op(V->base, 0, 0); return; }
    level_bc(A, l, 0);
    exchange(A, l, OR_F_PHI, 0);
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        bx_nonlinear(A, P);
        or_vcnlcomputeop2d(&P->lphi, &P->phi, A->alpha, &P->acoef, A->beta, &P->bx, &P->by, &P->nl, P->vb, V->dx);
    }
}
static void lv_residual(OrAmrM *A, int l)            /* residualI: RES of the level */
{
    LvM *V = &A->lv[l];
    if (l == 0) { or_level_residual(V->base, 0); return; }
    level_bc(A, l, 0);
    exchange(A, l, OR_F_PHI, 0);
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        bx_nonlinear(A, P);
        or_vcnlcomputeres2d(&P->res, &P->phi, &P->rhs, A->alpha, &P->acoef, A->beta, &P->bx, &P->by, &P->nl, P->vb, V->dx);
    }
}

/* ---------------- [Chombo] QuadCFInterp, ratio 2 ---------------- */
/* coarse cell (I,J) of level C is "good" for the tangential stencils of level F = C + 1 */
static int good_cell(const OrAmrM *A, const LvM *C, const LvM *F, int I, int J)
{
    if (!wrap_cell(A, C, &I, &J)) return 0;
    return F->owner[(size_t)(2 * J) * F->nxd + 2 * I] < 0;
}
static double cdom(const OrAmrM *A, const LvM *C, const double *c, int i, int j)
{
    wrap_cell(A, C, &i, &j);
    return c[(size_t)j * C->nxd + i];
}
/* coarse-fine ghost cells (sides, no corners) of component comp of `field` of every box of level l <- coarse (domain
 * array of level l-1, valid cells) */
void or_amrm_cf_interp_fabs(OrAmrM *A, int l, OrFab **fabs, int comp, const double *coarse)
{
    const LvM *F = &A->lv[l], *C = &A->lv[l - 1];
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    for (int k = 0; k < F->nbox; k++) {
        Bx *P = &F->b[k];
        OrFab *f = fabs[k];
        for (int dir = 0; dir < 2; dir++) {
            int tdir = 1 - dir;
            int ndomf = dir == 0 ? F->nxd : F->nyd;
            for (int side = 0; side < 2; side++) {
                int vlo = dir == 0 ? P->vb.lo0 : P->vb.lo1, vhi = dir == 0 ? P->vb.hi0 : P->vb.hi1;
                int g = side == 0 ? vlo - 1 : vhi + 1, inward = side == 0 ? 1 : -1;
                if ((g < 0 || g > ndomf - 1) && !A->bc.periodic[dir]) continue;
                int tlo = tdir == 0 ? P->vb.lo0 : P->vb.lo1, thi = tdir == 0 ? P->vb.hi0 : P->vb.hi1;
                for (int t = tlo; t <= thi; t++) {
                    int ig = dir == 0 ? g : t, jg = dir == 0 ? t : g;
                    if (owner_of(A, F, ig, jg) >= 0) continue;               /* fine-fine cell: exchange() */
                    int icn = g >> 1, ict = t >> 1;
                    double xt = (t & 1) ? 0.25 : -0.25;
#define GOOD(o) (dir == 0 ? good_cell(A, C, F, icn, ict + (o)) : good_cell(A, C, F, ict + (o), icn))
#define CV(o) (dir == 0 ? cdom(A, C, coarse, icn, ict + (o)) : cdom(A, C, coarse, ict + (o), icn))
                    int have_lo = GOOD(-1), have_hi = GOOD(1);
                    double c0 = CV(0), d1 = 0.0, d2 = 0.0;
                    if (have_lo && have_hi) { double cm = CV(-1), cp = CV(1); d1 = 0.5 * (cp - cm); d2 = cp - 2.0 * c0 + cm; }
                    else if (have_hi) {
                        double cp = CV(1);
                        if (GOOD(2)) { double cpp = CV(2); d1 = 0.5 * (-3.0 * c0 + 4.0 * cp - cpp); d2 = c0 - 2.0 * cp + cpp; }
                        else d1 = cp - c0;
                    } else if (have_lo) {
                        double cm = CV(-1);
                        if (GOOD(-2)) { double cmm = CV(-2); d1 = 0.5 * (3.0 * c0 - 4.0 * cm + cmm); d2 = c0 - 2.0 * cm + cmm; }
                        else d1 = c0 - cm;
                    }
#undef CV
#undef GOOD
                    double phistar = c0 + xt * d1 + (0.5 * xt * xt) * d2;
                    int i1 = dir == 0 ? g + inward : t, j1 = dir == 0 ? t : g + inward;
                    int i2 = dir == 0 ? g + 2 * inward : t, j2 = dir == 0 ? t : g + 2 * inward;
                    AT(f, ig, jg, comp) = c_s * phistar + c_b * AT(f, i1, j1, comp) + c_a * AT(f, i2, j2, comp);
                }
            }
        }
    }
}
static void cf_interp(OrAmrM *A, int l, int field, int comp, const double *coarse)
{
    LvM *V = &A->lv[l];
    OrFab **fabs = (OrFab **)malloc(sizeof(OrFab *) * (size_t)V->nbox);
    for (int k = 0; k < V->nbox; k++) fabs[k] = bx_field(&V->b[k], field);
    or_amrm_cf_interp_fabs(A, l, fabs, comp, coarse);
    free(fabs);
}
/* head of level l: coarse-fine ghosts from level l-1 (no-op on the base level) */
static void cf_interp_phi(OrAmrM *A, int l)
{
    if (l == 0) return;
    double *c = dom_get(A, l - 1, OR_F_PHI);
    cf_interp(A, l, OR_F_PHI, 0, c);
    free(c);
}

/* [Chombo] FORT_AVERAGE of a field of level l into the covered cells of level l-1 */
static void average_down(OrAmrM *A, int l, int field)
{
    LvM *F = &A->lv[l], *C = &A->lv[l - 1];
    double *c = (double *)calloc((size_t)C->nxd * C->nyd, sizeof(double));
    for (int k = 0; k < F->nbox; k++) {
        Bx *P = &F->b[k];
        const OrFab *fine = bx_field(P, field);
        OrBox cov = {P->vb.lo0 / 2, P->vb.lo1 / 2, (P->vb.hi0 - 1) / 2, (P->vb.hi1 - 1) / 2};
        for (int J = cov.lo1; J <= cov.hi1; J++)
            for (int I = cov.lo0; I <= cov.hi0; I++) {
                double s = 0.0;
                for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + AT(fine, 2 * I + ii, 2 * J + jj, 0);
                c[(size_t)J * C->nxd + I] = s * 0.25;
            }
        dom_put(A, l - 1, field, c, cov);
    }
    free(c);
}

/* ---------------- UpdateOperator of level l >= 1 with its coarser level ---------------- */
/* cell-centred gradient of level l on its valid cells (compGradientCC, src/AmrHydro.cpp:1443-1451, 1466-1480), domain arrays */
static void level_gradient(OrAmrM *A, int l, double *gx, double *gy)
{
    LvM *V = &A->lv[l];
    int nx = V->nxd, ny = V->nyd;
    int hm = A->ph.use_mask_gradients;
    double f0 = 1.0 / V->dx[0], f1 = 1.0 / V->dx[1];
    level_bc(A, l, 0);
    if (l == 0) {
        int P = nx + 2;
        double *h = (double *)calloc((size_t)P * (ny + 2), sizeof(double)), *m = (double *)calloc((size_t)P * (ny + 2), sizeof(double));
        or_level_get(V->base, 0, OR_F_PHI, h, 1); or_level_get(V->base, 0, OR_F_MASK, m, 1);
#define G(a, i, j) (a)[(size_t)((j) + 1) * P + ((i) + 1)]
        if (A->bc.periodic[0]) for (int j = 0; j < ny; j++) { G(h, -1, j) = G(h, nx - 1, j); G(h, nx, j) = G(h, 0, j); G(m, -1, j) = G(m, nx - 1, j); G(m, nx, j) = G(m, 0, j); }
        if (A->bc.periodic[1]) for (int i = 0; i < nx; i++) { G(h, i, -1) = G(h, i, ny - 1); G(h, i, ny) = G(h, i, 0); G(m, i, -1) = G(m, i, ny - 1); G(m, i, ny) = G(m, i, 0); }
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
                gx[(size_t)j * nx + i] = 0.5 * (gW + gE); gy[(size_t)j * nx + i] = 0.5 * (gS + gN);
            }
#undef G
        free(h); free(m);
        return;
    }
    exchange(A, l, OR_F_PHI, 0);
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        for (int j = P->vb.lo1; j <= P->vb.hi1; j++)
            for (int i = P->vb.lo0; i <= P->vb.hi0; i++) {
                double gW = f0 * (AT(&P->phi, i, j, 0) - AT(&P->phi, i - 1, j, 0)), gE = f0 * (AT(&P->phi, i + 1, j, 0) - AT(&P->phi, i, j, 0));
                double gS = f1 * (AT(&P->phi, i, j, 0) - AT(&P->phi, i, j - 1, 0)), gN = f1 * (AT(&P->phi, i, j + 1, 0) - AT(&P->phi, i, j, 0));
                if (hm) {
                    int mc = AT(&P->mask, i, j, 0) < 1e-6;
                    if (mc || AT(&P->mask, i - 1, j, 0) < 1e-6) gW = 0.0;
                    if (mc || AT(&P->mask, i + 1, j, 0) < 1e-6) gE = 0.0;
                    if (mc || AT(&P->mask, i, j - 1, 0) < 1e-6) gS = 0.0;
                    if (mc || AT(&P->mask, i, j + 1, 0) < 1e-6) gN = 0.0;
                }
                gx[(size_t)j * nx + i] = 0.5 * (gW + gE); gy[(size_t)j * nx + i] = 0.5 * (gS + gN);
            }
    }
}
static void bx_extrap(const OrAmrM *A, const LvM *V, OrFab *f)
{
    for (int dir = 0; dir < 2; dir++) {
        if (A->bc.periodic[dir]) continue;
        int ndom = dir == 0 ? V->nxd : V->nyd;
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
static void level_update_operator(OrAmrM *A, int l)
{
    LvM *V = &A->lv[l], *C = &A->lv[l - 1];
    cf_interp_phi(A, l - 1);                              /* the coarser level's own coarse-fine ghosts (its gradient reads them) */
    exchange(A, l, OR_F_PHI, 0);                          /* UpdateOperator :47 */
    level_bc(A, l, 0);
    int hasMask = A->ph.use_mask_gradients;
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        OrBox v = P->vb;
        memset(P->gradH.p, 0, sizeof(double) * 2 * (size_t)(P->gradH.hi0 - P->gradH.lo0 + 1) * (P->gradH.hi1 - P->gradH.lo1 + 1));
        for (int dir = 0; dir < 2; dir++) {
            OrBox eb = v; if (dir == 0) eb.hi0 += 1; else eb.hi1 += 1;
            OrFab eg = fab_alloc(eb, 0, 1);
            or_newmacgrad(&eg, &P->mask, &P->phi, eb, V->dx, dir, hasMask);
            int ii = dir == 0, jj = dir == 1;
            for (int j = v.lo1; j <= v.hi1; j++)
                for (int i = v.lo0; i <= v.hi0; i++) AT(&P->gradH, i, j, dir) = 0.5 * (AT(&eg, i, j, 0) + AT(&eg, i + ii, j + jj, 0));
            free(eg.p);
        }
    }
    size_t ng = (size_t)C->nxd * C->nyd;
    double *gxc = (double *)calloc(ng, sizeof(double)), *gyc = (double *)calloc(ng, sizeof(double));
    level_gradient(A, l - 1, gxc, gyc);
    cf_interp(A, l, ORM_F_GRADH, 0, gxc);
    cf_interp(A, l, ORM_F_GRADH, 1, gyc);
    free(gxc); free(gyc);
    exchange(A, l, ORM_F_GRADH, 1);                       /* lvlgradH.exchange() src/AmrHydro.cpp:1490 */
    for (int k = 0; k < V->nbox; k++) {
        Bx *P = &V->b[k];
        bx_extrap(A, V, &P->gradH);
        OrBox region = {P->Re.lo0, P->Re.lo1, P->Re.hi0, P->Re.hi1};
        or_computere(&P->B, &P->gradH, region, &P->Re, &A->ph);
        for (int dir = 0; dir < 2; dir++) {
            OrFab *bC = dir == 0 ? &P->bx : &P->by;
            OrBox fb = {bC->lo0, bC->lo1, bC->hi0, bC->hi1};
            OrFab B_ec = fab_alloc(fb, 0, 1), Re_ec = fab_alloc(fb, 0, 1), IM_ec = fab_alloc(fb, 0, 1);
            int ii = dir == 0, jj = dir == 1, face_hi = dir == 0 ? V->nxd : V->nyd;
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
        bx_reset_lambda(A, V, P);
    }
}

/* ---------------- [Chombo] LevelFluxRegister: reflux of level l's fluxes into L(phi) of level l-1 ---------------- */
static void reflux(OrAmrM *A, int l, double *lofphi /* level l-1, domain array */)
{
    LvM *F = &A->lv[l], *C = &A->lv[l - 1];
    double *phic = dom_get(A, l - 1, OR_F_PHI), *bxc = dom_get(A, l - 1, OR_F_BX), *byc = dom_get(A, l - 1, OR_F_BY);
    const double rscale = 1.0 / (C->dx[0] * C->dx[1]);
    for (int k = 0; k < F->nbox; k++) {
        Bx *P = &F->b[k];
        const int ci0 = P->vb.lo0 / 2, cj0 = P->vb.lo1 / 2, ci1 = (P->vb.hi0 - 1) / 2, cj1 = (P->vb.hi1 - 1) / 2;
        for (int dir = 0; dir < 2; dir++) {
            int ndomc = dir == 0 ? C->nxd : C->nyd;
            double tsize = C->dx[1 - dir];
            double cs = A->beta * 1 / C->dx[dir], fs = A->beta * 2 / C->dx[dir];
            const OrFab *bf = dir == 0 ? &P->bx : &P->by;
            for (int side = 0; side < 2; side++) {
                int Fc_ = dir == 0 ? (side == 0 ? ci0 : ci1 + 1) : (side == 0 ? cj0 : cj1 + 1);
                int outside = side == 0 ? Fc_ - 1 : Fc_;
                if ((outside < 0 || outside > ndomc - 1) && !A->bc.periodic[dir]) continue;
                double sign = side == 0 ? 1.0 : -1.0;
                int tlo = dir == 0 ? cj0 : ci0, thi = dir == 0 ? cj1 : ci1;
                for (int T = tlo; T <= thi; T++) {
                    int oi = dir == 0 ? outside : T, oj = dir == 0 ? T : outside;          /* the coarse cell outside */
                    if (owner_of(A, F, 2 * oi, 2 * oj) >= 0) continue;                     /* fine-fine side: no register */
                    int ii = dir == 0 ? Fc_ : T, ij = dir == 0 ? T : Fc_;                  /* the coarse cell on the face's high side */
                    int li = dir == 0 ? Fc_ - 1 : T, lj = dir == 0 ? T : Fc_ - 1;
                    double phihi = cdom(A, C, phic, ii, ij), philo = cdom(A, C, phic, li, lj), bc_;
                    {   /* coarse face Fc_ (periodic image: face 0 == face ndomc) */
                        int fi = ii, fj = ij;
                        if (dir == 0) { if (fi < 0) fi += C->nxd; if (fi > C->nxd) fi -= C->nxd; bc_ = bxc[(size_t)fj * (C->nxd + 1) + fi]; }
                        else { if (fj < 0) fj += C->nyd; if (fj > C->nyd) fj -= C->nyd; bc_ = byc[(size_t)fj * C->nxd + fi]; }
                    }
                    double Fcoarse = -bc_ * ((phihi - philo) * cs);
                    double reg = -(tsize * Fcoarse);
                    for (int m = 0; m < 2; m++) {
                        int fi = dir == 0 ? 2 * Fc_ : 2 * T + m, fj = dir == 0 ? 2 * T + m : 2 * Fc_;
                        double ph_hi = AT(&P->phi, fi, fj, 0), ph_lo = dir == 0 ? AT(&P->phi, fi - 1, fj, 0) : AT(&P->phi, fi, fj - 1, 0);
                        double Ff = -AT(bf, fi, fj, 0) * ((ph_hi - ph_lo) * fs);
                        reg = reg + (tsize * Ff) * 0.5;
                    }
                    wrap_cell(A, C, &oi, &oj);
                    lofphi[(size_t)oj * C->nxd + oi] = lofphi[(size_t)oj * C->nxd + oi] + sign * rscale * reg;
                }
            }
        }
    }
    free(phic); free(bxc); free(byc);
}

/* RES of level l-1 = rhs - [applyOpI(phi) + reflux from level l] (AMRResidual / AMROperator :889-967); the level's own
 * coarse-fine ghosts (from level l-2) and level l's ghosts are interpolated first.  LPHI keeps the plain L(phi). */
static void composite_residual(OrAmrM *A, int l /* the FINE level of the pair */)
{
    LvM *C = &A->lv[l - 1];
    cf_interp_phi(A, l - 1);
    lv_apply_op(A, l - 1);
    double *lphi = dom_get(A, l - 1, OR_F_LPHI), *rhs = dom_get(A, l - 1, OR_F_RHS);
    cf_interp_phi(A, l);
    level_bc(A, l, 0);
    reflux(A, l, lphi);
    size_t n = (size_t)C->nxd * C->nyd;
    for (size_t k = 0; k < n; k++) lphi[k] = -1.0 * lphi[k] + 1.0 * rhs[k];
    dom_put(A, l - 1, OR_F_RES, lphi, whole_domain(C));
    free(lphi); free(rhs);
}

static double max_abs_excluding(OrAmrM *A, int l, int has_finer)
{
    LvM *V = &A->lv[l];
    double *r = dom_get(A, l, OR_F_RES), nrm = 0.0;
    for (int j = 0; j < V->nyd; j++)
        for (int i = 0; i < V->nxd; i++) {
            if (l > 0 && V->owner[(size_t)j * V->nxd + i] < 0) continue;
            if (has_finer && A->lv[l + 1].owner[(size_t)(2 * j) * A->lv[l + 1].nxd + 2 * i] >= 0) continue;
            double a = fabs(r[(size_t)j * V->nxd + i]);
            if (a > nrm) nrm = a;
        }
    free(r);
    return nrm;
}
/* composite residual of the hierarchy and its max norm (AMRNorm: covered cells do not count) */
double or_amrm_residual(OrAmrM *A)
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
static void vcycle_amr(OrAmrM *A, int l, const OrSolverParams *sp)
{
    if (l == 0) { or_level_vcycle(A->lv[0].base, sp); return; }
    LvM *F = &A->lv[l], *C = &A->lv[l - 1];
    size_t nc = (size_t)C->nxd * C->nyd;
    cf_interp_phi(A, l);
    if (sp->bcoeff_otf) level_update_operator(A, l);
    lv_gsrb(A, l, sp->num_smooth);                          /* relaxNF */
    average_down(A, l, OR_F_PHI);                           /* AMRRestrictS(skip_res) */
    cf_interp_phi(A, l);
    lv_residual(A, l);                                      /* res_l = rhs_l - L_l(phi_l) (no reflux: a FAS rhs already holds the finer levels) */
    composite_residual(A, l);                               /* RES_{l-1} = rhs_{l-1} - [L + reflux], LPHI_{l-1} = L */
    average_down(A, l, OR_F_RES);                           /* covered cells <- average(res_l) */
    double *rhs_save = dom_get(A, l - 1, OR_F_RHS), *res = dom_get(A, l - 1, OR_F_RES), *lphi = dom_get(A, l - 1, OR_F_LPHI);
    double *phiold = dom_get(A, l - 1, OR_F_PHI);
    double *rhsp = (double *)malloc(sizeof(double) * nc);
    for (size_t k = 0; k < nc; k++) rhsp[k] = res[k] + lphi[k];
    dom_put(A, l - 1, OR_F_RHS, rhsp, whole_domain(C));
    vcycle_amr(A, l - 1, sp);
    dom_put(A, l - 1, OR_F_RHS, rhs_save, whole_domain(C));
    /* AMRProlongS_2: per box the coarse correction over coarsen(box) grown by one cell, BC on the domain sides */
    double *phic = dom_get(A, l - 1, OR_F_PHI);
    for (int k = 0; k < F->nbox; k++) {
        Bx *P = &F->b[k];
        OrBox cb = {P->vb.lo0 / 2, P->vb.lo1 / 2, (P->vb.hi0 - 1) / 2, (P->vb.hi1 - 1) / 2};
        OrFab ct = fab_alloc(cb, 1, 1);
        for (int J = ct.lo1; J <= ct.hi1; J++)
            for (int I = ct.lo0; I <= ct.hi0; I++) {
                int i = I, j = J;
                if (wrap_cell(A, C, &i, &j)) AT(&ct, I, J, 0) = phic[(size_t)j * C->nxd + i] - phiold[(size_t)j * C->nxd + i];
            }
        box_bc(A, &ct, 0, C->dx, C->nxd, C->nyd, cb);
        or_prolong_2_nl(&P->phi, &ct, P->vb, 2);
        free(ct.p);
    }
    free(rhs_save); free(res); free(lphi); free(phiold); free(rhsp); free(phic);
    cf_interp_phi(A, l);
    lv_gsrb(A, l, sp->num_smooth);
}
void or_amrm_vcycle(OrAmrM *A, const OrSolverParams *sp) { vcycle_amr(A, A->nlev - 1, sp); }

int or_amrm_solve(OrAmrM *A, const OrSolverParams *sp, double *hist)
{
    double initial_rnorm = or_amrm_residual(A);
    double rnorm = initial_rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    int goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    int goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        or_amrm_vcycle(A, sp);
        rnorm = or_amrm_residual(A);
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    return iter;
}

/* pieces, for kernel-level parity of the device library */
void or_amrm_cf_interp_phi(OrAmrM *A, int l) { cf_interp_phi(A, l); }
void or_amrm_exchange(OrAmrM *A, int l, int field, int corners) { exchange(A, l, field, corners); }
void or_amrm_gsrb(OrAmrM *A, int l, int sweeps) { lv_gsrb(A, l, sweeps); }
void or_amrm_level_residual(OrAmrM *A, int l) { lv_residual(A, l); }
void or_amrm_update_operator(OrAmrM *A, int l) { level_update_operator(A, l); }
void or_amrm_average_down(OrAmrM *A, int l, int field) { average_down(A, l, field); }
