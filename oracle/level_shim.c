/*
 * level_shim.c -- TEST INFRASTRUCTURE ONLY.  See level_shim.h (parity status: suhmo_oracle.h).
 *
 * Restates, on a minimal box/LevelData stand-in, the C++ orchestration of the
 * reference's head solve.  Citations are relative to the SUHMO checkout.  Pieces
 * whose arithmetic lives in the un-vendored Chombo fork (exchange, DiriBC/NeumBC,
 * CellToEdge/EdgeToCell, CoarseAverage(Face), the FAS cycle itself) are restated
 * from upstream Chombo 3.2 semantics and are marked [Chombo].
 */
#include "level_shim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define AT(f, i, j, n) (*or_at((f), (i), (j), (n)))
#define MAXDEPTH 16

enum { W_PHIOLD = OR_F_DNL + 1, W_CORR, NCELLF };

typedef struct Depth {
    int nx, ny;
    double dx[2];
    int bsx, bsy, nbx, nby, nbox;
    OrBox *valid;
    OrFab *cell[NCELLF]; /* per-box fabs */
    int    ghost[NCELLF];
    OrFab *face[2];      /* bCoef: per-box x-face and y-face fabs (0 ghost) */
    /* WFlx_level work (depth 0 only, lazily allocated) */
    OrFab *gradH, *Re;
} Depth;

struct OrLevel {
    int ndepth;
    Depth d[MAXDEPTH];
    OrBC bc;
    OrPhys ph;
    double alpha, beta;
    int nthreads;
    int lambda_dirty[MAXDEPTH];
};

static const int k_ghost_of_field[NCELLF] = {
    /* PHI */ 1, /* RHS */ 0, /* ACOEF */ 0, /* B */ 1, /* PI */ 1, /* ZB */ 1, /* MASK */ 1,
    /* BX */ 0, /* BY */ 0, /* LAMBDA */ 0, /* RES */ 0, /* LPHI */ 0, /* NL */ 0, /* DNL */ 0,
    /* PHIOLD */ 1, /* CORR */ 1};

static OrFab fab_alloc(OrBox b, int g, int ncomp)
{
    OrFab f;
    f.lo0 = b.lo0 - g; f.lo1 = b.lo1 - g; f.hi0 = b.hi0 + g; f.hi1 = b.hi1 + g;
    f.ncomp = ncomp;
    size_t n = (size_t)(f.hi0 - f.lo0 + 1) * (size_t)(f.hi1 - f.lo1 + 1) * (size_t)ncomp;
    f.p = (double *)calloc(n, sizeof(double));
    return f;
}
static void fab_setval(OrFab *f, double v)
{
    size_t n = (size_t)(f->hi0 - f->lo0 + 1) * (size_t)(f->hi1 - f->lo1 + 1) * (size_t)f->ncomp;
    if (v == 0.0) memset(f->p, 0, n * sizeof(double));
    else for (size_t k = 0; k < n; k++) f->p[k] = v;
}
static void fab_copy(OrFab *dst, const OrFab *src)
{ /* same box assumed */
    size_t n = (size_t)(dst->hi0 - dst->lo0 + 1) * (size_t)(dst->hi1 - dst->lo1 + 1) * (size_t)dst->ncomp;
    memcpy(dst->p, src->p, n * sizeof(double));
}

static void depth_init(Depth *D, int nx, int ny, double dx, double dy, int bsx, int bsy)
{
    memset(D, 0, sizeof(*D));
    D->nx = nx; D->ny = ny; D->dx[0] = dx; D->dx[1] = dy;
    D->bsx = bsx; D->bsy = bsy;
    D->nbx = (nx + bsx - 1) / bsx; D->nby = (ny + bsy - 1) / bsy;
    D->nbox = D->nbx * D->nby;
    D->valid = (OrBox *)malloc(sizeof(OrBox) * D->nbox);
    for (int bj = 0; bj < D->nby; bj++)
        for (int bi = 0; bi < D->nbx; bi++) {
            OrBox b;
            b.lo0 = bi * bsx; b.hi0 = (bi + 1) * bsx - 1; if (b.hi0 > nx - 1) b.hi0 = nx - 1;
            b.lo1 = bj * bsy; b.hi1 = (bj + 1) * bsy - 1; if (b.hi1 > ny - 1) b.hi1 = ny - 1;
            D->valid[bj * D->nbx + bi] = b;
        }
    for (int f = 0; f < NCELLF; f++) {
        if (f == OR_F_BX || f == OR_F_BY) continue;
        D->ghost[f] = k_ghost_of_field[f];
        D->cell[f] = (OrFab *)malloc(sizeof(OrFab) * D->nbox);
        for (int b = 0; b < D->nbox; b++) D->cell[f][b] = fab_alloc(D->valid[b], D->ghost[f], 1);
    }
    for (int dir = 0; dir < 2; dir++) {
        D->face[dir] = (OrFab *)malloc(sizeof(OrFab) * D->nbox);
        for (int b = 0; b < D->nbox; b++) {
            OrBox fb = D->valid[b];
            if (dir == 0) fb.hi0 += 1; else fb.hi1 += 1; /* surroundingNodes(dir) */
            D->face[dir][b] = fab_alloc(fb, 0, 1);
        }
    }
}

static void depth_free(Depth *D)
{
    for (int f = 0; f < NCELLF; f++)
        if (D->cell[f]) { for (int b = 0; b < D->nbox; b++) free(D->cell[f][b].p); free(D->cell[f]); }
    for (int dir = 0; dir < 2; dir++)
        if (D->face[dir]) { for (int b = 0; b < D->nbox; b++) free(D->face[dir][b].p); free(D->face[dir]); }
    if (D->gradH) { for (int b = 0; b < D->nbox; b++) free(D->gradH[b].p); free(D->gradH); }
    if (D->Re) { for (int b = 0; b < D->nbox; b++) free(D->Re[b].p); free(D->Re); }
    free(D->valid);
}

/* boxes.coarsenable(r): every box refine(coarsen(b,r),r) == b  [Chombo] */
static int coarsenable(const Depth *D0, int r)
{
    for (int b = 0; b < D0->nbox; b++) {
        OrBox v = D0->valid[b];
        if (v.lo0 % r || v.lo1 % r || (v.hi0 + 1) % r || (v.hi1 + 1) % r) return 0;
    }
    return 1;
}

OrLevel *or_level_create(int nx, int ny, double dx, double dy, int max_box,
                         const OrBC *bc, const OrPhys *phys, double alpha, double beta,
                         int nthreads)
{
    OrLevel *L = (OrLevel *)calloc(1, sizeof(OrLevel));
    L->bc = *bc; L->ph = *phys; L->alpha = alpha; L->beta = beta;
    L->nthreads = nthreads > 0 ? nthreads : 1;
    depth_init(&L->d[0], nx, ny, dx, dy, max_box, max_box);
    L->ndepth = 1;
    /* MGnewOp depth rule: src/VCAMRNonLinearPoissonOp.cpp:1044-1060, s_maxCoarse = 2
     * (src/AMRNonLinearPoissonOp.cpp:32): depth d exists iff boxes coarsenable(2^d * 2);
     * layout = coarsen_dbl(boxes, 2^d): same box count, every box coarsened. */
    for (int dep = 1; dep < MAXDEPTH; dep++) {
        int c = 1 << dep;
        if (!coarsenable(&L->d[0], c * 2)) break;
        depth_init(&L->d[dep], nx / c, ny / c, dx * c, dy * c, max_box / c, max_box / c);
        L->ndepth = dep + 1;
    }
    for (int dep = 0; dep < L->ndepth; dep++) L->lambda_dirty[dep] = 1;
    return L;
}

void or_level_destroy(OrLevel *L)
{
    if (!L) return;
    for (int dep = 0; dep < L->ndepth; dep++) depth_free(&L->d[dep]);
    free(L);
}
/* solver.cut_solve_outside_domain of the coming solves (src/AmrHydro.cpp:872; the SHMIP post-processing inputs leave it at its
 * default 0 for their two steps, exec/E_SHMIP/E1/input.hydro_pp) */
void or_level_set_cutoffb(OrLevel *L, int v) { L->ph.cutOffB = v; }
int or_level_num_depths(const OrLevel *L) { return L->ndepth; }
int or_level_num_boxes(const OrLevel *L) { return L->d[0].nbox; }

/* ---------------- global <-> per-box copies ---------------- */
void or_level_set(OrLevel *L, int depth, int field, const double *g, int ghosted)
{
    Depth *D = &L->d[depth];
    if (field == OR_F_BX || field == OR_F_BY) {
        int dir = (field == OR_F_BY);
        long pitch = dir == 0 ? D->nx + 1 : D->nx;
        for (int b = 0; b < D->nbox; b++) {
            OrFab *f = &D->face[dir][b];
            for (int j = f->lo1; j <= f->hi1; j++)
                for (int i = f->lo0; i <= f->hi0; i++) AT(f, i, j, 0) = g[(long)j * pitch + i];
        }
        L->lambda_dirty[depth] = 1;
        return;
    }
    int gg = ghosted ? 1 : 0;
    long pitch = D->nx + 2 * gg;
    for (int b = 0; b < D->nbox; b++) {
        OrFab *f = &D->cell[field][b];
        OrBox v = D->valid[b];
        int gf = ghosted ? D->ghost[field] : 0;
        for (int j = v.lo1 - gf; j <= v.hi1 + gf; j++)
            for (int i = v.lo0 - gf; i <= v.hi0 + gf; i++)
                AT(f, i, j, 0) = g[(long)(j + gg) * pitch + (i + gg)];
    }
    if (field == OR_F_ACOEF) L->lambda_dirty[depth] = 1;
}

void or_level_get(const OrLevel *L, int depth, int field, double *g, int ghosted)
{
    const Depth *D = &L->d[depth];
    if (field == OR_F_BX || field == OR_F_BY) {
        int dir = (field == OR_F_BY);
        long pitch = dir == 0 ? D->nx + 1 : D->nx;
        for (int b = 0; b < D->nbox; b++) {
            const OrFab *f = &D->face[dir][b];
            for (int j = f->lo1; j <= f->hi1; j++)
                for (int i = f->lo0; i <= f->hi0; i++) g[(long)j * pitch + i] = AT(f, i, j, 0);
        }
        return;
    }
    int gg = ghosted ? 1 : 0;
    long pitch = D->nx + 2 * gg;
    for (int b = 0; b < D->nbox; b++) {
        const OrFab *f = &D->cell[field][b];
        OrBox v = D->valid[b];
        /* ghosted read-back returns, for ghost cells, only those outside the domain
         * (taken from the box that owns the adjacent valid cell); interior ones
         * duplicate neighbouring valid data. */
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++)
                g[(long)(j + gg) * pitch + (i + gg)] = AT(f, i, j, 0);
        if (ghosted && D->ghost[field] > 0) {
            if (v.lo0 == 0) for (int j = v.lo1; j <= v.hi1; j++) g[(long)(j + 1) * pitch + 0] = AT(f, -1, j, 0);
            if (v.hi0 == D->nx - 1) for (int j = v.lo1; j <= v.hi1; j++) g[(long)(j + 1) * pitch + D->nx + 1] = AT(f, D->nx, j, 0);
            if (v.lo1 == 0) for (int i = v.lo0; i <= v.hi0; i++) g[(long)0 * pitch + i + 1] = AT(f, i, -1, 0);
            if (v.hi1 == D->ny - 1) for (int i = v.lo0; i <= v.hi0; i++) g[(long)(D->ny + 1) * pitch + i + 1] = AT(f, i, D->ny, 0);
        }
    }
}

/* ---------------- exchange / BC ---------------- */

/* LevelData::exchange with a Copier built by exchangeDefine(grids, Unit) + trimEdges
 * (src/VCAMRNonLinearPoissonOp.cpp:912-913): 1-cell face ghosts from the adjacent
 * box's valid cells, periodic images included, no corner regions.  [Chombo] */
static void exchange_fabs(const OrLevel *L, const Depth *D, OrFab *fabs, int ncomp, int corners)
{
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        int bi = b % D->nbx, bj = b / D->nbx;
        OrBox v = D->valid[b];
        OrFab *f = &fabs[b];
        for (int side = 0; side < 2; side++) {
            /* x direction */
            int nbi = bi + (side ? 1 : -1), shift = 0;
            if (nbi < 0) { if (!L->bc.periodic[0]) nbi = -99; else { nbi = D->nbx - 1; shift = D->nx; } }
            else if (nbi >= D->nbx) { if (!L->bc.periodic[0]) nbi = -99; else { nbi = 0; shift = -D->nx; } }
            if (nbi != -99) {
                const OrFab *nf = &fabs[bj * D->nbx + nbi];
                int ig = side ? v.hi0 + 1 : v.lo0 - 1;
                for (int n = 0; n < ncomp; n++)
                    for (int j = v.lo1; j <= v.hi1; j++) AT(f, ig, j, n) = AT(nf, ig + shift, j, n);
            }
            /* y direction */
            int nbj = bj + (side ? 1 : -1); shift = 0;
            if (nbj < 0) { if (!L->bc.periodic[1]) nbj = -99; else { nbj = D->nby - 1; shift = D->ny; } }
            else if (nbj >= D->nby) { if (!L->bc.periodic[1]) nbj = -99; else { nbj = 0; shift = -D->ny; } }
            if (nbj != -99) {
                const OrFab *nf = &fabs[nbj * D->nbx + bi];
                int jg = side ? v.hi1 + 1 : v.lo1 - 1;
                for (int n = 0; n < ncomp; n++)
                    for (int i = v.lo0; i <= v.hi0; i++) AT(f, i, jg, n) = AT(nf, i, jg + shift, n);
            }
        }
        if (corners) { /* plain exchange() (no trimEdges): corner ghosts too */
            for (int sx = -1; sx <= 1; sx += 2)
                for (int sy = -1; sy <= 1; sy += 2) {
                    int ig = sx < 0 ? v.lo0 - 1 : v.hi0 + 1, jg = sy < 0 ? v.lo1 - 1 : v.hi1 + 1;
                    int is = ig, js = jg;
                    if (is < 0) { if (!L->bc.periodic[0]) continue; is += D->nx; }
                    if (is >= D->nx) { if (!L->bc.periodic[0]) continue; is -= D->nx; }
                    if (js < 0) { if (!L->bc.periodic[1]) continue; js += D->ny; }
                    if (js >= D->ny) { if (!L->bc.periodic[1]) continue; js -= D->ny; }
                    const OrFab *nf = &fabs[(js / D->bsy) * D->nbx + (is / D->bsx)];
                    for (int n = 0; n < ncomp; n++) AT(f, ig, jg, n) = AT(nf, is, js, n);
                }
        }
    }
}

void or_level_exchange(OrLevel *L, int depth, int field)
{
    Depth *D = &L->d[depth];
    exchange_fabs(L, D, D->cell[field], 1, 0);
}

/* AmrHydro::mixBCValues, src/AmrHydro.cpp:248-309, on one box.  DiriBC(order 1):
 * ghost = 2*value - near; NeumBC: ghost = near + sign*dx*value  [Chombo BCFunc]. */
static void mix_bc_values(const OrLevel *L, const Depth *D, OrFab *state, OrBox valid, int homogeneous)
{
    /* :254  nothing to do if the ghosted fab lies inside the domain */
    if (state->lo0 >= 0 && state->lo1 >= 0 && state->hi0 <= D->nx - 1 && state->hi1 <= D->ny - 1) return;
    for (int dir = 0; dir < 2; dir++) {
        if (L->bc.periodic[dir]) continue; /* :263 */
        int ndom = dir == 0 ? D->nx : D->ny;
        for (int side = 0; side < 2; side++) {
            int vlo = dir == 0 ? valid.lo0 : valid.lo1, vhi = dir == 0 ? valid.hi0 : valid.hi1;
            int g = side == 0 ? vlo - 1 : vhi + 1;         /* adjCellBox(valid,dir,side,1) */
            if (g >= 0 && g <= ndom - 1) continue;          /* :267/:287 strip inside the domain */
            int isign = side == 0 ? -1 : 1;
            int type = L->bc.type[dir][side];
            double value = homogeneous ? 0.0 : L->bc.value[dir][side];
            int tlo = dir == 0 ? valid.lo1 : valid.lo0, thi = dir == 0 ? valid.hi1 : valid.hi0;
            for (int t = tlo; t <= thi; t++) {
                int ig = dir == 0 ? g : t, jg = dir == 0 ? t : g;
                int in = dir == 0 ? g - isign : t, jn = dir == 0 ? t : g - isign;
                double nearVal = AT(state, in, jn, 0);
                if (type == 0) {
                    AT(state, ig, jg, 0) = 2.0 * value - nearVal;
                } else if (type == 1) {
                    double gv = nearVal;
                    if (!homogeneous) gv += (double)isign * D->dx[dir] * value;
                    AT(state, ig, jg, 0) = gv;
                }
            }
        }
    }
}

void or_level_bc(OrLevel *L, int depth, int field, int homogeneous)
{
    Depth *D = &L->d[depth];
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) mix_bc_values(L, D, &D->cell[field][b], D->valid[b], homogeneous);
}

/* ---------------- operator pieces ---------------- */

/* VCAMRNonLinearPoissonOp::resetLambda, src/VCAMRNonLinearPoissonOp.cpp:505-534 */
void or_level_reset_lambda(OrLevel *L, int depth)
{
    if (!L->lambda_dirty[depth]) return;
    Depth *D = &L->d[depth];
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        OrFab *lam = &D->cell[OR_F_LAMBDA][b];
        const OrFab *a = &D->cell[OR_F_ACOEF][b];
        OrBox cur = D->valid[b];
        for (int j = cur.lo1; j <= cur.hi1; j++)
            for (int i = cur.lo0; i <= cur.hi0; i++) AT(lam, i, j, 0) = AT(a, i, j, 0) * L->alpha; /* copy; mult :517-518 */
        for (int dir = 0; dir < 2; dir++) {
            double scale = 1.0 / (D->dx[dir] * D->dx[dir]);
            or_sumfacesnl(lam, L->beta, &D->face[dir][b], cur, dir, scale);
        }
    }
    L->lambda_dirty[depth] = 0;
}

/* AmrHydro::NonLinear_level, src/AmrHydro.cpp:1542-1574 */
void or_level_nonlinear(OrLevel *L, int depth)
{
    Depth *D = &L->d[depth];
    if (!L->ph.use_NL) {
        /* reference leaves freshly allocated (indeterminate) data here (:1551); the
         * restatement defines it as zero */
        for (int b = 0; b < D->nbox; b++) { fab_setval(&D->cell[OR_F_NL][b], 0.0); fab_setval(&D->cell[OR_F_DNL][b], 0.0); }
        return;
    }
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++)
        or_computenonlinearterms(&D->cell[OR_F_PHI][b], &D->cell[OR_F_B][b], &D->cell[OR_F_MASK][b],
                                 &D->cell[OR_F_PI][b], &D->cell[OR_F_ZB][b], D->valid[b],
                                 &D->cell[OR_F_NL][b], &D->cell[OR_F_DNL][b], &L->ph);
}

/* VCAMRNonLinearPoissonOp::levelGSRB, src/VCAMRNonLinearPoissonOp.cpp:654-760,
 * called `sweeps` times by AMRNonLinearPoissonOp::relax (:707-750, relaxMode 1). */
void or_level_gsrb(OrLevel *L, int depth, int sweeps)
{
    Depth *D = &L->d[depth];
    for (int it = 0; it < sweeps; it++) {
        or_level_reset_lambda(L, depth);                               /* :671 */
        for (int whichPass = 0; whichPass <= 1; whichPass++) {         /* :680 */
            or_level_exchange(L, depth, OR_F_PHI);                      /* :692 */
            or_level_bc(L, depth, OR_F_PHI, 0);                         /* :698-703 */
            or_level_nonlinear(L, depth);                               /* :705 */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
            for (int b = 0; b < D->nbox; b++)                           /* :708-745 */
                or_gsrbhelmholtzvcnl2d(&D->cell[OR_F_PHI][b], &D->cell[OR_F_RHS][b], D->valid[b], D->dx,
                                       L->alpha, &D->cell[OR_F_ACOEF][b], L->beta,
                                       &D->face[0][b], &D->face[1][b],
                                       &D->cell[OR_F_NL][b], &D->cell[OR_F_DNL][b],
                                       &D->cell[OR_F_LAMBDA][b], whichPass);
        }
        or_level_exchange(L, depth, OR_F_PHI);                          /* :751 */
        or_level_bc(L, depth, OR_F_PHI, 1);                             /* :757-759 homogeneous */
    }
}

/* applyOpI + applyOpNoBoundary, src/VCAMRNonLinearPoissonOp.cpp:273-345 */
void or_level_apply_op(OrLevel *L, int depth, int homogeneous)
{
    Depth *D = &L->d[depth];
    or_level_bc(L, depth, OR_F_PHI, homogeneous);  /* :283-286 */
    or_level_exchange(L, depth, OR_F_PHI);         /* :304 */
    or_level_nonlinear(L, depth);                  /* :308 */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++)
        or_vcnlcomputeop2d(&D->cell[OR_F_LPHI][b], &D->cell[OR_F_PHI][b], L->alpha, &D->cell[OR_F_ACOEF][b],
                           L->beta, &D->face[0][b], &D->face[1][b], &D->cell[OR_F_NL][b], D->valid[b], D->dx);
}

/* residualI, src/VCAMRNonLinearPoissonOp.cpp:98-167 */
void or_level_residual(OrLevel *L, int depth)
{
    Depth *D = &L->d[depth];
    or_level_bc(L, depth, OR_F_PHI, 0);
    or_level_exchange(L, depth, OR_F_PHI);
    or_level_nonlinear(L, depth);
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++)
        or_vcnlcomputeres2d(&D->cell[OR_F_RES][b], &D->cell[OR_F_PHI][b], &D->cell[OR_F_RHS][b], L->alpha,
                            &D->cell[OR_F_ACOEF][b], L->beta, &D->face[0][b], &D->face[1][b],
                            &D->cell[OR_F_NL][b], D->valid[b], D->dx);
}

/* view of a fab with its index space shifted by -(s0,s1)  (CHF_FRA_SHIFT) */
static OrFab shifted(const OrFab *f, int s0, int s1)
{
    OrFab g = *f;
    g.lo0 -= s0; g.hi0 -= s0; g.lo1 -= s1; g.hi1 -= s1;
    return g;
}

/* restrictResidual (5-arg, phiCoarse == nullptr), src/VCAMRNonLinearPoissonOp.cpp:384-460 */
void or_level_restrict_residual(OrLevel *L, int depth)
{
    Depth *D = &L->d[depth], *C = &L->d[depth + 1];
    or_level_bc(L, depth, OR_F_PHI, 0);     /* :400-403 */
    or_level_exchange(L, depth, OR_F_PHI);  /* :405 */
    or_level_nonlinear(L, depth);           /* :409 */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        OrBox region = D->valid[b];
        int iv0 = region.lo0, iv1 = region.lo1;
        int civ0 = iv0 / 2, civ1 = iv1 / 2;                 /* coarsen(iv,2), iv >= 0 */
        OrFab *res = &C->cell[OR_F_RES][b];
        fab_setval(res, 0.0);                               /* :427 */
        OrFab rs = shifted(res, civ0, civ1);
        OrFab phi = shifted(&D->cell[OR_F_PHI][b], iv0, iv1), rhs = shifted(&D->cell[OR_F_RHS][b], iv0, iv1);
        OrFab a = shifted(&D->cell[OR_F_ACOEF][b], iv0, iv1), nl = shifted(&D->cell[OR_F_NL][b], iv0, iv1);
        OrFab b0 = shifted(&D->face[0][b], iv0, iv1), b1 = shifted(&D->face[1][b], iv0, iv1);
        OrBox rshift = {0, 0, region.hi0 - iv0, region.hi1 - iv1};
        or_restrictresvcnl2d(&rs, &phi, &rhs, L->alpha, &a, L->beta, &b0, &b1, &nl, rshift, D->dx);
    }
}

/* restrictR, src/VCAMRNonLinearPoissonOp.cpp:347-372 */
void or_level_restrict_r(OrLevel *L, int depth)
{
    Depth *D = &L->d[depth], *C = &L->d[depth + 1];
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        OrBox region = D->valid[b];
        int iv0 = region.lo0, iv1 = region.lo1;
        OrFab *pc = &C->cell[OR_F_PHI][b];
        fab_setval(pc, 0.0);                                /* :365 (ghosts zeroed too) */
        OrFab pcs = shifted(pc, iv0 / 2, iv1 / 2);
        OrFab pf = shifted(&D->cell[OR_F_PHI][b], iv0, iv1);
        OrBox rshift = {0, 0, region.hi0 - iv0, region.hi1 - iv1};
        or_restrictvcnl(&pcs, &pf, rshift);
    }
}

/* prolongIncrement, src/AMRNonLinearPoissonOp.cpp:856-886; the correction is held in
 * W_CORR of depth+1.  If coarse_corr != NULL it is first loaded from that global array. */
void or_level_prolong_increment(OrLevel *L, int depth, const double *coarse_corr)
{
    Depth *D = &L->d[depth], *C = &L->d[depth + 1];
    if (coarse_corr) or_level_set(L, depth + 1, W_CORR, coarse_corr, 0);
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        OrBox region = D->valid[b];
        int iv0 = region.lo0, iv1 = region.lo1;
        OrFab phi = shifted(&D->cell[OR_F_PHI][b], iv0, iv1);
        OrFab crs = shifted(&C->cell[W_CORR][b], iv0 / 2, iv1 / 2);
        OrBox rshift = {0, 0, region.hi0 - iv0, region.hi1 - iv1};
        or_prolongnl(&phi, &crs, rshift, 2);
    }
}

/* util/ExtrapGhostCells.cpp:94-180 for cell-centred data (dir_edges = -1), rad = 1 */
static void extrap_ghost_cells(const OrLevel *L, const Depth *D, OrFab *phi)
{
    for (int dir = 0; dir < 2; dir++) {
        if (L->bc.periodic[dir]) continue; /* :110 */
        int ndom = dir == 0 ? D->nx : D->ny;
        for (int hiLo = 0; hiLo < 2; hiLo++) {
            /* adjCellLo/Hi(domainBox, dir, 1), grown by 1 transversally (:133-134,165-166),
             * intersected with the fab box */
            OrBox s;
            int g = hiLo == 0 ? -1 : ndom;
            if (dir == 0) { s.lo0 = s.hi0 = g; s.lo1 = -1; s.hi1 = D->ny; }
            else          { s.lo1 = s.hi1 = g; s.lo0 = -1; s.hi0 = D->nx; }
            if (s.lo0 < phi->lo0) s.lo0 = phi->lo0;
            if (s.hi0 > phi->hi0) s.hi0 = phi->hi0;
            if (s.lo1 < phi->lo1) s.lo1 = phi->lo1;
            if (s.hi1 > phi->hi1) s.hi1 = phi->hi1;
            if (s.lo0 > s.hi0 || s.lo1 > s.hi1) continue;
            or_simpleextrapbc(phi, s, dir, hiLo);
        }
    }
}

/* AmrHydro::WFlx_level, src/AmrHydro.cpp:1415-1539 (single level: a_ucoarse == NULL) */
static void wflx_level(OrLevel *L, int depth)
{
    Depth *D = &L->d[depth];
    if (!D->gradH) {
        D->gradH = (OrFab *)malloc(sizeof(OrFab) * D->nbox);
        D->Re = (OrFab *)malloc(sizeof(OrFab) * D->nbox);
        for (int b = 0; b < D->nbox; b++) { D->gradH[b] = fab_alloc(D->valid[b], 1, 2); D->Re[b] = fab_alloc(D->valid[b], 1, 1); }
    }
    int hasMask = L->ph.use_mask_gradients;
    /* Gradient::compGradientCC (util/Gradient.cpp:477-624): MAC normal gradient on the
     * faces of the valid box (levelGradientMAC :96-127, NEWMACGRAD), then EdgeToCell
     * (:623, [Chombo]: cell = half*(face(i) + face(i+e))) */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        OrBox v = D->valid[b];
        fab_setval(&D->gradH[b], 0.0);
        for (int dir = 0; dir < 2; dir++) {
            OrBox eb = v; if (dir == 0) eb.hi0 += 1; else eb.hi1 += 1;
            OrFab eg = fab_alloc(eb, 0, 1);
            or_newmacgrad(&eg, &D->cell[OR_F_MASK][b], &D->cell[OR_F_PHI][b], eb, D->dx, dir, hasMask);
            int ii = dir == 0, jj = dir == 1;
            for (int j = v.lo1; j <= v.hi1; j++)
                for (int i = v.lo0; i <= v.hi0; i++)
                    AT(&D->gradH[b], i, j, dir) = 0.5 * (AT(&eg, i, j, 0) + AT(&eg, i + ii, j + jj, 0));
            free(eg.p);
        }
    }
    exchange_fabs(L, D, D->gradH, 2, 1);                       /* lvlgradH.exchange() :1490 */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        extrap_ghost_cells(L, D, &D->gradH[b]);                /* :1491 */
        OrBox region = {D->Re[b].lo0, D->Re[b].lo1, D->Re[b].hi0, D->Re[b].hi1}; /* ghosted :1497 */
        or_computere(&D->cell[OR_F_B][b], &D->gradH[b], region, &D->Re[b], &L->ph);
    }
    /* CellToEdge(Re), CellToEdge(B) [Chombo: face = half*(cell(i) + cell(i-e))],
     * setup_iceMask_EC (src/HydroIBC.cpp:139-184), COMPUTEBCOEFF per direction */
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < D->nbox; b++) {
        for (int dir = 0; dir < 2; dir++) {
            OrFab *bC = &D->face[dir][b];
            OrBox fb = {bC->lo0, bC->lo1, bC->hi0, bC->hi1};
            OrFab B_ec = fab_alloc(fb, 0, 1), Re_ec = fab_alloc(fb, 0, 1), IM_ec = fab_alloc(fb, 0, 1);
            int ii = dir == 0, jj = dir == 1;
            const OrFab *Bc = &D->cell[OR_F_B][b], *Rc = &D->Re[b], *IM = &D->cell[OR_F_MASK][b];
            int face_lo = 0, face_hi = dir == 0 ? D->nx : D->ny; /* domain face box ends */
            for (int j = fb.lo1; j <= fb.hi1; j++)
                for (int i = fb.lo0; i <= fb.hi0; i++) {
                    AT(&Re_ec, i, j, 0) = 0.5 * (AT(Rc, i, j, 0) + AT(Rc, i - ii, j - jj, 0));
                    AT(&B_ec, i, j, 0) = 0.5 * (AT(Bc, i, j, 0) + AT(Bc, i - ii, j - jj, 0));
                    double m = AT(IM, i, j, 0), mm1 = AT(IM, i - ii, j - jj, 0), mec;
                    if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; /* HydroIBC.cpp:162-167 */
                    else mec = 0.0;
                    int idx = dir == 0 ? i : j;
                    if (idx == face_lo) mec = 0.0;                            /* :172-177 */
                    if (idx == face_hi) mec = 0.0;
                    AT(&IM_ec, i, j, 0) = mec;
                }
            or_computebcoeff(&B_ec, &Re_ec, fb, bC, &IM_ec, &L->ph);           /* :1528-1535 */
            free(B_ec.p); free(Re_ec.p); free(IM_ec.p);
        }
    }
}

/* UpdateOperator, src/VCAMRNonLinearPoissonOp.cpp:34-64 */
void or_level_update_operator(OrLevel *L, int depth)
{
    or_level_exchange(L, depth, OR_F_PHI);   /* :47 */
    or_level_bc(L, depth, OR_F_PHI, 0);      /* :50-53 */
    wflx_level(L, depth);                    /* :56 */
    L->lambda_dirty[depth] = 1;              /* :61 */
    or_level_reset_lambda(L, depth);         /* :62 */
}

/* AverageOperator, src/VCAMRNonLinearPoissonOp.cpp:66-95: CoarseAverageFace of the
 * depth-0 bCoef with ratio 2^depth, arithmetic.  [Chombo AVERAGEFACE: sequential sum of
 * the r collinear fine faces, divided by r] */
void or_level_average_operator(OrLevel *L, int depth)
{
    if (depth == 0) { L->lambda_dirty[0] = 1; or_level_reset_lambda(L, 0); return; }
    Depth *F = &L->d[0], *C = &L->d[depth];
    int r = 1 << depth;
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
    for (int b = 0; b < C->nbox; b++) {
        for (int dir = 0; dir < 2; dir++) {
            OrFab *cf = &C->face[dir][b];
            const OrFab *ff = &F->face[dir][b];
            for (int jc = cf->lo1; jc <= cf->hi1; jc++)
                for (int ic = cf->lo0; ic <= cf->hi0; ic++) {
                    double s = 0.0;
                    for (int k = 0; k < r; k++)
                        s = s + (dir == 0 ? AT(ff, ic * r, jc * r + k, 0) : AT(ff, ic * r + k, jc * r, 0));
                    AT(cf, ic, jc, 0) = s / (double)r;
                }
        }
    }
    L->lambda_dirty[depth] = 1;   /* :93 */
    or_level_reset_lambda(L, depth);
}

/* MGnewOp coefficient coarsening, src/VCAMRNonLinearPoissonOp.cpp:1096-1173: aCoef, B,
 * Pi, zb, iceMask by CoarseAverage (arithmetic) of the depth-0 data with ratio 2^depth;
 * bCoef by CoarseAverageFace; Neumann copy of B into domain ghosts (NeumBCForB
 * :1309-1341).  [Chombo AVERAGE: sequential sum, ii fastest, times 1/r^2] */
void or_level_build_mg_coefficients(OrLevel *L)
{
    static const int fields[5] = {OR_F_ACOEF, OR_F_B, OR_F_PI, OR_F_ZB, OR_F_MASK};
    Depth *F = &L->d[0];
    for (int dep = 1; dep < L->ndepth; dep++) {
        Depth *C = &L->d[dep];
        int r = 1 << dep;
        double refScale = 1.0 / (double)(r * r);
#pragma omp parallel for num_threads(L->nthreads) schedule(static)
        for (int b = 0; b < C->nbox; b++) {
            OrBox v = C->valid[b];
            for (int q = 0; q < 5; q++) {
                OrFab *cf = &C->cell[fields[q]][b];
                const OrFab *ff = &F->cell[fields[q]][b];
                for (int jc = v.lo1; jc <= v.hi1; jc++)
                    for (int ic = v.lo0; ic <= v.hi0; ic++) {
                        double s = 0.0;
                        for (int jj = 0; jj < r; jj++)
                            for (int ii = 0; ii < r; ii++) s = s + AT(ff, ic * r + ii, jc * r + jj, 0);
                        AT(cf, ic, jc, 0) = s * refScale;
                    }
            }
        }
        /* ghost cells of the coarse B/Pi/zb/mask: inter-box by exchange, domain sides by
         * Neumann copy (B only in the reference; harmless for the others, which are
         * never read outside valid cells at depth > 0) */
        for (int q = 1; q < 5; q++) {
            exchange_fabs(L, C, C->cell[fields[q]], 1, 0);
            for (int b = 0; b < C->nbox; b++) {
                OrFab *f = &C->cell[fields[q]][b];
                OrBox v = C->valid[b];
                if (!L->bc.periodic[0]) {
                    if (v.lo0 == 0) for (int j = v.lo1; j <= v.hi1; j++) AT(f, -1, j, 0) = AT(f, 0, j, 0);
                    if (v.hi0 == C->nx - 1) for (int j = v.lo1; j <= v.hi1; j++) AT(f, C->nx, j, 0) = AT(f, C->nx - 1, j, 0);
                }
                if (!L->bc.periodic[1]) {
                    if (v.lo1 == 0) for (int i = v.lo0; i <= v.hi0; i++) AT(f, i, -1, 0) = AT(f, i, 0, 0);
                    if (v.hi1 == C->ny - 1) for (int i = v.lo0; i <= v.hi0; i++) AT(f, i, C->ny, 0) = AT(f, i, C->ny - 1, 0);
                }
            }
        }
        or_level_average_operator(L, dep);
    }
}

/* norm over valid cells: ord 0 = max |x|; ord 2 = sqrt(sum x^2) in box order, j, i */
double or_level_norm(OrLevel *L, int depth, int field, int ord)
{
    Depth *D = &L->d[depth];
    double r = 0.0;
    for (int b = 0; b < D->nbox; b++) {
        const OrFab *f = &D->cell[field][b];
        OrBox v = D->valid[b];
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++) {
                double x = AT(f, i, j, 0);
                if (ord == 0) { if (fabs(x) > r) r = fabs(x); }
                else r += x * x;
            }
    }
    return ord == 0 ? r : sqrt(r);
}

/* ---------------- FAS multigrid cycle ----------------
 * The cycle driver (AMRFASMultiGrid) lives in the un-vendored Chombo fork; this is the
 * reconstruction documented in SURVEY.md Appendix D / DESIGN.md ("unpinned"). */
/* Test-only variants of what the un-vendored fork may do differently (tools/stopping_rule_sweep.py; never set in a parity run):
 *   SUHMO_ORACLE_STOP    bit 0: no exit on normThresh (only eps x the initial norm, the hang test and maxIter end a solve)
 *                        bit 1: imin is a hard minimum number of cycles (upstream: it only postpones the hang test)
 *   SUHMO_ORACLE_BOTTOM  1: the bottom relaxes are followed by Chombo's RelaxSolver::solve as the fork's three-argument preCond
 *                        (src/VCAMRNonLinearPoissonOp.cpp:233-271 = relax(phi, rhs, 2)) would run it: up to imax = 40 rounds of two
 *                        sweeps, ended by an l2 residual below 1e-6 x its first value or reduced by less than 10 % */
static int or_variant(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
static void fas_cycle(OrLevel *L, int dep, const OrSolverParams *sp, int ndepth)
{
    Depth *D = &L->d[dep];
    if (dep == ndepth - 1) {                       /* coarsest: bottom relaxes */
        or_level_gsrb(L, dep, sp->num_bottom);
        if (or_variant("SUHMO_ORACLE_BOTTOM") == 1) {
            or_level_residual(L, dep);
            double norm = or_level_norm(L, dep, OR_F_RES, 2), first = norm;
            for (int it = 0; it < 40 && norm > 1.0e-20; it++) {
                or_level_gsrb(L, dep, 2);
                or_level_residual(L, dep);
                double old = norm;
                norm = or_level_norm(L, dep, OR_F_RES, 2);
                if (norm < 1.0e-6 * first || norm > old * (1.0 - 0.1)) break;
            }
        }
        return;
    }
    Depth *C = &L->d[dep + 1];
    or_level_gsrb(L, dep, sp->num_smooth);         /* pre-smooth */
    or_level_restrict_residual(L, dep);            /* RES[dep+1] */
    or_level_restrict_r(L, dep);                   /* PHI[dep+1] */
    for (int b = 0; b < C->nbox; b++) fab_copy(&C->cell[W_PHIOLD][b], &C->cell[OR_F_PHI][b]);
    or_level_apply_op(L, dep + 1, 0);              /* applyOpMg: LPHI[dep+1] = L_c(R phi) */
    for (int b = 0; b < C->nbox; b++) {            /* rhs_c = res_c + L_c(R phi) */
        OrBox v = C->valid[b];
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++)
                AT(&C->cell[OR_F_RHS][b], i, j, 0) = AT(&C->cell[OR_F_RES][b], i, j, 0) + AT(&C->cell[OR_F_LPHI][b], i, j, 0);
    }
    fas_cycle(L, dep + 1, sp, ndepth);
    for (int b = 0; b < C->nbox; b++) {            /* corr = phi_c - phi_c_old */
        OrBox v = C->valid[b];
        for (int j = v.lo1; j <= v.hi1; j++)
            for (int i = v.lo0; i <= v.hi0; i++)
                AT(&C->cell[W_CORR][b], i, j, 0) = AT(&C->cell[OR_F_PHI][b], i, j, 0) - AT(&C->cell[W_PHIOLD][b], i, j, 0);
    }
    or_level_prolong_increment(L, dep, NULL);
    or_level_gsrb(L, dep, sp->num_smooth);         /* post-smooth */
    (void)D;
}

static int eff_depths(const OrLevel *L, const OrSolverParams *sp)
{
    int nd = L->ndepth;
    if (sp->max_depth >= 0 && sp->max_depth + 1 < nd) nd = sp->max_depth + 1;
    return nd;
}

void or_level_vcycle(OrLevel *L, const OrSolverParams *sp)
{
    int nd = eff_depths(L, sp);
    if (sp->bcoeff_otf) {
        or_level_update_operator(L, 0);            /* VCAMR...cpp:32-64 on the finest depth */
        for (int k = 1; k < nd; k++) or_level_average_operator(L, k); /* :66-95 */
    }
    fas_cycle(L, 0, sp, nd);
}

int or_level_solve(OrLevel *L, const OrSolverParams *sp, double *hist)
{
    or_level_residual(L, 0);
    double initial_rnorm = or_level_norm(L, 0, OR_F_RES, 0);
    double rnorm = initial_rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    int goNorm = rnorm > sp->norm_thresh;
    int goRedu = rnorm > sp->eps * initial_rnorm;
    int goIter = iter < sp->max_iter;
    int goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last;
    int goMin = iter < sp->iter_min;
    const int variant = or_variant("SUHMO_ORACLE_STOP");           /* test-only, see fas_cycle */
    if (variant & 1) goNorm = 1;
    if (variant & 2) goMin = goMin || iter < sp->imin;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        or_level_vcycle(L, sp);
        or_level_residual(L, 0);
        rnorm = or_level_norm(L, 0, OR_F_RES, 0);
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh;
        goRedu = rnorm > sp->eps * initial_rnorm;
        goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last;
        goMin = iter < sp->iter_min;
        if (variant & 1) goNorm = 1;
        if (variant & 2) goMin = goMin || iter < sp->imin;
    }
    return iter;
}

/* ---------------- single-kernel helpers on global arrays ---------------- */
void or_prolong2_global(double *fine, const double *coarse_ghosted, int nx, int ny)
{
    OrFab phi = {fine, 0, 0, nx - 1, ny - 1, 1};
    OrFab crs = {(double *)coarse_ghosted, -1, -1, nx / 2, ny / 2, 1};
    OrBox region = {0, 0, nx - 1, ny - 1};
    or_prolong_2_nl(&phi, &crs, region, 2);
}

void or_divergence_global(const double *ux, const double *uy, double *div, int nx, int ny,
                          double dx, double dy)
{
    OrFab fx = {(double *)ux, 0, 0, nx, ny - 1, 1}, fy = {(double *)uy, 0, 0, nx - 1, ny, 1};
    OrFab d = {div, 0, 0, nx - 1, ny - 1, 1};
    OrBox g = {0, 0, nx - 1, ny - 1};
    or_divergence(&fx, &d, g, dx, 0);
    or_divergence(&fy, &d, g, dy, 1);
}

void or_difterm_global(const double *phi_ghosted, const double *dx_face, const double *dy_face,
                       double *dterm, int nx, int ny, double dx, double dy)
{
    OrFab phi = {(double *)phi_ghosted, -1, -1, nx, ny, 1};
    OrFab fx = {(double *)dx_face, 0, 0, nx, ny - 1, 1}, fy = {(double *)dy_face, 0, 0, nx - 1, ny, 1};
    OrFab d = {dterm, 0, 0, nx - 1, ny - 1, 1};
    OrBox g = {0, 0, nx - 1, ny - 1};
    double dxv[2] = {dx, dy};
    or_computedifterm2d(&phi, g, dxv, &d, &fx, &fy);
}

void or_getflux_global(const double *phi_ghosted, const double *bface, double *flux,
                       int nx, int ny, int dir, double beta, double dx_dir, int ref)
{
    OrFab phi = {(double *)phi_ghosted, -1, -1, nx, ny, 1};
    OrBox fb = {0, 0, dir == 0 ? nx : nx - 1, dir == 0 ? ny - 1 : ny};
    OrFab b = {(double *)bface, fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1};
    OrFab f = {flux, fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1};
    or_vc_getflux(&f, &phi, &b, fb, dir, beta, dx_dir, ref);
}
