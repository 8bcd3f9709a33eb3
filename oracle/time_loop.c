/*
 * time_loop.c -- TEST INFRASTRUCTURE ONLY ("next rows" of SURVEY.md 8f: the caller of the
 * head solve).  CPU restatement of one AmrHydro::timeStepFAS (src/AmrHydro.cpp:2254-3460) for a
 * single AMR level (a hierarchy: amr_step.c strings the phases below together per level): distributed or moulin input,
 * with or without the diffusive term, explicit or implicit gap-height update:
 *   [I]   ghosts of h and b, copy into old                         :2360-2445
 *   [II]  Picard loop: lagged quantities -> RHS_h -> SolveForHead_nl -> convergence test
 *                                                                   :2477-3235
 *   [III] re-evaluate Re, Qw, melt rate with the new head; gap-height RHS; forward Euler
 *                                                                   :3238-3421
 * All fields live in GLOBAL ghosted arrays ((ny+2) x (nx+2), cell (i,j) at [(j+1)*(nx+2)+i+1]):
 * every operation here is pointwise or a fixed stencil, so (as for the solve itself) the box
 * decomposition does not change a bit; the head solve goes through the level shim
 * (or_level_solve), i.e. the un-fused box-by-box path.
 * Diffusive term (suhmo.diffFactor): the reference multiplies COMPUTEDIFTERM2D by DiffFactor
 * (:3071, :2145); with DiffFactor = 0 (SHMIP A) it contributes +-0 and is skipped.
 * Pinned end to end by the reference's committed SHMIP A1-A6 / B1-B5 tables (tests/test_oracle_timeloop.py).
 */
#include "level_shim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "time_loop.h"

#define NXG (M->nx + 2)
#define CC(a, i, j) (a)[(size_t)((j) + 1) * NXG + ((i) + 1)]
#define FX(a, i, j) (a)[(size_t)(j) * (M->nx + 1) + (i)]      /* x-faces: ny x (nx+1) */
#define FY(a, i, j) (a)[(size_t)(j) * M->nx + (i)]            /* y-faces: (ny+1) x nx */

OrModel *or_model_create(OrLevel *L, int nx, int ny, double dx, double dy, const OrBC *bc, const OrPhys *ph,
                         const OrModelParams *mp)
{
    OrModel *M = (OrModel *)calloc(1, sizeof(OrModel));
    M->L = L; M->nx = nx; M->ny = ny; M->dx = dx; M->dy = dy; M->bc = *bc; M->ph = *ph; M->mp = *mp;
    size_t nc = (size_t)(nx + 2) * (ny + 2), nfx = (size_t)(nx + 1) * ny, nfy = (size_t)nx * (ny + 1);
    for (int f = 0; f < OM_NCELL; f++) M->c[f] = (double *)calloc(nc, sizeof(double));
    double **fxs[] = {&M->gxf, &M->zxf, &M->bxf, &M->rxf, &M->qx, &M->t1x, &M->t2x, &M->mrxf, &M->dxf};
    double **fys[] = {&M->gyf, &M->zyf, &M->byf, &M->ryf, &M->qy, &M->t1y, &M->t2y, &M->mryf, &M->dyf};
    for (int k = 0; k < 9; k++) { *fxs[k] = (double *)calloc(nfx, sizeof(double)); *fys[k] = (double *)calloc(nfy, sizeof(double)); }
    M->dterm = (double *)calloc((size_t)nx * ny, sizeof(double));
    M->G_max_box = 64; M->G_nthreads = 1;
    M->nxg = nx; M->nyg = ny;                               /* the level spans its domain */
    return M;
}
void or_model_destroy(OrModel *M)
{
    if (!M) return;
    for (int f = 0; f < OM_NCELL; f++) free(M->c[f]);
    free(M->gxf); free(M->gyf); free(M->zxf); free(M->zyf); free(M->bxf); free(M->byf); free(M->rxf); free(M->ryf);
    free(M->qx); free(M->qy); free(M->t1x); free(M->t1y); free(M->t2x); free(M->t2y);
    free(M->mrxf); free(M->mryf); free(M->dxf); free(M->dyf); free(M->dterm);
    if (M->G) or_level_destroy(M->G);
    free(M);
}
double *or_model_field(OrModel *M, int id)
{
    if (id == OM_QWX) return M->qx;
    if (id == OM_QWY) return M->qy;
    return (id >= 0 && id < OM_NCELL) ? M->c[id] : NULL;
}
int or_model_step_index(const OrModel *M) { return M->cur_step; }
void or_model_set_ramp(OrModel *M, double ramp) { M->mp.ramp = ramp; }
void or_model_set_cutoffb(OrModel *M, int v) { M->ph.cutOffB = v; if (M->L) or_level_set_cutoffb(M->L, v); }   /* solver.cut_solve_outside_domain */      /* suhmo.ramp factor of the coming step (:2448-2467) */
void or_model_gap_solver_layout(OrModel *M, int max_box, int nthreads) { M->G_max_box = max_box; M->G_nthreads = nthreads; }

void or_model_set_patch(OrModel *M, int i0, int j0, int nxg, int nyg)
{
    M->i0 = i0; M->j0 = j0; M->nxg = nxg; M->nyg = nyg;
    M->cf[0][0] = i0 > 0; M->cf[0][1] = i0 + M->nx < nxg; M->cf[1][0] = j0 > 0; M->cf[1][1] = j0 + M->ny < nyg;
}
#define DOMSIDE(dir, side) (!M->cf[dir][side])             /* this side of the level is a side of the domain */
/* exchange (periodic wrap) of a ghosted global array; a patch never wraps */
static void wrap_ghosts(OrModel *M, double *a)
{
    if (M->cf[0][0] || M->cf[0][1] || M->cf[1][0] || M->cf[1][1]) return;
    if (M->bc.periodic[0]) for (int j = 0; j < M->ny; j++) { CC(a, -1, j) = CC(a, M->nx - 1, j); CC(a, M->nx, j) = CC(a, 0, j); }
    if (M->bc.periodic[1]) for (int i = 0; i < M->nx; i++) { CC(a, i, -1) = CC(a, i, M->ny - 1); CC(a, i, M->ny) = CC(a, i, 0); }
}
/* exchange + mixBCValues (src/AmrHydro.cpp:248-309) */
static void head_ghosts(OrModel *M, double *h)
{
    wrap_ghosts(M, h);
    for (int dir = 0; dir < 2; dir++) {
        if (M->bc.periodic[dir]) continue;
        int n = dir == 0 ? M->ny : M->nx;
        for (int side = 0; side < 2; side++) {
            if (!DOMSIDE(dir, side)) continue;
            double isign = side == 0 ? -1.0 : 1.0, value = M->bc.value[dir][side];
            for (int t = 0; t < n; t++) {
                int ig, jg, in, jn;
                if (dir == 0) { ig = side ? M->nx : -1; in = side ? M->nx - 1 : 0; jg = jn = t; }
                else { jg = side ? M->ny : -1; jn = side ? M->ny - 1 : 0; ig = in = t; }
                double nearVal = CC(h, in, jn);
                if (M->bc.type[dir][side] == 0) CC(h, ig, jg) = 2.0 * value - nearVal;
                else CC(h, ig, jg) = nearVal + isign * (dir == 0 ? M->dx : M->dy) * value;
            }
        }
    }
}
/* exchange + CopyGhostCells (util/ExtrapGhostCells.cpp:182-269): ghost = nearest interior cell */
static void copy_ghosts(OrModel *M, double *a)
{
    wrap_ghosts(M, a);
    if (!M->bc.periodic[0]) for (int j = -1; j <= M->ny; j++) {
        if (DOMSIDE(0, 0)) CC(a, -1, j) = CC(a, 0, j);
        if (DOMSIDE(0, 1)) CC(a, M->nx, j) = CC(a, M->nx - 1, j); }
    if (!M->bc.periodic[1]) for (int i = -1; i <= M->nx; i++) {
        if (DOMSIDE(1, 0)) CC(a, i, -1) = CC(a, i, 0);
        if (DOMSIDE(1, 1)) CC(a, i, M->ny) = CC(a, i, M->ny - 1); }
}
/* exchange + ExtrapGhostCells (:94-180): ghost = 2*near - far */
static void extrap_ghosts(OrModel *M, double *a)
{
    wrap_ghosts(M, a);
    if (!M->bc.periodic[0]) for (int j = -1; j <= M->ny; j++) {
        if (DOMSIDE(0, 0)) CC(a, -1, j) = 2.0 * CC(a, 0, j) - CC(a, 1, j);
        if (DOMSIDE(0, 1)) CC(a, M->nx, j) = 2.0 * CC(a, M->nx - 1, j) - CC(a, M->nx - 2, j); }
    if (!M->bc.periodic[1]) for (int i = -1; i <= M->nx; i++) {
        if (DOMSIDE(1, 0)) CC(a, i, -1) = 2.0 * CC(a, i, 0) - CC(a, i, 1);
        if (DOMSIDE(1, 1)) CC(a, i, M->ny) = 2.0 * CC(a, i, M->ny - 1) - CC(a, i, M->ny - 2); }
}
void or_model_head_ghosts(OrModel *M, double *h) { head_ghosts(M, h); }
void or_model_copy_ghosts(OrModel *M, double *a) { copy_ghosts(M, a); }
void or_model_extrap_ghosts(OrModel *M, double *a) { extrap_ghosts(M, a); }

/* Gradient::compGradientMAC, normal branch of NEWMACGRAD (util/GradientF.ChF:57-70) */
static void mac_grad(OrModel *M, const double *phi, double *gx, double *gy)
{
    const double *mask = M->c[OM_MASK];
    int hm = M->ph.use_mask_gradients;
    double f0 = 1.0 / M->dx, f1 = 1.0 / M->dy;
    for (int j = 0; j < M->ny; j++)
        for (int i = 0; i <= M->nx; i++) {
            double v = f0 * (CC(phi, i, j) - CC(phi, i - 1, j));
            if (hm && ((CC(mask, i, j) < 1e-6) || (CC(mask, i - 1, j) < 1e-6))) v = 0.0;
            FX(gx, i, j) = v;
        }
    for (int j = 0; j <= M->ny; j++)
        for (int i = 0; i < M->nx; i++) {
            double v = f1 * (CC(phi, i, j) - CC(phi, i, j - 1));
            if (hm && ((CC(mask, i, j) < 1e-6) || (CC(mask, i, j - 1) < 1e-6))) v = 0.0;
            FY(gy, i, j) = v;
        }
}
/* CellToEdge [Chombo]: face = half*(cell(i) + cell(i-e)) on the valid faces */
static void cell_to_edge(OrModel *M, const double *c, double *fx, double *fy)
{
    for (int j = 0; j < M->ny; j++) for (int i = 0; i <= M->nx; i++) FX(fx, i, j) = 0.5 * (CC(c, i, j) + CC(c, i - 1, j));
    for (int j = 0; j <= M->ny; j++) for (int i = 0; i < M->nx; i++) FY(fy, i, j) = 0.5 * (CC(c, i, j) + CC(c, i, j - 1));
}

/* compute_grad_head (:1610-1674) / evaluate_Re_quadratic (:1711-1778) / evaluate_Qw_ec (:1677-1709) */
void or_model_grad(OrModel *M)
{
    double *h = M->c[OM_H], *gx = M->c[OM_GRADX], *gy = M->c[OM_GRADY];
    mac_grad(M, h, M->gxf, M->gyf);
    for (int j = 0; j < M->ny; j++)
        for (int i = 0; i < M->nx; i++) {              /* EdgeToCell */
            CC(gx, i, j) = 0.5 * (FX(M->gxf, i, j) + FX(M->gxf, i + 1, j));
            CC(gy, i, j) = 0.5 * (FY(M->gyf, i, j) + FY(M->gyf, i, j + 1));
        }
    extrap_ghosts(M, gx); extrap_ghosts(M, gy);
}
void or_model_re(OrModel *M)
{
    double *B = M->c[OM_B], *gx = M->c[OM_GRADX], *gy = M->c[OM_GRADY], *Re = M->c[OM_RE];
    for (int j = -1; j <= M->ny; j++)
        for (int i = -1; i <= M->nx; i++) {            /* COMPUTERE on the ghosted box, AmrHydroF.ChF:92-109 */
            double s = sqrt(CC(gx, i, j) * CC(gx, i, j) + CC(gy, i, j) * CC(gy, i, j));
            double b = CC(B, i, j);
            double discr = 1.0 + 4.0 * M->ph.omega * (b * b * b * M->ph.grav * s) / (12.0 * M->ph.nu * M->ph.nu);
            CC(Re, i, j) = (-1.0 + sqrt(discr)) / (2.0 * M->ph.omega);
        }
}
void or_model_qw(OrModel *M)
{
    cell_to_edge(M, M->c[OM_RE], M->rxf, M->ryf);
    /* COMPUTEQW, AmrHydroF.ChF:137-150 */
    for (int j = 0; j < M->ny; j++) for (int i = 0; i <= M->nx; i++) {
        double b = FX(M->bxf, i, j);
        double num_q = -(b * b * b * M->ph.grav * FX(M->gxf, i, j));
        double denom_q = 12.0 * M->ph.nu * (1.0 + M->ph.omega * FX(M->rxf, i, j));
        FX(M->qx, i, j) = num_q / denom_q;
    }
    for (int j = 0; j <= M->ny; j++) for (int i = 0; i < M->nx; i++) {
        double b = FY(M->byf, i, j);
        double num_q = -(b * b * b * M->ph.grav * FY(M->gyf, i, j));
        double denom_q = 12.0 * M->ph.nu * (1.0 + M->ph.omega * FY(M->ryf, i, j));
        FY(M->qy, i, j) = num_q / denom_q;
    }
}

static void grad_re_qw(OrModel *M) { or_model_grad(M); or_model_re(M); or_model_qw(M); }
/* aCoeff_bCoeff (src/AmrHydro.cpp:1781-1817, called at :3087-3102 before SolveForHead_nl): the bCoef the solver's operators are
 * DEFINED with -- COMPUTEBCOEFF of the lagged B_ec and Re_ec with the edge ice mask.  The first residual of the solve (and the
 * stopping rule's initial norm) sees it; every V-cycle then updates it from the head (bcoeff_otf).  bx: ny x (nx+1), by: (ny+1) x nx */
void or_model_bcoef(const OrModel *M, double *bx, double *by)
{
    const double *IM = M->c[OM_MASK];
    for (int dir = 0; dir < 2; dir++) {
        int ii = dir == 0, jj = dir == 1;
        for (int j = 0; j < M->ny + jj; j++)
            for (int i = 0; i < M->nx + ii; i++) {
                double m = CC(IM, i, j), mm1 = CC(IM, i - ii, j - jj), mec;
                if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; else mec = 0.0;
                int idx = dir == 0 ? i + M->i0 : j + M->j0;
                if (idx == 0 || idx == (dir == 0 ? M->nxg : M->nyg)) mec = 0.0;
                double b = dir == 0 ? FX(M->bxf, i, j) : FY(M->byf, i, j), re = dir == 0 ? FX(M->rxf, i, j) : FY(M->ryf, i, j);
                double num_q = -(b * b * b * M->ph.grav);
                double denom_q = 12.0 * M->ph.nu * (1.0 + M->ph.omega * re);
                double v = (mec < 0.0 && M->ph.cutOffB > 0) ? 0.0 : num_q / denom_q;
                if (dir == 0) bx[(size_t)j * (M->nx + 1) + i] = v; else by[(size_t)j * M->nx + i] = v;
            }
    }
}


/* COMPUTESCAPROD + EdgeToCell + Calc_meltingRate (:2954-2979, :2174-2252) on valid cells */
static void melting_rate(OrModel *M)
{
    const OrModelParams *p = &M->mp;
    double *h = M->c[OM_H], *zb = M->c[OM_ZB], *Pw = M->c[OM_PW], *Pi = M->c[OM_PI], *B = M->c[OM_B], *mR = M->c[OM_MR], *IM = M->c[OM_MASK];
    for (int j = 0; j < M->ny; j++) for (int i = 0; i <= M->nx; i++) {
        FX(M->t1x, i, j) = FX(M->qx, i, j) * FX(M->gxf, i, j); FX(M->t2x, i, j) = FX(M->qx, i, j) * FX(M->zxf, i, j); }
    for (int j = 0; j <= M->ny; j++) for (int i = 0; i < M->nx; i++) {
        FY(M->t1y, i, j) = FY(M->qy, i, j) * FY(M->gyf, i, j); FY(M->t2y, i, j) = FY(M->qy, i, j) * FY(M->zyf, i, j); }
    for (int j = 0; j < M->ny; j++)
        for (int i = 0; i < M->nx; i++) {
            double t0 = 0.5 * (FX(M->t1x, i, j) + FX(M->t1x, i + 1, j)), t1 = 0.5 * (FY(M->t1y, i, j) + FY(M->t1y, i, j + 1));
            double u0 = 0.5 * (FX(M->t2x, i, j) + FX(M->t2x, i + 1, j)), u1 = 0.5 * (FY(M->t2y, i, j) + FY(M->t2y, i, j + 1));
            CC(Pw, i, j) = p->gravity * p->rho_w * (CC(h, i, j) - CC(zb, i, j));                     /* :2215 */
            double sca_prod = 0.0;
            if (p->basal_friction) sca_prod = 20. * 20. * p->ub0 * fabs(CC(Pi, i, j) - CC(Pw, i, j)) * p->ub0;   /* :2220 */
            double abs_QPw = t0 + t1 - (u0 + u1);                                                   /* :2225 */
            if ((abs_QPw < 0) && (CC(B, i, j) < 1e-6)) abs_QPw = 0.0;
            double m = p->G + sca_prod - p->rho_w * p->gravity * (t0 + t1)
                       + p->ct * p->cw * p->rho_w * p->rho_w * p->gravity * abs_QPw;                  /* :2231-2234 */
            m = m / p->L;
            m = fmax(m, 0.0);
            if (CC(IM, i, j) < 0.0) m = 0.0;
            CC(mR, i, j) = m;
        }
}

/* dCoeff (src/AmrHydro.cpp:1831-1862, COMPUTEDCOEFF src/AmrHydroF.ChF:241-265) from mR_ec and B_ec; the EC ice mask as
 * setup_iceMask_EC (src/HydroIBC.cpp:139-184); then DiffusiveTerm = COMPUTEDIFTERM2D(b, D) (:2982-2992, ...F.ChF:289-343) */
static void diffusion_coefficients(OrModel *M)
{
    const double *IM = M->c[OM_MASK], *B = M->c[OM_B];
    double *mR = M->c[OM_MR];
    int nx = M->nx, ny = M->ny;
    extrap_ghosts(M, mR);                                   /* levelmR.exchange(); ExtrapGhostCells(levelmR) :2513,:2526 */
    cell_to_edge(M, mR, M->mrxf, M->mryf);                  /* :2530 */
    for (int dir = 0; dir < 2; dir++) {
        int ii = dir == 0, jj = dir == 1;
        for (int j = 0; j < ny + jj; j++)
            for (int i = 0; i < nx + ii; i++) {
                double m = CC(IM, i, j), mm1 = CC(IM, i - ii, j - jj), mec;
                if (fabs(m - mm1) < 1e-10) mec = (m > 0.0) ? 1.0 : -1.0; else mec = 0.0;
                int idx = dir == 0 ? i + M->i0 : j + M->j0;      /* faces of the DOMAIN boundary */
                if (idx == 0 || idx == (dir == 0 ? M->nxg : M->nyg)) mec = 0.0;
                double bec = dir == 0 ? FX(M->bxf, i, j) : FY(M->byf, i, j), mrec = dir == 0 ? FX(M->mrxf, i, j) : FY(M->mryf, i, j), d;
                if (mec < 0.0 && M->ph.cutOffB > 0) d = 0.0; else d = fmax(bec * mrec / M->mp.rho_i, 5.0e-6);
                if (dir == 0) FX(M->dxf, i, j) = d; else FY(M->dyf, i, j) = d;
            }
    }
    double dxinv0 = 1.0 / (M->dx * M->dx), dxinv1 = 1.0 / (M->dy * M->dy);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++)
            M->dterm[(size_t)j * nx + i] =
                (FX(M->dxf, i + 1, j) * (CC(B, i + 1, j) - CC(B, i, j)) * dxinv0 - FX(M->dxf, i, j) * (CC(B, i, j) - CC(B, i - 1, j)) * dxinv0
                 + FY(M->dyf, i, j + 1) * (CC(B, i, j + 1) - CC(B, i, j)) * dxinv1 - FY(M->dyf, i, j) * (CC(B, i, j) - CC(B, i, j - 1)) * dxinv1);
}

/* SolveForGap_nl (src/AmrHydro.cpp:593-662): (aCoef - dt diffFactor div(D grad)) b = rhs, aCoef = 1, bCoef = D, FixedNeumBCFill
 * (copy) on the domain sides; AMRMultiGrid V-cycles with 2 + 2 smoothings, 4 at the bottom, eps = normThresh = 1e-7,
 * hang 1e-6, iterMin 2, imin 10 while step < 50.  [Chombo] VCAMRPoissonOp2 is not in the reference's tree: the operator
 * here is the linear part of VCAMRNonLinearPoissonOp (alpha a phi - beta div(b grad phi)), the cycle the FAS cycle of
 * level_shim.c, which for a linear operator converges to the same solution. */
static void solve_gap_implicit(OrModel *M, double dt, const double *rhs_valid)
{
    int nx = M->nx, ny = M->ny;
    if (!M->G || M->G_dt != dt) {
        if (M->G) or_level_destroy(M->G);
        OrBC nb = M->bc;
        for (int d = 0; d < 2; d++) for (int s = 0; s < 2; s++) { nb.type[d][s] = 1; nb.value[d][s] = 0.0; }
        OrPhys lp = M->ph; lp.use_NL = 0;
        double bsign = 1.0;
        {   /* diagnosis knob (DESIGN.md "suite E"): sign of beta in the stand-in for VCAMRPoissonOp2.  +1: alpha a phi - beta div(D grad phi)
             * (diffusion, the convention of VCAMRNonLinearPoissonOp); -1: upstream Chombo's alpha a phi + beta div(b grad phi) */
            const char *e = getenv("SUHMO_ORACLE_GAP_BETA_SIGN");
            if (e) bsign = atof(e);
        }
        M->G = or_level_create(nx, ny, M->dx, M->dy, M->G_max_box, &nb, &lp, 1.0, bsign * dt * M->mp.diffFactor, M->G_nthreads);
        M->G_dt = dt;
        double *one = (double *)malloc(sizeof(double) * (size_t)nx * ny);
        for (size_t k = 0; k < (size_t)nx * ny; k++) one[k] = 1.0;
        or_level_set(M->G, 0, OR_F_ACOEF, one, 0);                               /* aCoeff_GH :1820-1828 */
        free(one);
        or_level_set(M->G, 0, OR_F_MASK, M->c[OM_MASK], 1);
    }
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nx * ny);
    for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) tmp[(size_t)j * nx + i] = CC(M->c[OM_B], i, j);
    or_level_set(M->G, 0, OR_F_PHI, tmp, 0);                                     /* a_gh_curr = b (initial guess) :3382-3385 */
    or_level_set(M->G, 0, OR_F_RHS, rhs_valid, 0);
    or_level_set(M->G, 0, OR_F_BX, M->dxf, 0);
    or_level_set(M->G, 0, OR_F_BY, M->dyf, 0);
    or_level_build_mg_coefficients(M->G);                                        /* coarse D = average of the fine faces */
    OrSolverParams sp;
    sp.num_smooth = 2; sp.num_bottom = 4; sp.max_iter = 100; sp.iter_min = 2; sp.imin = M->cur_step < 50 ? 10 : 5;
    sp.eps = 1.0e-7; sp.hang = 1.0e-6; sp.norm_thresh = 1.0e-7; sp.bcoeff_otf = 0; sp.max_depth = -1;
    (void)or_level_solve(M->G, &sp, NULL);
    or_level_get(M->G, 0, OR_F_PHI, tmp, 0);
    {   /* diagnosis knob (DESIGN.md "suite E"): cells without ice keep their gap height through the implicit solve (no diffusion of b
         * across the ice margin).  Unset = the restated source: D >= 5e-6 on margin faces (COMPUTEDCOEFF cuts only faces with IMec < 0) */
        const char *e = getenv("SUHMO_ORACLE_GAP_FREEZE_ICEFREE");
        int freeze = (e && atoi(e)) || M->mp.freeze_icefree_gap;
        for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
            if (!(freeze && CC(M->c[OM_MASK], i, j) < 0.0)) CC(M->c[OM_B], i, j) = tmp[(size_t)j * nx + i];
    }
    free(tmp);
}

static double max_valid(OrModel *M, const double *a)
{
    double m = -1e300;
    for (int j = 0; j < M->ny; j++) for (int i = 0; i < M->nx; i++) if (CC(a, i, j) > m) m = CC(a, i, j);
    return m;
}

/* ---- phases of one step; or_model_timestep strings them together for one level, amr_step.c for a hierarchy ---- */
void or_model_begin_step(OrModel *M)                                         /* [I] :2360-2445 */
{
    size_t nc = (size_t)(M->nx + 2) * (M->ny + 2);
    M->cur_step += 1;                                                        /* :2259 */
    head_ghosts(M, M->c[OM_H]); copy_ghosts(M, M->c[OM_B]);
    memcpy(M->c[OM_BOLD], M->c[OM_B], nc * sizeof(double));
    mac_grad(M, M->c[OM_ZB], M->zxf, M->zyf);                                /* compute_grad_zb_ec :1577-1608 (zb is static) */
}
void or_model_solver_params(const OrModel *M, OrSolverParams *sp)            /* SolveForHead_nl :737-762 */
{
    sp->num_smooth = 4; sp->num_bottom = 16; sp->max_iter = 100; sp->iter_min = 2; sp->imin = 5;
    sp->eps = 1.0e-7; sp->hang = 0.01; sp->norm_thresh = 1.0e-7; sp->bcoeff_otf = 1; sp->max_depth = -1;
    if (M->cur_step < 50) { sp->num_bottom = 10; sp->eps = 1.0e-10; sp->hang = 0.0001; sp->imin = 20; }
}
void or_model_begin_iteration(OrModel *M)                                    /* :2482-2532 */
{
    size_t nc = (size_t)(M->nx + 2) * (M->ny + 2);
    head_ghosts(M, M->c[OM_H]); copy_ghosts(M, M->c[OM_B]);
    memcpy(M->c[OM_HLAG], M->c[OM_H], nc * sizeof(double));
    cell_to_edge(M, M->c[OM_B], M->bxf, M->byf);
}
void or_model_rhs_h(OrModel *M)                                              /* :2797-3078 */
{
    const OrModelParams *p = &M->mp;
    int nx = M->nx, ny = M->ny;
    double *B = M->c[OM_B], *IM = M->c[OM_MASK], *src = M->c[OM_SRC], *rhs = M->c[OM_RHSH], *mR = M->c[OM_MR];
    for (int j = -1; j <= ny; j++) for (int i = -1; i <= nx; i++) {         /* distributed input :2865-2877 */
        if (p->use_moulin_source) CC(src, i, j) = CC(M->c[OM_MSRC], i, j) * p->ramp + p->distributed_input;   /* :3065 */
        else CC(src, i, j) = (CC(IM, i, j) > 0.0) ? p->distributed_input : 0.0;
    }
    if (p->diffFactor != 0.0) diffusion_coefficients(M);                      /* lagged mR (before this iteration's melt rate) :2548-2551 */
    melting_rate(M);
    double rho_coef = (1.0 / p->rho_w - 1.0 / p->rho_i);                     /* :3023 */
    {   /* diagnosis knob (tools/run_shmip_a.py --head-melt-coef, DESIGN.md "end-to-end pin"): scales the melt
         * term of RHS_h; unset = the reference's source as it is */
        const char *e = getenv("SUHMO_ORACLE_HEAD_MELT_COEF");
        if (e) rho_coef *= atof(e);
        else if (p->head_melt_off) rho_coef *= 0.0;
    }
    double ub_norm = sqrt(p->ub0 * p->ub0 + p->ub1 * p->ub1);                /* magVel, SqrtIBC.cpp:280-281 */
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {                                       /* :3044-3077 */
            double r = CC(mR, i, j) * rho_coef;
            if (CC(B, i, j) < p->br) r -= ub_norm * (p->br - CC(B, i, j)) / p->lr;
            r += CC(src, i, j);
            if (p->diffFactor != 0.0) r -= p->diffFactor * M->dterm[(size_t)j * nx + i];   /* :3071 */
            if (CC(IM, i, j) < 0.0) r = 0.0;
            CC(rhs, i, j) = r;
        }
}
int or_model_picard_converged(const OrModel *M, double res, int cur_picard)  /* :3196-3228 */
{
    if (M->cur_step < 2) return res < 0.05 && cur_picard > 2;
    if (M->cur_step < 50) return res < 0.05;
    return res < M->mp.eps_picard;
}
/* [III] after the chain was re-evaluated with the new head: melt rate, CalcRHS_gapHeightFAS :2069-2171 and -- explicit
 * update -- forward Euler :3406.  With use_impl_diff the gap height stays and rhs_b (valid cells, nx * ny) receives the
 * right-hand side b + dt RHS of the implicit solve. */
void or_model_gap_rhs(OrModel *M, double dt, double *rhs_b)
{
    const OrModelParams *p = &M->mp;
    int nx = M->nx, ny = M->ny;
    double *B = M->c[OM_B], *Bold = M->c[OM_BOLD], *IM = M->c[OM_MASK], *mR = M->c[OM_MR], *Pi = M->c[OM_PI], *Pw = M->c[OM_PW], *CD = M->c[OM_CD];
    melting_rate(M);
    double ub_norm = sqrt(p->ub0 * p->ub0 + p->ub1 * p->ub1);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double b = CC(B, i, j);
            double RHS = CC(mR, i, j) * (1.0 / p->rho_i), RHS_A = RHS, RHS_B = 0.0;
            if ((CC(IM, i, j) < 0.0) && p->use_mask_rhs_b) { RHS = 0.0; CC(CD, i, j) = 0.0; if (p->use_impl_diff) RHS = b; }
            else {
                if (b < p->br) { RHS += ub_norm * (p->br - b) / p->lr; RHS_B = ub_norm * (p->br - b) / p->lr; }
                double PimPw = CC(Pi, i, j) - CC(Pw, i, j), AbsPimPw = fabs(PimPw);
                if (M->ph.cutOffbr > b) RHS -= M->ph.A * (AbsPimPw * AbsPimPw) * PimPw * b * (1.0 - (M->ph.cutOffbr - b) / M->ph.cutOffbr);
                else if (M->ph.maxOffbr < b) RHS -= M->ph.A * (AbsPimPw * AbsPimPw) * PimPw * b * (1.0 - (M->ph.maxOffbr - b) / M->ph.maxOffbr);
                else RHS -= M->ph.A * (AbsPimPw * AbsPimPw) * PimPw * b;
                if (!p->use_impl_diff && p->diffFactor != 0.0) RHS += p->diffFactor * M->dterm[(size_t)j * nx + i];   /* :2145,:2152,:2159 */
                CC(CD, i, j) = RHS_A / (RHS_A + RHS_B);
                if (p->use_impl_diff) RHS = b + dt * RHS;                                                          /* :2165 */
            }
            if (p->use_impl_diff) rhs_b[(size_t)j * nx + i] = RHS;
            else CC(B, i, j) = RHS * dt + CC(Bold, i, j);     /* forward Euler :3406 */
        }
}
/* ... + the implicit solve of one level :3425-3439, ghosts of b on the domain sides :3419-3420 */
void or_model_gap_update(OrModel *M, double dt)
{
    double *rhs_b = M->mp.use_impl_diff ? (double *)malloc(sizeof(double) * (size_t)M->nx * M->ny) : NULL;
    or_model_gap_rhs(M, dt, rhs_b);
    if (M->mp.use_impl_diff) { solve_gap_implicit(M, dt, rhs_b); free(rhs_b); }
    copy_ghosts(M, M->c[OM_B]);
}
const double *or_model_dcoef(const OrModel *M, int dir) { return dir == 0 ? M->dxf : M->dyf; }

/* one timestep; returns 0, or -1 if the Picard loop exceeds 100 iterations (:3190-3195) */
int or_model_timestep(OrModel *M, double dt, int *picard_iters, int *vcycles_total)
{
    int nx = M->nx, ny = M->ny;
    double *h = M->c[OM_H], *B = M->c[OM_B], *rhs = M->c[OM_RHSH], *hl = M->c[OM_HLAG];
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nx * ny);
    or_model_begin_step(M);
    OrSolverParams sp;
    or_model_solver_params(M, &sp);
    int converged = 0, ite_idx = 0, cur_picard = 0, nv = 0;
    while (!converged) {                                                     /* [II] :2477 */
        or_model_begin_iteration(M);
        grad_re_qw(M);
        or_model_rhs_h(M);
        /* SolveForHead_nl: fields into the operator (factory define), solve */
        for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) tmp[(size_t)j * nx + i] = CC(h, i, j);
        or_level_set(M->L, 0, OR_F_PHI, tmp, 0);
        for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) tmp[(size_t)j * nx + i] = CC(rhs, i, j);
        or_level_set(M->L, 0, OR_F_RHS, tmp, 0);
        or_level_set(M->L, 0, OR_F_B, B, 1);
        {
            double *bx = (double *)malloc(sizeof(double) * (size_t)(nx + 1) * ny), *by = (double *)malloc(sizeof(double) * (size_t)nx * (ny + 1));
            or_model_bcoef(M, bx, by);                                           /* aCoeff_bCoeff :3087-3102 */
            or_level_set(M->L, 0, OR_F_BX, bx, 0); or_level_set(M->L, 0, OR_F_BY, by, 0);
            free(bx); free(by);
        }
        or_level_build_mg_coefficients(M->L);
        if (cur_picard > 0 && getenv("SUHMO_ORACLE_MIN_FIRST_SOLVE_ONLY")) { sp.iter_min = 0; sp.imin = 0; }   /* test-only variant (tools/stopping_rule_sweep.py):
                                                                                    iterMin / imin held per time step, not per solve */
        nv += or_level_solve(M->L, &sp, NULL);
        or_level_get(M->L, 0, OR_F_PHI, tmp, 0);
        for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) CC(h, i, j) = tmp[(size_t)j * nx + i];
        head_ghosts(M, h);                                                   /* :3157-3165 */
        double maxHead = max_valid(M, h), res = 0.0;                         /* :3169-3185 */
        for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
            double d = fabs((CC(hl, i, j) - CC(h, i, j)) / maxHead);
            if (d > res) res = d;
        }
        if (ite_idx > 100) { free(tmp); return -1; }
        converged = or_model_picard_converged(M, res, cur_picard);
        ite_idx++; cur_picard++;
    }
    /* [III] :3238-3421 */
    grad_re_qw(M);                                    /* evaluate_Re_quadratic(lev, true) + evaluate_Qw_ec with the lagged B_ec */
    or_model_gap_update(M, dt);
    M->time += dt;
    if (picard_iters) *picard_iters = ite_idx;
    if (vcycles_total) *vcycles_total = nv;
    free(tmp);
    return 0;
}


/* COMPUTE_TIMEVARYINGRECHARGE (src/AmrHydroF.ChF:346-373) over n cells */
void or_time_varying_recharge(int n, const double *zs, double TK, double background, double *recharge)
{
    const double ddf = 0.01 / 86400., dT_dZ = -0.0075;
    for (int k = 0; k < n; k++) recharge[k] = fmax(ddf * (TK + zs[k] * dT_dZ), 0.0) + background;
}

/* Calc_moulin_integral + Calc_moulin_source_term_distributed (src/AmrHydro.cpp:1866-2066), single level:
 * every moulin is a Gaussian evaluated with a 3 x 3 Gauss-Legendre rule per cell, normalised by its integral over
 * the level so that it delivers exactly moulin_flux; time_factor = max(1 - runoff sin(2 pi (t - t0)/86400), 0).
 * src: valid cells ny x nx (m/s); integ: the nm integrals (m2). */
void or_moulin_source(int nx, int ny, double dx, double dy, int nm, const double *pos, const double *sigma,
                      const double *flux, double time_factor, double *integ, double *src)
{
    const double v[3] = {0.5555555555, 0.8888888888, 0.5555555555};
    const double l[3] = {-0.77459666924 / 2.0, 0.0, 0.77459666924 / 2.0};
    double *ms = (double *)malloc(sizeof(double) * (size_t)nx * ny * nm);
    for (int m = 0; m < nm; m++) integ[m] = 0.0;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double xl[3], yl[3];
            for (int k = 0; k < 3; k++) { xl[k] = (i + 0.5 + l[k]) * dx; yl[k] = (j + 0.5 + l[k]) * dy; }
            for (int m = 0; m < nm; m++) {
                double prefac = 1.0 / (sigma[m] * sqrt(2.0 * 3.14));
                double MS[9];
                for (int b = 0; b < 3; b++)
                    for (int a = 0; a < 3; a++) {
                        double ex = xl[a] - pos[2 * m], ey = yl[b] - pos[2 * m + 1];
                        double rad = ex * ex + ey * ey;                          /* std::pow(., 2) */
                        MS[3 * b + a] = prefac * exp(-1.0 / (2.0 * sigma[m] * sigma[m]) * rad);
                    }
                double val = v[0] * v[0] * MS[0] + v[1] * v[0] * MS[1] + v[2] * v[0] * MS[2]
                           + v[0] * v[1] * MS[3] + v[1] * v[1] * MS[4] + v[2] * v[1] * MS[5]
                           + v[0] * v[2] * MS[6] + v[1] * v[2] * MS[7] + v[2] * v[2] * MS[8];
                ms[((size_t)j * nx + i) * nm + m] = val;
            }
        }
    for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int m = 0; m < nm; m++)
        integ[m] += ms[((size_t)j * nx + i) * nm + m] * dx * dy;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            double sum = 0.0;
            for (int m = 0; m < nm; m++) sum += ms[((size_t)j * nx + i) * nm + m] * time_factor / integ[m] * flux[m];
            src[(size_t)j * nx + i] = sum;
        }
    free(ms);
}
