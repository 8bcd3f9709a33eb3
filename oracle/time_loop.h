/* time_loop.h -- TEST INFRASTRUCTURE ONLY: the model state of oracle/time_loop.c and its phases, shared with the AMR
 * time step (oracle/amr_step.c) */
#ifndef OR_TIME_LOOP_H
#define OR_TIME_LOOP_H
#include "level_shim.h"

typedef struct OrModelParams {
    double rho_i, rho_w, gravity;       /* suhmo_params.cpp:51-53 */
    double G, L, ct, cw;                /* GeoFlux, LatHeat, ct, cw */
    double ub0, ub1;                    /* SlidingVelocity */
    double br, lr;                      /* bump height / spacing */
    double diffFactor;
    double distributed_input;
    double eps_picard;                  /* solver.eps_PicardIte */
    int basal_friction;
    int use_mask_rhs_b;
    int use_moulin_source;              /* suhmo.n_moulins > 0: RHS_h += msrc * ramp + distributed_input (:3060-3066) */
    double ramp;                        /* suhmo.ramp (:2448-2467), 1 when off */
    int use_impl_diff;                  /* solver.use_ImplDiff: gap height by the implicit VC Helmholtz solve (:593-662, :3376-3455) */
    /* run-state settings of the reference's committed result tables; 0, 0 = the committed source.  The
     * environment knobs SUHMO_ORACLE_HEAD_MELT_COEF / SUHMO_ORACLE_GAP_FREEZE_ICEFREE the committed pin runs were made with do the same */
    int head_melt_off;                  /* RHS_h without the melt term (:3046-3048): the term times 0.0 */
    int freeze_icefree_gap;             /* cells with iceMask < 0 keep their gap height through SolveForGap_nl */
} OrModelParams;

enum { OM_H = 0, OM_B, OM_BOLD, OM_PI, OM_ZB, OM_MASK, OM_MR, OM_PW, OM_SRC, OM_RHSH, OM_CD,
       OM_GRADX, OM_GRADY, OM_RE, OM_HLAG, OM_MSRC, OM_NCELL, OM_QWX = 100, OM_QWY = 101 };

typedef struct OrModel {
    OrLevel *L;
    int nx, ny;
    double dx, dy;
    OrBC bc;
    OrPhys ph;
    OrModelParams mp;
    double *c[OM_NCELL];                /* ghosted cell arrays */
    double *gxf, *gyf, *zxf, *zyf;      /* face gradients of h and zb */
    double *bxf, *byf, *rxf, *ryf;      /* B_ec, Re_ec */
    double *qx, *qy;                    /* Qw_ec */
    double *t1x, *t1y, *t2x, *t2y;      /* Qw*gradH, Qw*gradZb on faces */
    double *mrxf, *mryf, *dxf, *dyf;    /* mR_ec, Dcoef (suhmo.diffFactor != 0) */
    double *dterm;                      /* div(D grad b), valid cells */
    OrLevel *G;                         /* implicit gap-height operator (alpha = 1, beta = dt * diffFactor, no NL) */
    double G_dt; int G_max_box, G_nthreads;
    int cur_step;
    double time;
    /* AMR patch (amr_step.c): place of this level in its refined domain; cf[dir][side] = that side of the rectangle lies
     * inside the domain: a coarse-fine side whose ghost cells are data (interpolated by the caller), not BC values */
    int i0, j0, nxg, nyg, cf[2][2];
} OrModel;


OrModel *or_model_create(OrLevel *L, int nx, int ny, double dx, double dy, const OrBC *bc, const OrPhys *ph,
                         const OrModelParams *mp);
void or_model_destroy(OrModel *M);
double *or_model_field(OrModel *M, int id);
void or_model_set_patch(OrModel *M, int i0, int j0, int nxg, int nyg);
int or_model_timestep(OrModel *M, double dt, int *picard_iters, int *vcycles_total);
/* phases of one step (or_model_timestep = these in the reference's order on one level) */
void or_model_head_ghosts(OrModel *M, double *h);       /* exchange + mixBCValues on the domain sides */
void or_model_copy_ghosts(OrModel *M, double *a);       /* exchange + CopyGhostCells */
void or_model_extrap_ghosts(OrModel *M, double *a);     /* exchange + ExtrapGhostCells */
void or_model_begin_step(OrModel *M);                   /* [I] */
void or_model_begin_iteration(OrModel *M);              /* ghosts of h and b, lagged copy, B_ec */
void or_model_grad(OrModel *M);                         /* compute_grad_head: faces, cells, domain-side ghosts */
void or_model_re(OrModel *M);                           /* COMPUTERE on the ghosted box */
void or_model_qw(OrModel *M);                           /* Re_ec, COMPUTEQW */
void or_model_bcoef(const OrModel *M, double *bx, double *by);   /* aCoeff_bCoeff: bCoef the solver is defined with */
void or_model_rhs_h(OrModel *M);                        /* source, lagged diffusion, melt rate, RHS_h */
void or_model_solver_params(const OrModel *M, OrSolverParams *sp);
int or_model_picard_converged(const OrModel *M, double res, int cur_picard);
void or_model_gap_rhs(OrModel *M, double dt, double *rhs_b);   /* melt rate, CalcRHS_gapHeightFAS; explicit: forward Euler, implicit: rhs_b */
const double *or_model_dcoef(const OrModel *M, int dir);        /* D on the faces (lagged, from the last or_model_rhs_h) */
void or_model_gap_update(OrModel *M, double dt);        /* melt rate, CalcRHS_gapHeightFAS, forward Euler / implicit solve, ghosts */
#endif
