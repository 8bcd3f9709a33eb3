/*
 * level_shim.h -- TEST INFRASTRUCTURE ONLY (not product code).  Parity status: suhmo_oracle.h
 * (pinned end-to-end by the reference's SHMIP A result tables; no kernel-level vectors).
 *
 * A minimal stand-in for the Chombo containers the reference's hot path runs on
 * (DisjointBoxLayout / LevelData<FArrayBox> / LevelData<FluxBox> / Copier::exchange /
 * BCHolder) plus a restatement of the C++ orchestration around the per-box Fortran
 * kernels: VCAMRNonLinearPoissonOp::{levelGSRB, applyOpI, residualI, restrictResidual,
 * restrictR, resetLambda, UpdateOperator, AverageOperator}, AMRNonLinearPoissonOp::
 * {relax, prolongIncrement}, AmrHydro::{WFlx_level, NonLinear_level, mixBCValues} and
 * the (un-vendored, reconstructed) FAS multigrid cycle.  The call order inside each
 * method is the reference's un-fused order, box by box; this is what bench.py times
 * as the CPU baseline ("CPU restatement of reference path").
 *
 * Single AMR level (rectangular domain split into boxes of at most max_box cells per
 * side).  All global arrays passed across this API are C row-major [j][i] (i fastest),
 * i.e. the same memory order as a Fortran a(i,j) array.
 */
#ifndef SUHMO_LEVEL_SHIM_H
#define SUHMO_LEVEL_SHIM_H

#include "suhmo_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bc.lo_bc / bc.hi_bc (0 = Dirichlet, 1 = Neumann) and x.lo_dirich_val etc.
 * (src/AmrHydro.cpp:99-155); index [dir][side], side 0 = lo, 1 = hi. */
typedef struct OrBC {
    int    type[2][2];
    double value[2][2];
    int    periodic[2]; /* AmrHydro.is_periodic */
} OrBC;

/* solver parameters, src/AmrHydro.cpp:737-762 */
typedef struct OrSolverParams {
    int    num_smooth;   /* pre = post = 4 */
    int    num_bottom;   /* 16 (10 if step < 50) */
    int    max_iter;     /* 100 */
    int    iter_min;     /* 2 */
    int    imin;         /* 20 if step < 50, else Chombo default 5 */
    double eps;          /* 1e-7 (1e-10) */
    double hang;         /* 0.01 (1e-4) */
    double norm_thresh;  /* 1e-7 */
    int    bcoeff_otf;   /* solver.bcoeff_otf */
    int    max_depth;    /* -1 = as deep as the boxes allow (MGnewOp rule) */
} OrSolverParams;

typedef struct OrLevel OrLevel; /* one operator per multigrid depth + work arrays */

/* field ids for or_level_set / or_level_get */
enum {
    OR_F_PHI = 0, OR_F_RHS, OR_F_ACOEF, OR_F_B, OR_F_PI, OR_F_ZB, OR_F_MASK,
    OR_F_BX, OR_F_BY, OR_F_LAMBDA, OR_F_RES, OR_F_LPHI, OR_F_NL, OR_F_DNL
};

OrLevel *or_level_create(int nx, int ny, double dx, double dy, int max_box,
                         const OrBC *bc, const OrPhys *phys, double alpha, double beta,
                         int nthreads);
void or_level_destroy(OrLevel *L);
int  or_level_num_depths(const OrLevel *L);
int  or_level_num_boxes(const OrLevel *L);
void or_level_set_cutoffb(OrLevel *L, int v);

/* Copy a global array into / out of the per-box storage of depth `depth`.
 * Cell fields: `ghosted` != 0 means the global array is (ny+2) x (nx+2) and ghost
 * cells are copied too (needed for B, mask whose domain ghosts are caller data);
 * otherwise ny x nx (valid cells only).  BX is ny x (nx+1); BY is (ny+1) x nx. */
void or_level_set(OrLevel *L, int depth, int field, const double *global, int ghosted);
void or_level_get(const OrLevel *L, int depth, int field, double *global, int ghosted);

/* --- restated operator methods (reference file:line in level_shim.c) --- */
void or_level_exchange(OrLevel *L, int depth, int field);
void or_level_bc(OrLevel *L, int depth, int field, int homogeneous);
void or_level_reset_lambda(OrLevel *L, int depth);
void or_level_nonlinear(OrLevel *L, int depth);            /* fills NL, DNL from PHI */
void or_level_gsrb(OrLevel *L, int depth, int sweeps);     /* relax(): levelGSRB x sweeps */
void or_level_apply_op(OrLevel *L, int depth, int homogeneous);  /* LPHI = L(PHI) */
void or_level_residual(OrLevel *L, int depth);             /* RES = RHS - L(PHI) */
void or_level_restrict_residual(OrLevel *L, int depth);    /* RES[depth+1] <- (RHS - L PHI)[depth] */
void or_level_restrict_r(OrLevel *L, int depth);           /* PHI[depth+1] <- avg PHI[depth] */
void or_level_prolong_increment(OrLevel *L, int depth, const double *coarse_corr);
                                                           /* PHI[depth] += P(corr[depth+1]) */
void or_level_update_operator(OrLevel *L, int depth);      /* exchange+BC, WFlx_level, lambda */
void or_level_average_operator(OrLevel *L, int depth);     /* bCoef[depth] <- avg bCoef[0] */
void or_level_build_mg_coefficients(OrLevel *L);           /* MGnewOp coefficient averaging */
double or_level_norm(OrLevel *L, int depth, int field, int ord);

/* one FAS V-cycle on PHI/RHS of depth 0; returns nothing (PHI updated in place) */
void or_level_vcycle(OrLevel *L, const OrSolverParams *sp);
/* AMRMultiGrid::solveNoInit-style loop; returns number of V-cycles taken;
 * resid_hist (if non-NULL, length max_iter+1) receives the residual norms. */
int or_level_solve(OrLevel *L, const OrSolverParams *sp, double *resid_hist);

/* PROLONG_2_NL on a single level pair, for kernel-level parity of a8:
 * fine (ny x nx) += bilinear(coarse ((ny/2+2) x (nx/2+2), ghosted)) */
void or_prolong2_global(double *fine, const double *coarse_ghosted, int nx, int ny);
/* DIVERGENCE / COMPUTEDIFTERM2D / getFlux on global arrays for kernel-level parity */
void or_divergence_global(const double *ux, const double *uy, double *div, int nx, int ny,
                          double dx, double dy);
void or_difterm_global(const double *phi_ghosted, const double *dx_face, const double *dy_face,
                       double *dterm, int nx, int ny, double dx, double dy);
void or_getflux_global(const double *phi_ghosted, const double *bface, double *flux,
                       int nx, int ny, int dir, double beta, double dx_dir, int ref);

#ifdef __cplusplus
}
#endif
#endif
