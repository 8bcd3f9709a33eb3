// suhmo_hier.h -- what suhmo_step.hip needs from a hierarchy of box unions (suhmo_hier.hip)
#pragma once
#include "suhmo_common.h"
struct suhmo_hier;
int suhmo_hier_nlev_(const suhmo_hier *H);
const std::vector<suhmo_level *> &suhmo_hier_boxes_(suhmo_hier *H, int l);
int suhmo_hier_device_(const suhmo_hier *H);
int suhmo_hier_ff_(suhmo_hier *H, int l, int f0, int f1, bool corners, hipStream_t st);      // Copier::exchange between the boxes of a level
int suhmo_hier_cf_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                    // QuadCFInterp from level l-1
int suhmo_hier_cf2_(suhmo_hier *H, int l, int ff0, int fc0, int ff1, int fc1, hipStream_t st);   // two fields over the same stencils, one launch
int suhmo_hier_pwl_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                   // PiecewiseLinearFillPatch from level l-1
int suhmo_hier_avg_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                   // CoarseAverage into level l-1
// the hierarchy of SolveForGap_nl: the same boxes, alpha = 1, beta = dt diffFactor, Neumann-0 sides, no nonlinear term
int suhmo_hier_gap_(suhmo_hier *H, const suhmo_model_params_t *mp, double dt, suhmo_hier **gap);

// every box of a level in ONE launch (blockIdx.z = box): device tables of the boxes' views and field pointers
constexpr int SUHMO_BOX_HALO = 8;    // cells around a box the plan `halo` of a level covers (k_gsrb_box_m advances through up to that many: 4 sweeps per launch)
struct suhmo_multi { const DV *dv; const FP *fp; int nbox, maxnx, maxny; double *red; /* reduction scratch, 64 nbox + 16 doubles */
                     int merged; /* hierarchy option merged_launches: gradient + its ghosts, Re + bCoef in one launch each */
                     const void *push; const int *pbase; /* fine-fine ghost cells a side cell feeds (int2 {box, offset}), first entry of every box */ };
int suhmo_multi_colour_pass(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, int pass, hipStream_t st, bool push = false);      // suhmo_gsrb.hip
// suhmo_gsrb.hip: 2 sweeps per launch (bc_ghosts: + the closing homogeneous ghost fill)
int suhmo_multi_gsrb_box(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, const void *halo, const int *hbase, int fsrc, int fdst, int npass,
    int bc_ghosts, hipStream_t st);
int suhmo_multi_fill_ghosts(const suhmo_multi &m, int field, int homog, hipStream_t st);                                  // suhmo_ops.hip ...
int suhmo_multi_apply(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, int mode, hipStream_t st);            // mode 0: LPHI, 1: RES, 3: both
int suhmo_apply_and_residual(suhmo_level *L, int depth, hipStream_t st);                                                  // LPHI and RES = rhs - LPHI in one pass
int suhmo_multi_grad_cc(const suhmo_multi &m, int hasMask, hipStream_t st);
int suhmo_multi_re(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st);
int suhmo_multi_bcoef_faces(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st);
int suhmo_multi_re_bcoef(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st);          // the two above, one launch when m.merged
int suhmo_multi_coef_ghosts(const suhmo_multi &m, int field, hipStream_t st);
int suhmo_multi_axby(const suhmo_multi &m, int fd, int fx, int fy, double a, double b, hipStream_t st);
int suhmo_multi_fas_enter(const suhmo_multi &m, hipStream_t st);   // RHS0 <- RHS, RHS <- RES + LPHI, PHIOLD <- PHI in one launch
int suhmo_multi_fas_leave(const suhmo_multi &m, hipStream_t st);   // RHS <- RHS0, CORR <- PHI - PHIOLD in one launch
int suhmo_multi_copy(const suhmo_multi &m, int fd, int fs, hipStream_t st);                                               // valid cells + ghost ring
int suhmo_multi_copy_between(const suhmo_multi &dst, const suhmo_multi &src, const int *fd, const int *fs, int n, hipStream_t st);   // same boxes, two hierarchies
int suhmo_multi_norm_max(const suhmo_multi &m, suhmo_level *slot, int field, double *out, hipStream_t st);                // max |x| over the valid cells of all boxes
// the same in pieces, for one read-back over a whole hierarchy: first stages (partial maxima of level 0 / of a level of boxes), then one launch over all lists
int suhmo_level_norm_max_partials(suhmo_level *L, int field, const double **partials, int *np, hipStream_t st);
int suhmo_multi_norm_max_partials(const suhmo_multi &m, int field, const double **partials, int *np, hipStream_t st);
int suhmo_norm_max_of_lists(suhmo_level *slot, const double *const *partials, const int *np, int cnt, double *out, hipStream_t st);
// SEVERAL levels of boxes in one launch (blockIdx.z runs over the boxes of the listed levels, one after the other): by value, indexed with
// constants only (an unrolled search), so that the tables stay in scalar registers
constexpr int SUHMO_LVMAX = 7;
struct suhmo_lvboxes { const DV *dv[SUHMO_LVMAX]; const FP *fp[SUHMO_LVMAX]; int nbox[SUHMO_LVMAX], mode[SUHMO_LVMAX]; int n, maxnx, maxny; };
int suhmo_levels_apply(const suhmo_lvboxes &lv, const suhmo_phys_t &ph, bool has_alpha, hipStream_t st);   // mode[q]: 1 RES = rhs - L(phi), 3: LPHI as well (suhmo_multi_apply)
// field <- 0 where SUHMO_F_COVER is set, and the first stage of max |field| over the rest, of the listed levels / of a whole level; one partial per workgroup
int suhmo_levels_norm_max_cover_partials(const suhmo_lvboxes &lv, int field, double *partial, int *np, hipStream_t st);
int suhmo_level_norm_max_cover_partials(suhmo_level *L, int field, const double **partials, int *np, hipStream_t st);
int suhmo_levels_grad_cc(const suhmo_lvboxes &lv, int hasMask, hipStream_t st);                              // suhmo_multi_grad_cc (merged form) of several levels
int suhmo_hier_multi_(suhmo_hier *H, int l, hipStream_t st, suhmo_multi *m);       // l >= 1
int suhmo_hier_ensure_(suhmo_hier *H, int l, int field);                            // allocate a field on every box of a level
void suhmo_hier_invalidate_(suhmo_hier *H);                                         // an entry point outside suhmo_hier.hip: the caller may have loaded new data
const double *suhmo_hier_base_cover_(suhmo_hier *H, DV *whole);
// owner computes (levels >= 1 dealt to the ranks): is it on; the boxes of level l this rank owns (everything suhmo_hier_multi_ launches on);
// MAX over the ranks of a value each computed on its own boxes; the all-gather of the hierarchy (device buffers, `count` doubles per rank)
bool suhmo_hier_partitioned_(const suhmo_hier *H);
void suhmo_hier_owned_(const suhmo_hier *H, int l, int *first, int *n);
int suhmo_hier_allreduce_max_(suhmo_hier *H, double *v);
int suhmo_hier_allgather_(suhmo_hier *H, const double *send, long count, double *recv, hipStream_t st);
int suhmo_hier_world_(const suhmo_hier *H);                     // level 0 cut into rank strips: COVER of the WHOLE level and its view; else NULL
