// suhmo_hier.h -- what suhmo_step.hip needs from a hierarchy of box unions (suhmo_hier.hip)
#pragma once
#include "suhmo_common.h"
struct suhmo_hier;
int suhmo_hier_nlev_(const suhmo_hier *H);
const std::vector<suhmo_level *> &suhmo_hier_boxes_(suhmo_hier *H, int l);
int suhmo_hier_device_(const suhmo_hier *H);
int suhmo_hier_ff_(suhmo_hier *H, int l, int f0, int f1, bool corners, hipStream_t st);      // Copier::exchange between the boxes of a level
int suhmo_hier_cf_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                    // QuadCFInterp from level l-1
int suhmo_hier_pwl_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                   // PiecewiseLinearFillPatch from level l-1
int suhmo_hier_avg_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st);                   // CoarseAverage into level l-1
// the hierarchy of SolveForGap_nl: the same boxes, alpha = 1, beta = dt diffFactor, Neumann-0 sides, no nonlinear term
int suhmo_hier_gap_(suhmo_hier *H, const suhmo_model_params_t *mp, double dt, suhmo_hier **gap);
