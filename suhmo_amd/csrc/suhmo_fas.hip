// suhmo_fas.hip -- FAS multigrid V-cycle and solve loop on the device-resident level.
//
// Stands in for Chombo's AMRFASMultiGrid<LevelData<FArrayBox>>::solve as driven by
// AmrHydro::SolveForHead_nl (src/AmrHydro.cpp:665-769).  The cycle driver itself lives in
// the un-vendored Chombo fork; the ordering below is the reconstruction documented in
// SURVEY.md Appendix D (UNPINNED) from the reference's operator overrides:
//   UpdateOperator / AverageOperator   src/VCAMRNonLinearPoissonOp.cpp:32-95
//   relax                              src/AMRNonLinearPoissonOp.cpp:707-750
//   restrictResidual / restrictR       src/VCAMRNonLinearPoissonOp.cpp:347-460
//   applyOpMg                          src/VCAMRNonLinearPoissonOp.cpp:211-231
//   prolongIncrement                   src/AMRNonLinearPoissonOp.cpp:856-886
// Single AMR level in this round; all work stays on `st`, the only host round trip is the
// 8-byte residual norm per V-cycle in the solve loop.
#include "suhmo_common.h"

static int eff_depths(const suhmo_level *L, const suhmo_solver_params_t *sp)
{
    int nd = L->ndepth;
    if (sp->max_depth >= 0 && sp->max_depth + 1 < nd) nd = sp->max_depth + 1;
    return nd;
}

// relax() of the operator: levelGSRB sweeps, then the homogeneous-BC ghost fill levelGSRB ends with
// (src/VCAMRNonLinearPoissonOp.cpp:757-759).  `tail`: halo rows (strips) worth keeping valid for the next reader.
static int relax(suhmo_level *L, int dep, int sweeps, int tail, suhmo_stream_t s)
{
    int rc = suhmo_launch_gsrb(L, dep, sweeps, tail, (hipStream_t)s);
    if (rc) return rc;
    if (sweeps > 0) return suhmo_level_fill_ghosts(L, dep, SUHMO_F_PHI, 1, s);
    return 0;
}

// Strips: halo rows of phi a depth wants valid right after the prolongation (= through its post-smoothing plus what
// the next reader takes: UpdateOperator of the next cycle reads 2 rows at depth 0, the parent's prolongation reads
// half of what the parent wants).  Purely an exchange-saving hint: the relax kernels advance that many halo rows
// redundantly when they have them, and Depth::phi_fresh keeps every stencil honest whatever the halo depth.
static int rows_after_prolong(int dep, int num_smooth)
{
    int tail_post = 2;
    for (int d = 1; d <= dep; d++) tail_post = (2 * num_smooth + tail_post + 1) / 2;
    return 2 * num_smooth + tail_post;
}
int suhmo_prolong_with_halo(suhmo_level *L, int depth, hipStream_t st);     // suhmo_level.hip

static int fas_cycle(suhmo_level *L, int dep, const suhmo_solver_params_t *sp, int nd, suhmo_stream_t s)
{
    int rc;
    const int S = sp->num_smooth;
    const int tail_post = dep == 0 ? 2 : (rows_after_prolong(dep - 1, S) + 1) / 2;
    if (dep == nd - 1) return relax(L, dep, sp->num_bottom, tail_post, s);        // bottom relaxes
    Depth &C = L->d[dep + 1];
    if ((rc = relax(L, dep, S, rows_after_prolong(dep, S), s))) return rc;        // pre-smooth (the restriction reads 1 halo row,
                                                                                  //  the prolongation below the rest)
    if ((rc = suhmo_restrict_both(L, dep, (hipStream_t)s))) return rc;            // RES[dep+1] and PHI[dep+1] = R(phi)
    if ((rc = suhmo_ensure_phi_halo(L, dep + 1, suhmo_halo_rows(C.v), (hipStream_t)s))) return rc;   // strips: before the copy, so that
    HIPCHK(hipMemcpyAsync(C.fp.f[SUHMO_F_PHIOLD], C.fp.f[SUHMO_F_PHI], C.elems * sizeof(double),      // PHIOLD carries the halo rows too
                          hipMemcpyDeviceToDevice, (hipStream_t)s));
    if ((rc = suhmo_level_apply_op(L, dep + 1, 0, s))) return rc;                 // LPHI = L_c(R phi)
    if ((rc = suhmo_level_axby(L, dep + 1, SUHMO_F_RHS, SUHMO_F_RES, SUHMO_F_LPHI, 1.0, 1.0, s))) return rc;
    if ((rc = suhmo_level_exchange(L, dep + 1, SUHMO_F_RHS, s))) return rc;   // strips: rhs halo rows for the fused relax
    if ((rc = fas_cycle(L, dep + 1, sp, nd, s))) return rc;
    if (suhmo_gsrb_can_fuse_prolong(L, dep, S)) {
        L->d[dep].prolong_pending = 1;      // phi += P(phi_c - phi_c,old) happens inside the first post-smoothing pass
    } else {
        if ((rc = suhmo_prolong_with_halo(L, dep, (hipStream_t)s))) return rc;    // corr = phi_c - phi_c,old; phi += P(corr), halo rows included
    }
    return relax(L, dep, S, tail_post, s);                                        // post-smooth
}

extern "C" int suhmo_level_vcycle(suhmo_level_t *L, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    ARG(L && sp);
    HIPCHK(hipSetDevice(L->device));
    int nd = eff_depths(L, sp), rc;
    if (sp->bcoeff_otf) {
        if ((rc = suhmo_level_update_operator(L, 0, s))) return rc;
        if ((rc = suhmo_average_operator_all(L, nd, (hipStream_t)s))) return rc;   // AverageOperator on every depth > 0
    }
    return fas_cycle(L, 0, sp, nd, s);
}

extern "C" int suhmo_level_solve(suhmo_level_t *L, const suhmo_solver_params_t *sp, int *iters, double *hist, suhmo_stream_t s)
{
    ARG(L && sp);
    HIPCHK(hipSetDevice(L->device));
    int rc;
    double rnorm = 0.0;
    if ((rc = suhmo_level_residual(L, 0, s))) return rc;
    if ((rc = suhmo_level_norm(L, 0, SUHMO_F_RES, 0, &rnorm, s))) return rc;
    double initial_rnorm = rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    bool goNorm = rnorm > sp->norm_thresh;
    bool goRedu = rnorm > sp->eps * initial_rnorm;
    bool goIter = iter < sp->max_iter;
    bool goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last;
    bool goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        if ((rc = suhmo_level_vcycle(L, sp, s))) return rc;
        if ((rc = suhmo_level_residual(L, 0, s))) return rc;
        if ((rc = suhmo_level_norm(L, 0, SUHMO_F_RES, 0, &rnorm, s))) return rc;
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh;
        goRedu = rnorm > sp->eps * initial_rnorm;
        goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last;
        goMin = iter < sp->iter_min;
    }
    if (iters) *iters = iter;
    return 0;
}
