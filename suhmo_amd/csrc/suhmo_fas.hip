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
#include <cstdlib>

static int eff_depths(const suhmo_level *L, const suhmo_solver_params_t *sp)
{
    int nd = L->ndepth;
    if (sp->max_depth >= 0 && sp->max_depth + 1 < nd) nd = sp->max_depth + 1;
    return nd;
}

// relax() of the operator: levelGSRB sweeps, then the homogeneous-BC ghost fill levelGSRB ends with
// (src/VCAMRNonLinearPoissonOp.cpp:757-759).  `tail`: halo rows (strips) worth keeping valid for the next reader.
// Inside the cycle nothing reads the stored ghost ring (every kernel evaluates the boundary condition from the
// adjacent interior value), so only the relax that ends the cycle -- whose state the caller can observe -- fills it.
static int relax(suhmo_level *L, int dep, int sweeps, int tail, suhmo_stream_t s, bool observable = false, int *restricted = nullptr)
{
    // the relax that ends the cycle (`observable`: depth 0, nothing follows) may leave the residual of the final phi behind for the solve loop
    L->resout_armed = observable && dep == 0 && L->resout_req && L->resid_in_relax;
    int rc = suhmo_launch_gsrb(L, dep, sweeps, tail, (hipStream_t)s, restricted);
    L->resout_armed = 0;
    if (rc) return rc;
    if (sweeps > 0 && observable) return suhmo_level_fill_ghosts(L, dep, SUHMO_F_PHI, 1, s);
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
int suhmo_prolong_with_halo(suhmo_level *L, int depth, hipStream_t st);     // suhmo_ops.hip

static int fas_cycle(suhmo_level *L, int dep, const suhmo_solver_params_t *sp, int nd, suhmo_stream_t s)
{
    int rc;
    const int S = sp->num_smooth;
    const int tail_post = dep == 0 ? 2 : (rows_after_prolong(dep - 1, S) + 1) / 2;
    if (dep == nd - 1) return relax(L, dep, sp->num_bottom, tail_post, s, dep == 0);   // bottom relaxes
    Depth &C = L->d[dep + 1];
    int restricted = 0;
    if ((rc = relax(L, dep, S, rows_after_prolong(dep, S), s, false, &restricted))) return rc;   // pre-smooth (the restriction reads 1 halo
                                                                                  //  row, the prolongation below the rest)
    if (!restricted && (rc = suhmo_restrict_both(L, dep, (hipStream_t)s))) return rc;   // RES[dep+1] and PHI[dep+1] = R(phi); the fused
                                                                                  //  relaxation may have written them already
    if (L->agg && dep + 1 == L->agg_depth) {
        // rank strips, agglomerated depths (suhmo_agg.hip): R phi and RES of every rank's rows -> the whole-level copy A every rank
        // holds; A forms its right-hand side and runs the rest of the cycle without a message; this rank's rows and halo rows of
        // phi and R phi come back, so the prolongation below needs no exchange either
        suhmo_level *A = L->agg;
        const int ndA = nd - L->agg_depth;
        // A created after the coefficients were built (agg_min_cells set, or the transport attached, later): its copies of them first
        if (L->agg_static_stale && (rc = suhmo_agg_gather_static(L, true, (hipStream_t)s))) return rc;
        if ((rc = suhmo_agg_gather_state(L, (hipStream_t)s))) return rc;
        const int next_sweeps = ndA == 1 ? sp->num_bottom : S;
        if (suhmo_gsrb_can_fuse_rhs(A, 0, next_sweeps)) A->d[0].rhs_pending = 1;
        else if ((rc = suhmo_fas_coarse_rhs(A, 0, (hipStream_t)s))) return rc;
        if ((rc = fas_cycle(A, 0, sp, ndA, s))) return rc;
        if ((rc = suhmo_agg_scatter(L, (hipStream_t)s))) return rc;
        if (suhmo_gsrb_can_fuse_prolong(L, dep, S)) L->d[dep].prolong_pending = 1;
        else if ((rc = suhmo_prolong_with_halo(L, dep, (hipStream_t)s))) return rc;
        return relax(L, dep, S, tail_post, s, dep == 0);
    }
    const bool one_pass_rhs = L->fas_rhs_fused != 0;
    // rank strips: R phi and RES travel in one message group and the right-hand side of the halo rows is computed here from them,
    // bit for bit what the neighbour computes for its own rows: the exchange of RHS disappears
    const bool rhs_local = L->ex && (C.v.rk[0] || C.v.rk[1]) && one_pass_rhs && L->desc.nx_global == 0 && L->strips_rhs_local
                           && suhmo_halo_rows(C.v) >= 2;
    if (rhs_local) {
        static const int both[2] = {SUHMO_F_PHI, SUHMO_F_RES};
        if ((rc = suhmo_exchange_list(L, dep + 1, both, 2, (hipStream_t)s))) return rc;
    } else if ((rc = suhmo_ensure_phi_halo(L, dep + 1, suhmo_halo_rows(C.v), (hipStream_t)s))) return rc;   // strips: before PHIOLD is taken, so
                                                                                                             // that it carries the halo rows too
    if (one_pass_rhs && (L->desc.nx_global == 0)) {
        const int next_sweeps = (dep + 1 == nd - 1) ? sp->num_bottom : S;
        if (suhmo_gsrb_can_fuse_rhs(L, dep + 1, next_sweeps, rhs_local)) C.rhs_pending = 1;   // ... inside the first relaxation of the depth
        else if ((rc = suhmo_fas_coarse_rhs(L, dep + 1, (hipStream_t)s, rhs_local ? suhmo_halo_rows(C.v) - 1 : 0))) return rc;   // PHIOLD = R phi,
                                                                                   // rhs_c = res_c + L_c(R phi): one pass
    } else {
        HIPCHK(hipMemcpyAsync(C.fp.f[SUHMO_F_PHIOLD], C.fp.f[SUHMO_F_PHI], C.elems * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)s));
        if ((rc = suhmo_level_apply_op(L, dep + 1, 0, s))) return rc;             // LPHI = L_c(R phi)
        if ((rc = suhmo_level_axby(L, dep + 1, SUHMO_F_RHS, SUHMO_F_RES, SUHMO_F_LPHI, 1.0, 1.0, s))) return rc;
    }
    if (!rhs_local && (rc = suhmo_level_exchange(L, dep + 1, SUHMO_F_RHS, s))) return rc;   // strips: rhs halo rows for the fused relax
    if ((rc = fas_cycle(L, dep + 1, sp, nd, s))) return rc;
    if (suhmo_gsrb_can_fuse_prolong(L, dep, S)) {
        L->d[dep].prolong_pending = 1;      // phi += P(phi_c - phi_c,old) happens inside the first post-smoothing pass
    } else {
        if ((rc = suhmo_prolong_with_halo(L, dep, (hipStream_t)s))) return rc;    // corr = phi_c - phi_c,old; phi += P(corr), halo rows included
    }
    return relax(L, dep, S, tail_post, s, dep == 0);                              // post-smooth
}

static int vcycle_body(suhmo_level *L, const suhmo_solver_params_t *sp, int nd, suhmo_stream_t s)
{
    int rc;
    if (sp->bcoeff_otf) {
        L->faces_deferred = 1;              // rank strips: the halo rows of the depth-0 faces travel with those of the coarse depths, one message
        rc = suhmo_level_update_operator(L, 0, s);
        if (rc) { L->faces_deferred = 0; return rc; }
        if (L->mask_reported) L->maskflag_epoch = L->mask_epoch;                   // the relaxations of THIS cycle may rely on the report
        if ((rc = suhmo_average_operator_all(L, nd, (hipStream_t)s))) return rc;   // AverageOperator on every depth > 0
    } else L->maskflag_epoch = 0;
    rc = fas_cycle(L, 0, sp, nd, s);
    L->maskflag_epoch = 0;                  // the report on the ice mask (k_bcoef_fused) holds for this cycle only: the caller may load another mask
    return rc;
}

// Small levels are launch-bound (a 1024^2 V-cycle is ~170 dependent launches of a few microseconds each): the cycle is a
// fixed sequence of kernels for given solver parameters, so from its second use it is replayed as a HIP graph.  The first
// use runs eagerly (lazy allocations, occupancy queries), later ones are captured on a private stream, one graph per state of
// the phi ping-pong pointers the cycle starts from (a depth with an odd number of out-of-place relaxation launches ends
// with its two canvases swapped: the next cycle is a different graph); the pointers the capture ended with are re-applied
// after every replay.  Not used on rank strips (the exchange hooks are host calls), while profiling, or above SUHMO_GRAPH_MAX_CELLS.
void suhmo_level_drop_graphs(suhmo_level *L)
{
    if (getenv("SUHMO_GRAPH_DEBUG")) { int ok = 0; for (VGraph &g : L->vgraphs) ok += g.exec != nullptr; fprintf(stderr,
        "[suhmo] level %dx%d: %zu V-cycle graphs, %d executable\n", L->d[0].v.nx, L->d[0].v.ny, L->vgraphs.size(), ok); }
    for (VGraph &g : L->vgraphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    L->vgraphs.clear();
    if (L->gstream) { (void)hipStreamDestroy(L->gstream); L->gstream = nullptr; }
}
static int vcycle_graph(suhmo_level *L, const suhmo_solver_params_t *sp, int nd, suhmo_stream_t s, bool &done)
{
    done = false;
    const int key[4] = {sp->num_smooth, sp->num_bottom, sp->bcoeff_otf, nd};
    auto at_start = [&](const VGraph &g) {
        if (memcmp(g.key, key, sizeof(key))) return false;
        for (int d = 0; d < L->ndepth; d++) if (L->d[d].fp.f[SUHMO_F_PHI] != g.p0[d] || L->d[d].phi_alt != g.a0[d]) return false;
        if (g.rout_req != (L->resid_in_relax ? L->resout_req : 0) || g.rout_rhs != L->resout_rhs) return false;   // (what the last launch leaves behind)
        return L->d[0].fp.f[SUHMO_F_RHS] == g.rhs;                       // an AMR cycle runs level 0 on a second right-hand-side canvas (suhmo_hier.hip)
    };
    for (const VGraph &g : L->vgraphs)
        if (at_start(g)) {
            if (!g.exec) return 0;                                       // known not to be capturable
            HIPCHK(hipGraphLaunch(g.exec, (hipStream_t)s));
            for (int d = 0; d < L->ndepth; d++) { L->d[d].fp.f[SUHMO_F_PHI] = g.p1[d]; L->d[d].phi_alt = g.a1[d]; } suhmo_fp_changed();
            if (g.rout_done) { L->resout_done = 1; L->resout_count++; L->resout_np = g.rout_np; }
            done = true;
            return 0;
        }
    // first sighting of these parameters: run eagerly now (lazy allocations, occupancy queries), capture from the next call on
    for (int k = 0; k < 4; k++) if (L->vgraph_seen[k] != key[k]) { memcpy(L->vgraph_seen, key, sizeof(key)); return 0; }
    if (L->vgraphs.size() >= 16) return 0;                               // (pointer states keep changing: give up capturing)
    if (!L->gstream) HIPCHK(hipStreamCreateWithFlags(&L->gstream, hipStreamNonBlocking));
    VGraph g; memcpy(g.key, key, sizeof(key)); g.exec = nullptr;
    for (int d = 0; d < SUHMO_MAXDEPTH; d++) { g.p0[d] = g.a0[d] = g.p1[d] = g.a1[d] = nullptr; }
    for (int d = 0; d < L->ndepth; d++) { g.p0[d] = L->d[d].fp.f[SUHMO_F_PHI]; g.a0[d] = L->d[d].phi_alt; }
    g.rhs = L->d[0].fp.f[SUHMO_F_RHS];
    g.rout_req = L->resid_in_relax ? L->resout_req : 0; g.rout_rhs = L->resout_rhs; g.rout_done = 0;
    const int rout_before = L->resout_done; const long rout_count = L->resout_count;
    HIPCHK(hipStreamSynchronize((hipStream_t)s));                        // the private stream starts from a quiescent state
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(L->gstream, hipStreamCaptureModeThreadLocal);
    int rc = 0;
    if (e == hipSuccess) {
        rc = vcycle_body(L, sp, nd, (suhmo_stream_t)L->gstream);
        e = hipStreamEndCapture(L->gstream, &graph);
    }
    g.rout_done = L->resout_count != rout_count;                         // (nothing was executed during capture: the flags go back)
    g.rout_np = L->resout_np;
    L->resout_done = rout_before; L->resout_count = rout_count;
    bool clean = true;                                                   // host-side state the cycle must leave behind: none pending
    for (int d = 0; d < L->ndepth; d++) {
        clean = clean && !L->d[d].prolong_pending && !L->d[d].rhs_pending;
        g.p1[d] = L->d[d].fp.f[SUHMO_F_PHI]; g.a1[d] = L->d[d].phi_alt;
        L->d[d].prolong_pending = 0; L->d[d].rhs_pending = 0;
        L->d[d].fp.f[SUHMO_F_PHI] = g.p0[d]; L->d[d].phi_alt = g.a0[d];  // nothing was executed during capture
    }
    if (e == hipSuccess && rc == 0 && clean && graph && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) g.exec = nullptr;
    if (!(e == hipSuccess && rc == 0 && clean)) g.exec = nullptr;
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    L->vgraphs.push_back(g);
    if (!g.exec) return 0;                                               // run eagerly
    HIPCHK(hipGraphLaunch(g.exec, (hipStream_t)s));
    for (int d = 0; d < L->ndepth; d++) { L->d[d].fp.f[SUHMO_F_PHI] = g.p1[d]; L->d[d].phi_alt = g.a1[d]; } suhmo_fp_changed();
    if (g.rout_done) { L->resout_done = 1; L->resout_count++; L->resout_np = g.rout_np; }
    done = true;
    return 0;
}

extern "C" int suhmo_level_vcycle(suhmo_level_t *L, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    SUHMO_TIME("AMRFASMultiGrid::VCycle");
    ARG(L && sp);
    HIPCHK(hipSetDevice(L->device));
    int nd = eff_depths(L, sp);
    const Depth &D = L->d[0];
    const bool ext = D.v.ext[0] || D.v.ext[1];
    if (L->graph_max_cells > 0 && !L->ex && !ext && !L->prof_on && (long)D.v.nx * D.v.ny <= L->graph_max_cells) {
        bool done = false;
        int rc = vcycle_graph(L, sp, nd, s, done);
        if (rc || done) return rc;
    }
    return vcycle_body(L, sp, nd, s);
}

extern "C" int suhmo_level_solve(suhmo_level_t *L, const suhmo_solver_params_t *sp, int *iters, double *hist, suhmo_stream_t s)
{
    SUHMO_TIME("AMRFASMultiGrid::solve");
    ARG(L && sp);
    HIPCHK(hipSetDevice(L->device));
    int rc;
    double rnorm = 0.0;
    if ((rc = suhmo_level_residual_and_norm(L, &rnorm, (hipStream_t)s))) return rc;
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
        // (the launch that ends the cycle leaves rhs - L(phi) of the final phi in RES when it can: the pass below is then not needed)
        L->resout_req = 1 | 4; L->resout_rhs = nullptr; L->resout_done = 0; L->resout_np = 0;      // (4: and the partial maxima of its max norm)
        rc = suhmo_level_vcycle(L, sp, s);
        const bool have_res = L->resout_done != 0;
        const int np = have_res ? L->resout_np : 0;
        L->resout_req = 0; L->resout_done = 0; L->resout_np = 0;
        if (rc) return rc;
        if (!have_res) rc = suhmo_level_residual_and_norm(L, &rnorm, (hipStream_t)s);        // (two launches: the pass leaves the partial maxima too)
        else if (np > 0) rc = suhmo_level_norm_from_partials(L, np, &rnorm, (hipStream_t)s);
        else rc = suhmo_level_norm(L, 0, SUHMO_F_RES, 0, &rnorm, s);
        if (rc) return rc;
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
