// VCAMRNonLinearPoissonOpHIP.cpp -- see the header.  Plain C++ (g++), links libsuhmo_hip.so.
#include "VCAMRNonLinearPoissonOpHIP.H"
#include <algorithm>
#include <cmath>
#include <string>

namespace suhmo_host {

static void chk(int rc, const char *what)
{
    if (rc != 0) {
        std::string m = std::string(what) + ": " + suhmo_last_error();
        MayDay::Error(m.c_str());       // the reference's error channel (process abort)
    }
}

HeadSolverParameters::HeadSolverParameters(int a_cur_step, bool a_bcoeff_otf)
{
    num_smooth = 4; num_bottom = 16; max_iter = 100; iter_min = 2; imin = 5;      // AmrHydro.cpp:737-743,762
    eps = 1.0e-7; hang = 0.01; norm_thresh = 1.0e-7;
    if (a_cur_step < 50) { num_bottom = 10; eps = 1.0e-10; hang = 0.0001; imin = 20; }  // :744-754
    bcoeff_otf = a_bcoeff_otf ? 1 : 0; max_depth = -1;
}

VCAMRNonLinearPoissonOpHIPFactory::VCAMRNonLinearPoissonOpHIPFactory() : m_level(nullptr), m_update_operator(true) {}
VCAMRNonLinearPoissonOpHIPFactory::~VCAMRNonLinearPoissonOpHIPFactory()
{
    if (m_hier) suhmo_hier_destroy(m_hier);
    if (m_fine) suhmo_level_destroy(m_fine);
    if (m_level) suhmo_level_destroy(m_level);
}

// a LevelData of one AMR level of the hierarchy <-> the device: level 0 box by box into the base handle, a level >= 1 box k into
// the level handle of that box (ghost cells included: a box of a union keeps its own ghost ring)
static void hier_io(suhmo_hier_t *H, int l, int field, LevelData<FArrayBox> &ld, bool to_device)
{
    const DisjointBoxLayout &g = ld.disjointBoxLayout();
    for (int k = 0; k < g.size(); k++) {
        FArrayBox &f = ld[k];
        const Box &fb = f.box();
        if (l == 0) {
            suhmo_level_t *B = suhmo_hier_box(H, 0, 0);
            if (to_device) chk(suhmo_level_put_box(B, 0, field, k, f.dataPtr(), fb.lo[0], fb.lo[1], fb.hi[0], fb.hi[1], ld.ghost() > 0, nullptr), "hierarchy: put_box");
            else chk(suhmo_level_get_box(B, 0, field, k, f.dataPtr(), fb.lo[0], fb.lo[1], fb.hi[0], fb.hi[1], nullptr), "hierarchy: get_box");
        } else {
            suhmo_level_t *B = suhmo_hier_box(H, l, k);
            if (!B || ld.ghost() > 1) MayDay::Error("hierarchy: box / ghost width");
            if (to_device) chk(suhmo_level_set_field(B, 0, field, f.dataPtr(), ld.ghost() > 0, 0, nullptr), "hierarchy: set_field");
            else chk(suhmo_level_get_field(B, 0, field, f.dataPtr(), ld.ghost() > 0, 0, nullptr), "hierarchy: get_field");
        }
    }
}

void VCAMRNonLinearPoissonOpHIPFactory::defineHierarchy(const ProblemDomain &a_dom, const std::vector<DisjointBoxLayout> &a_grids, const RealVect &a_dx,
                                                        const suhmo_bc_t &a_bc, const Real &a_alpha, const Real &a_beta, const suhmo_phys_t &a_phys,
                                                        const std::vector<LevelData<FArrayBox> *> &a_aCoef, const std::vector<LevelData<FArrayBox> *> &a_B,
                                                        const std::vector<LevelData<FArrayBox> *> &a_Pi, const std::vector<LevelData<FArrayBox> *> &a_zb,
                                                        const std::vector<LevelData<FArrayBox> *> &a_iceMask, int a_device)
{
    const int nlev = (int)a_grids.size();
    if (nlev < 1 || (int)a_aCoef.size() != nlev || (int)a_B.size() != nlev || (int)a_Pi.size() != nlev || (int)a_zb.size() != nlev || (int)a_iceMask.size() != nlev)
        MayDay::Error("defineHierarchy: one layout and one set of coefficients per level");
    if (m_hier) { suhmo_hier_destroy(m_hier); m_hier = nullptr; }
    suhmo_level_desc_t d = {};
    d.nx = a_dom.dom.size(0); d.ny = a_dom.dom.size(1); d.j0 = 0; d.ny_global = d.ny;
    d.dx = a_dx[0]; d.dy = a_dx[1];
    std::vector<int> base;
    for (int k = 0; k < a_grids[0].size(); k++) { const Box &b = a_grids[0][k]; base.insert(base.end(), {b.lo[0], b.lo[1], b.hi[0], b.hi[1]}); }
    d.nbox = a_grids[0].size(); d.boxes = base.data(); d.max_box = 0;
    d.alpha = a_alpha; d.beta = a_beta; d.bc = a_bc; d.phys = a_phys; d.device = a_device; d.halo_rows = 1;
    std::vector<int> nbox(nlev, 0), boxes;
    for (int l = 1; l < nlev; l++) {
        nbox[l] = a_grids[l].size();
        for (int k = 0; k < a_grids[l].size(); k++) { const Box &b = a_grids[l][k]; boxes.insert(boxes.end(), {b.lo[0], b.lo[1], b.hi[0], b.hi[1]}); }
    }
    chk(suhmo_hier_create(&m_hier, &d, nlev, nbox.data(), boxes.data()), "VCAMRNonLinearPoissonOpHIPFactory::defineHierarchy");
    m_hierGrids = a_grids;
    for (int l = 0; l < nlev; l++) {
        hier_io(m_hier, l, SUHMO_F_ACOEF, *a_aCoef[l], true); hier_io(m_hier, l, SUHMO_F_B, *a_B[l], true); hier_io(m_hier, l, SUHMO_F_PI, *a_Pi[l], true);
        hier_io(m_hier, l, SUHMO_F_ZB, *a_zb[l], true); hier_io(m_hier, l, SUHMO_F_MASK, *a_iceMask[l], true);
    }
    chk(suhmo_level_build_mg_coefficients(suhmo_hier_box(m_hier, 0, 0), nullptr), "MGnewOp coefficient coarsening");
}

int VCAMRNonLinearPoissonOpHIPFactory::solveHierarchy(std::vector<LevelData<FArrayBox> *> &a_phi, const std::vector<LevelData<FArrayBox> *> &a_rhs,
                                                      const HeadSolverParameters &a_sp, std::vector<Real> *a_hist)
{
    if (!m_hier || a_phi.size() != m_hierGrids.size() || a_rhs.size() != m_hierGrids.size()) MayDay::Error("solveHierarchy: defineHierarchy first, one phi and rhs per level");
    for (size_t l = 0; l < a_phi.size(); l++) {
        hier_io(m_hier, (int)l, SUHMO_F_PHI, *a_phi[l], true);
        hier_io(m_hier, (int)l, SUHMO_F_RHS, *const_cast<LevelData<FArrayBox> *>(a_rhs[l]), true);
    }
    std::vector<Real> hist(a_sp.max_iter + 2, 0.0);
    int iters = 0;
    chk(suhmo_hier_solve(m_hier, &a_sp, &iters, hist.data(), nullptr), "AMRFASMultiGrid::solve (hierarchy of box unions)");
    for (size_t l = 0; l < a_phi.size(); l++) hier_io(m_hier, (int)l, SUHMO_F_PHI, *a_phi[l], false);
    if (a_hist) a_hist->assign(hist.begin(), hist.begin() + iters + 1);
    return iters;
}

void VCAMRNonLinearPoissonOpHIPFactory::define(const ProblemDomain &a_dom, const DisjointBoxLayout &a_grids,
                                               const RealVect &a_dx, const suhmo_bc_t &a_bc, const Real &a_alpha,
                                               const LevelData<FArrayBox> &a_aCoef, const Real &a_beta,
                                               const LevelData<FluxBox> &a_bCoef, const suhmo_phys_t &a_phys,
                                               const LevelData<FArrayBox> &a_B, const LevelData<FArrayBox> &a_Pi,
                                               const LevelData<FArrayBox> &a_zb, const LevelData<FArrayBox> &a_iceMask,
                                               bool a_update_operator, int a_device)
{
    if (m_level) { suhmo_level_destroy(m_level); m_level = nullptr; m_ops.clear(); m_grids.clear(); }
    suhmo_level_desc_t d = {};                 // i0 = nx_global = 0: the level spans the domain in x
    d.nx = a_dom.dom.size(0); d.ny = a_dom.dom.size(1); d.j0 = 0; d.ny_global = d.ny;
    d.dx = a_dx[0]; d.dy = a_dx[1];
    std::vector<int> boxes;
    for (int k = 0; k < a_grids.size(); k++) { const Box &b = a_grids[k]; boxes.insert(boxes.end(), {b.lo[0], b.lo[1], b.hi[0], b.hi[1]}); }
    d.nbox = a_grids.size(); d.boxes = boxes.data(); d.max_box = 0;
    d.alpha = a_alpha; d.beta = a_beta; d.bc = a_bc; d.phys = a_phys; d.device = a_device; d.halo_rows = 1;
    chk(suhmo_level_create(&m_level, &d), "VCAMRNonLinearPoissonOpHIPFactory::define");
    m_desc = d; m_desc.boxes = nullptr; m_desc.nbox = 0;
    m_update_operator = a_update_operator;
    int nd = suhmo_level_num_depths(m_level);
    for (int dep = 0; dep < nd; dep++) m_grids.push_back(dep == 0 ? a_grids : a_grids.coarsened(1 << dep));
    m_ops.resize(nd);
    for (int dep = 0; dep < nd; dep++) m_ops[dep].reset(new VCAMRNonLinearPoissonOpHIP(this, dep));
    VCAMRNonLinearPoissonOpHIP &op0 = *m_ops[0];
    op0.put(SUHMO_F_ACOEF, a_aCoef, 0);
    op0.put(SUHMO_F_B, a_B, 0, true); op0.put(SUHMO_F_PI, a_Pi, 0, true);
    op0.put(SUHMO_F_ZB, a_zb, 0, true); op0.put(SUHMO_F_MASK, a_iceMask, 0, true);
    for (int k = 0; k < a_bCoef.size(); k++)
        for (int dir = 0; dir < 2; dir++) {
            const FArrayBox &f = a_bCoef[k][dir];
            chk(suhmo_level_put_box(m_level, 0, dir == 0 ? SUHMO_F_BX : SUHMO_F_BY, k, f.dataPtr(), f.box().lo[0], f.box().lo[1],
                                    f.box().hi[0], f.box().hi[1], 0, nullptr), "put bCoef");
        }
    chk(suhmo_level_build_mg_coefficients(m_level, nullptr), "MGnewOp coefficient coarsening");
}

void VCAMRNonLinearPoissonOpHIPFactory::defineFineLevel(const ProblemDomain &a_fineDomain, const DisjointBoxLayout &a_fineGrids,
                                                        const LevelData<FArrayBox> &a_aCoef, const LevelData<FluxBox> &a_bCoef,
                                                        const LevelData<FArrayBox> &a_B, const LevelData<FArrayBox> &a_Pi,
                                                        const LevelData<FArrayBox> &a_zb, const LevelData<FArrayBox> &a_iceMask)
{
    if (!m_level) MayDay::Error("defineFineLevel before define");
    if (m_fine) { suhmo_level_destroy(m_fine); m_fine = nullptr; }
    Box bb = a_fineGrids[0];
    for (int k = 1; k < a_fineGrids.size(); k++)
        for (int d = 0; d < 2; d++) { bb.lo[d] = std::min(bb.lo[d], a_fineGrids[k].lo[d]); bb.hi[d] = std::max(bb.hi[d], a_fineGrids[k].hi[d]); }
    suhmo_level_desc_t d = m_desc;
    d.nx = bb.size(0); d.ny = bb.size(1); d.i0 = bb.lo[0]; d.j0 = bb.lo[1];
    d.nx_global = a_fineDomain.dom.size(0); d.ny_global = a_fineDomain.dom.size(1);
    d.dx = m_desc.dx / 2.0; d.dy = m_desc.dy / 2.0;                                  // refRatio 2
    std::vector<int> boxes;
    for (int k = 0; k < a_fineGrids.size(); k++) { const Box &b = a_fineGrids[k]; boxes.insert(boxes.end(), {b.lo[0], b.lo[1], b.hi[0], b.hi[1]}); }
    d.nbox = a_fineGrids.size(); d.boxes = boxes.data(); d.max_box = 0;
    chk(suhmo_level_create(&m_fine, &d), "VCAMRNonLinearPoissonOpHIPFactory::defineFineLevel");
    m_fineDomain = a_fineDomain;
    m_fineOp.reset(new VCAMRNonLinearPoissonOpHIP(this, 0, 1));
    VCAMRNonLinearPoissonOpHIP &op = *m_fineOp;
    op.put(SUHMO_F_ACOEF, a_aCoef, 0);
    op.put(SUHMO_F_B, a_B, 0, true); op.put(SUHMO_F_PI, a_Pi, 0, true); op.put(SUHMO_F_ZB, a_zb, 0, true); op.put(SUHMO_F_MASK, a_iceMask, 0, true);
    for (int k = 0; k < a_bCoef.size(); k++)
        for (int dir = 0; dir < 2; dir++) {
            const FArrayBox &f = a_bCoef[k][dir];
            chk(suhmo_level_put_box(m_fine, 0, dir == 0 ? SUHMO_F_BX : SUHMO_F_BY, k, f.dataPtr(), f.box().lo[0], f.box().lo[1],
                                    f.box().hi[0], f.box().hi[1], 0, nullptr), "put fine bCoef");
        }
}

int VCAMRNonLinearPoissonOpHIPFactory::solveAMR(std::vector<LevelData<FArrayBox> *> &a_phi, const std::vector<LevelData<FArrayBox> *> &a_rhs,
                                                const HeadSolverParameters &a_sp, std::vector<Real> *a_hist)
{
    if (!m_fine || a_phi.size() != 2 || a_rhs.size() != 2) MayDay::Error("solveAMR: two levels");
    m_ops[0]->put(SUHMO_F_PHI, *a_phi[0], 0); m_ops[0]->put(SUHMO_F_RHS, *a_rhs[0], 0);
    m_fineOp->put(SUHMO_F_PHI, *a_phi[1], 0); m_fineOp->put(SUHMO_F_RHS, *a_rhs[1], 0);
    std::vector<Real> hist(a_sp.max_iter + 2, 0.0);
    int iters = 0;
    chk(suhmo_amr2_solve(m_level, m_fine, &a_sp, &iters, hist.data(), nullptr), "AMRFASMultiGrid::solve (2 levels)");
    m_ops[0]->get(SUHMO_F_PHI, *a_phi[0], 0); m_fineOp->get(SUHMO_F_PHI, *a_phi[1], 0);
    if (a_hist) a_hist->assign(hist.begin(), hist.begin() + iters + 1);
    return iters;
}

int VCAMRNonLinearPoissonOpHIPFactory::refToFiner(const ProblemDomain &a_domain) const
{
    if (m_level && a_domain.dom.size(0) == m_desc.nx && a_domain.dom.size(1) == m_desc.ny_global) return m_fine ? 2 : 1;   // m_refRatios[0]
    if (m_fine && a_domain.dom.size(0) == m_fineDomain.dom.size(0) && a_domain.dom.size(1) == m_fineDomain.dom.size(1)) return 1;   // finest level
    MayDay::Abort("Domain not found in AMR hierarchy");
    return -1;
}

int VCAMRNonLinearPoissonOpHIPFactory::numDepths() const { return m_level ? suhmo_level_num_depths(m_level) : 0; }

VCAMRNonLinearPoissonOpHIP *VCAMRNonLinearPoissonOpHIPFactory::MGnewOp(const ProblemDomain &, int a_depth, bool)
{
    if (a_depth < 0 || a_depth >= numDepths()) return nullptr;      // !coarsenable(2^depth * s_maxCoarse)
    return m_ops[a_depth].get();
}

int VCAMRNonLinearPoissonOpHIPFactory::solve(LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_rhs,
                                             const HeadSolverParameters &a_sp, std::vector<Real> *a_hist)
{
    VCAMRNonLinearPoissonOpHIP &op = *m_ops[0];
    op.put(SUHMO_F_PHI, a_phi, 0);
    op.put(SUHMO_F_RHS, a_rhs, 0);
    std::vector<Real> hist(a_sp.max_iter + 2, 0.0);
    int iters = 0;
    chk(suhmo_level_solve(m_level, &a_sp, &iters, hist.data(), nullptr), "AMRFASMultiGrid::solve");
    op.get(SUHMO_F_PHI, a_phi, 0);
    if (a_hist) a_hist->assign(hist.begin(), hist.begin() + iters + 1);
    return iters;
}

// ---------------------------------------------------------------- operator
suhmo_level_t *VCAMRNonLinearPoissonOpHIP::h() const { return m_amrLevel == 1 ? m_factory->m_fine : m_factory->m_level; }

void VCAMRNonLinearPoissonOpHIP::put(int field, const LevelData<FArrayBox> &ld, int depth, bool domainGhosts)
{
    for (int k = 0; k < ld.size(); k++) {
        const FArrayBox &f = ld[k];
        chk(suhmo_level_put_box(h(), depth, field, k, f.dataPtr(), f.box().lo[0], f.box().lo[1],
                                f.box().hi[0], f.box().hi[1], domainGhosts ? 1 : 0, nullptr), "put_box");
    }
}
void VCAMRNonLinearPoissonOpHIP::get(int field, LevelData<FArrayBox> &ld, int depth)
{
    for (int k = 0; k < ld.size(); k++) {
        FArrayBox &f = ld[k];
        chk(suhmo_level_get_box(h(), depth, field, k, f.dataPtr(), f.box().lo[0], f.box().lo[1],
                                f.box().hi[0], f.box().hi[1], nullptr), "get_box");
    }
}

void VCAMRNonLinearPoissonOpHIP::levelGSRB(LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_rhs, int, int, int a_depth)
{
    put(SUHMO_F_PHI, a_phi, a_depth); put(SUHMO_F_RHS, a_rhs, a_depth);
    chk(suhmo_level_gsrb(h(), a_depth, 1, nullptr), "levelGSRB");
    get(SUHMO_F_PHI, a_phi, a_depth);
}
void VCAMRNonLinearPoissonOpHIP::relax(LevelData<FArrayBox> &a_e, const LevelData<FArrayBox> &a_residual, int a_iterations, int, int a_depth)
{
    put(SUHMO_F_PHI, a_e, a_depth); put(SUHMO_F_RHS, a_residual, a_depth);
    chk(suhmo_level_gsrb(h(), a_depth, a_iterations, nullptr), "relax");   // s_relaxMode 1 only
    get(SUHMO_F_PHI, a_e, a_depth);
}
void VCAMRNonLinearPoissonOpHIP::applyOpI(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_phi, bool a_homogeneous)
{
    put(SUHMO_F_PHI, a_phi, m_depth);
    chk(suhmo_level_apply_op(h(), m_depth, a_homogeneous ? 1 : 0, nullptr), "applyOpI");
    get(SUHMO_F_LPHI, a_lhs, m_depth);
    // the reference leaves phi's ghosts filled (it const-casts a_phi, :278); mirror that
    chk(suhmo_level_fill_ghosts(h(), m_depth, SUHMO_F_PHI, a_homogeneous ? 1 : 0, nullptr), "BC");
    get(SUHMO_F_PHI, const_cast<LevelData<FArrayBox> &>(a_phi), m_depth);
}
void VCAMRNonLinearPoissonOpHIP::applyOpMg(LevelData<FArrayBox> &a_lhs, LevelData<FArrayBox> &a_phi, LevelData<FArrayBox> *a_phiCoarse, bool a_homogeneous)
{
    if (a_phiCoarse != nullptr) MayDay::Abort("VCAMRNonLinearPoissonOpHIP::applyOpMg: coarse-fine interpolation not built yet");
    applyOpI(a_lhs, a_phi, a_homogeneous);
}
void VCAMRNonLinearPoissonOpHIP::residualI(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_phi,
                                           const LevelData<FArrayBox> &a_rhs, bool a_homogeneous)
{
    if (a_homogeneous) MayDay::Abort("VCAMRNonLinearPoissonOp::residualI homogeneous");            // :107-109
    put(SUHMO_F_PHI, a_phi, m_depth); put(SUHMO_F_RHS, a_rhs, m_depth);
    chk(suhmo_level_residual(h(), m_depth, nullptr), "residualI");
    get(SUHMO_F_RES, a_lhs, m_depth);
}
void VCAMRNonLinearPoissonOpHIP::restrictResidual(LevelData<FArrayBox> &a_resCoarse, LevelData<FArrayBox> &a_phiFine,
                                                  const LevelData<FArrayBox> *a_phiCoarse, const LevelData<FArrayBox> &a_rhsFine, bool homogeneous)
{
    if (homogeneous) MayDay::Abort("VCAMRNonLinearPoissonOp::restrictResidual homogeneous");       // :391-393
    if (a_phiCoarse != nullptr) MayDay::Abort("VCAMRNonLinearPoissonOpHIP::restrictResidual: coarse-fine interpolation not built yet");
    put(SUHMO_F_PHI, a_phiFine, m_depth); put(SUHMO_F_RHS, a_rhsFine, m_depth);
    chk(suhmo_level_restrict_residual(h(), m_depth, nullptr), "restrictResidual");
    get(SUHMO_F_RES, a_resCoarse, m_depth + 1);
}
void VCAMRNonLinearPoissonOpHIP::restrictR(LevelData<FArrayBox> &a_phiCoarse, const LevelData<FArrayBox> &a_phiFine)
{
    put(SUHMO_F_PHI, a_phiFine, m_depth);
    for (int k = 0; k < a_phiCoarse.size(); k++) a_phiCoarse[k].setVal(0.0);                      // :365
    chk(suhmo_level_restrict_r(h(), m_depth, nullptr), "restrictR");
    // valid cells only (the coarse ghosts stay zero as in the reference)
    LevelData<FArrayBox> tmp(a_phiCoarse.disjointBoxLayout(), 1, 0);
    get(SUHMO_F_PHI, tmp, m_depth + 1);
    for (int k = 0; k < tmp.size(); k++) {
        const Box &b = tmp[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_phiCoarse[k](i, j) = tmp[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::prolongIncrement(LevelData<FArrayBox> &a_phiThisLevel, const LevelData<FArrayBox> &a_correctCoarse)
{
    put(SUHMO_F_PHI, a_phiThisLevel, m_depth); put(SUHMO_F_CORR, a_correctCoarse, m_depth + 1);
    chk(suhmo_level_prolong_increment(h(), m_depth, nullptr), "prolongIncrement");
    LevelData<FArrayBox> tmp(a_phiThisLevel.disjointBoxLayout(), 1, 0);
    get(SUHMO_F_PHI, tmp, m_depth);
    for (int k = 0; k < tmp.size(); k++) {
        const Box &b = tmp[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_phiThisLevel[k](i, j) = tmp[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::UpdateOperator(const LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> *a_phicoarsePtr, int a_depth, int, bool a_homogeneous)
{
    if (a_homogeneous) MayDay::Abort("VCAMRNonLinearPoissonOp::UpdateOperator homogeneous");       // :42-44
    put(SUHMO_F_PHI, a_phi, a_depth);
    if (a_phicoarsePtr != nullptr) {                                                               // WFlx_level with a coarser level, AmrHydro.cpp:1455-1488
        if (m_amrLevel != 1) MayDay::Error("UpdateOperator with a coarser level: fine operator only");
        m_factory->m_ops[0]->put(SUHMO_F_PHI, *a_phicoarsePtr, 0);
        chk(suhmo_amr2_fine_update_operator(m_factory->m_level, m_factory->m_fine, nullptr), "UpdateOperator (fine)");
        return;
    }
    chk(suhmo_level_update_operator(h(), a_depth, nullptr), "UpdateOperator");
}
void VCAMRNonLinearPoissonOpHIP::AverageOperator(const VCAMRNonLinearPoissonOpHIP &, int a_depth)
{
    chk(suhmo_level_average_operator(h(), a_depth, nullptr), "AverageOperator");
}
void VCAMRNonLinearPoissonOpHIP::getBCoef(LevelData<FluxBox> &a_bCoef)
{
    for (int k = 0; k < a_bCoef.size(); k++)
        for (int dir = 0; dir < 2; dir++) {
            FArrayBox &f = a_bCoef[k][dir];
            chk(suhmo_level_get_box(h(), m_depth, dir == 0 ? SUHMO_F_BX : SUHMO_F_BY, k, f.dataPtr(), f.box().lo[0],
                                    f.box().lo[1], f.box().hi[0], f.box().hi[1], nullptr), "get bCoef");
        }
}
Real VCAMRNonLinearPoissonOpHIP::norm(const LevelData<FArrayBox> &a_x, int a_ord)
{
    put(SUHMO_F_RES, a_x, m_depth);
    double r = 0.0;
    chk(suhmo_level_norm(h(), m_depth, SUHMO_F_RES, a_ord, &r, nullptr), "norm");
    return r;
}
// ---- AMR level methods
void VCAMRNonLinearPoissonOpHIP::relaxNF(LevelData<FArrayBox> &a_e, const LevelData<FArrayBox> *a_eCoarse, const LevelData<FArrayBox> &a_residual,
                                         int a_iterations, int, int a_depth, bool)
{
    put(SUHMO_F_PHI, a_e, a_depth); put(SUHMO_F_RHS, a_residual, a_depth);
    if (a_eCoarse != nullptr) {                                                                   // m_interpWithCoarser.coarseFineInterp :700-702
        if (m_amrLevel != 1) MayDay::Error("relaxNF with a coarser level: fine operator only");
        m_factory->m_ops[0]->put(SUHMO_F_PHI, *a_eCoarse, 0);
        chk(suhmo_amr2_cf_interp(m_factory->m_level, m_factory->m_fine, SUHMO_F_PHI, SUHMO_F_PHI, nullptr), "coarseFineInterp");
    }
    chk(suhmo_level_gsrb(h(), a_depth, a_iterations, nullptr), "relaxNF");
    get(SUHMO_F_PHI, a_e, a_depth);
}
void VCAMRNonLinearPoissonOpHIP::AMRResidualNF(LevelData<FArrayBox> &a_residual, const LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_phiCoarse,
                                               const LevelData<FArrayBox> &a_rhs, bool a_homogeneousPhysBC)
{
    if (a_homogeneousPhysBC) MayDay::Abort("VCAMRNonLinearPoissonOp::residualI homogeneous");
    if (m_amrLevel != 1) MayDay::Error("AMRResidualNF: fine operator only");
    put(SUHMO_F_PHI, a_phi, 0); put(SUHMO_F_RHS, a_rhs, 0);
    m_factory->m_ops[0]->put(SUHMO_F_PHI, a_phiCoarse, 0);
    chk(suhmo_amr2_cf_interp(m_factory->m_level, m_factory->m_fine, SUHMO_F_PHI, SUHMO_F_PHI, nullptr), "coarseFineInterp");
    chk(suhmo_level_residual(h(), 0, nullptr), "AMRResidualNF");
    get(SUHMO_F_RES, a_residual, 0);
}
void VCAMRNonLinearPoissonOpHIP::AMRRestrictS(LevelData<FArrayBox> &a_resCoarse, const LevelData<FArrayBox> &a_residual, const LevelData<FArrayBox> &a_correction,
                                              const LevelData<FArrayBox> &a_coarseCorrection, LevelData<FArrayBox> &a_scratch, bool a_skip_res)
{
    if (m_amrLevel != 1) MayDay::Error("AMRRestrictS: fine operator only");
    // a_resCoarse lives on the coarse LEVEL here (the reference's coarsened-fine layout + copyTo collapse into one step)
    m_factory->m_ops[0]->put(SUHMO_F_RES, a_resCoarse, 0);
    if (!a_skip_res) { AMRResidualNF(a_scratch, a_correction, a_coarseCorrection, a_residual, false); }
    else { put(SUHMO_F_RES, a_residual, 0); assign(a_scratch, a_residual); }                       // "just copy data (phi in this case)"
    chk(suhmo_amr2_average(m_factory->m_level, m_factory->m_fine, SUHMO_F_RES, SUHMO_F_RES, nullptr), "FORT_AVERAGE");
    m_factory->m_ops[0]->get(SUHMO_F_RES, a_resCoarse, 0);
}
void VCAMRNonLinearPoissonOpHIP::AMRProlongS_2(LevelData<FArrayBox> &a_correction, const LevelData<FArrayBox> &a_coarseCorrection)
{
    if (m_amrLevel != 1) MayDay::Error("AMRProlongS_2: fine operator only");
    put(SUHMO_F_PHI, a_correction, 0);
    m_factory->m_ops[0]->put(SUHMO_F_CORR, a_coarseCorrection, 0);
    chk(suhmo_amr2_prolong2(m_factory->m_level, m_factory->m_fine, SUHMO_F_CORR, nullptr), "AMRProlongS_2");
    LevelData<FArrayBox> tmp(a_correction.disjointBoxLayout(), 1, 0);
    get(SUHMO_F_PHI, tmp, 0);
    for (int k = 0; k < tmp.size(); k++) {
        const Box &b = tmp[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_correction[k](i, j) = tmp[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::AMRResidual(LevelData<FArrayBox> &a_residual, const LevelData<FArrayBox> &a_phiFine, const LevelData<FArrayBox> &a_phi,
                                             const LevelData<FArrayBox> &a_rhs, bool a_homogeneousPhysBC, VCAMRNonLinearPoissonOpHIP *a_finerOp)
{
    if (a_homogeneousPhysBC) MayDay::Abort("VCAMRNonLinearPoissonOp::applyOpI homogeneous AMR");
    if (m_amrLevel != 0 || a_finerOp == nullptr || a_finerOp->m_amrLevel != 1) MayDay::Error("AMRResidual: base operator with its finer operator");
    put(SUHMO_F_PHI, a_phi, 0); put(SUHMO_F_RHS, a_rhs, 0);
    a_finerOp->put(SUHMO_F_PHI, a_phiFine, 0);
    chk(suhmo_amr2_residual(m_factory->m_level, m_factory->m_fine, nullptr, nullptr), "AMRResidual");   // applyOpI + reflux; covered cells zeroed
    get(SUHMO_F_RES, a_residual, 0);
}
Real VCAMRNonLinearPoissonOpHIP::AMRNorm(const LevelData<FArrayBox> &a_coarResid, const LevelData<FArrayBox> &a_fineResid, const int &a_refRat, const int &a_ord)
{
    if (a_refRat != 2) MayDay::Error("AMRNorm: refinement ratio 2");
    put(SUHMO_F_RES, a_coarResid, 0);
    if (a_fineResid.size() > 0 && m_factory->m_fine)
        chk(suhmo_amr2_set_covered(m_factory->m_level, m_factory->m_fine, SUHMO_F_RES, 0.0, nullptr), "AMRNorm: zero under the finer grids");
    double r = 0.0;
    chk(suhmo_level_norm(h(), 0, SUHMO_F_RES, a_ord, &r, nullptr), "AMRNorm");
    return r;
}

// ---- the rest of the MGLevelOp / AMRLevelOp interface
void VCAMRNonLinearPoissonOpHIP::residual(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_rhs, bool)
{ residualI(a_lhs, a_phi, a_rhs, false); }                                                         // m_use_FAS: always the inhomogeneous form
void VCAMRNonLinearPoissonOpHIP::residualNF(LevelData<FArrayBox> &a_lhs, LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> *a_phiCoarse,
                                            const LevelData<FArrayBox> &a_rhs, bool a_homogeneous)
{
    if (a_phiCoarse != nullptr) AMRResidualNF(a_lhs, a_phi, *a_phiCoarse, a_rhs, a_homogeneous);
    else residualI(a_lhs, a_phi, a_rhs, a_homogeneous);
}
void VCAMRNonLinearPoissonOpHIP::preCond(LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_rhs)
{
    chk(suhmo_level_compute_lambda(h(), m_depth, nullptr), "resetLambda");
    LevelData<FArrayBox> lam(a_rhs.disjointBoxLayout(), 1, 0);
    get(SUHMO_F_LAMBDA, lam, m_depth);
    for (int k = 0; k < a_phi.size(); k++) {                                                       // a_phi = a_rhs / lambda on the rhs box
        const Box &b = a_rhs[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_phi[k](i, j) = a_rhs[k](i, j) / lam[k](i, j);
    }
    relax(a_phi, a_rhs, 2, 0, m_depth);
}
void VCAMRNonLinearPoissonOpHIP::preCond(LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &, const LevelData<FArrayBox> &a_rhs)
{ relax(a_phi, a_rhs, 2, 0, m_depth); }
void VCAMRNonLinearPoissonOpHIP::applyOp(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_phi, bool)
{ applyOpI(a_lhs, a_phi, false); }
void VCAMRNonLinearPoissonOpHIP::applyOpNoBoundary(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_phi)
{
    // the reference evaluates the stencil on a_phi's ghost cells as they are; it only ever calls this right after the BC
    // fill of applyOpI (:273-289).  The device evaluates the physical BC on the fly, so for that calling sequence the
    // result is the same; ghost values a caller invented are not honoured.
    put(SUHMO_F_PHI, a_phi, m_depth);
    chk(suhmo_level_apply_op(h(), m_depth, 0, nullptr), "applyOpNoBoundary");
    get(SUHMO_F_LPHI, a_lhs, m_depth);
}
void VCAMRNonLinearPoissonOpHIP::setAlphaAndBeta(const Real &a_alpha, const Real &a_beta)
{
    chk(suhmo_level_set_alpha_beta(h(), a_alpha, a_beta), "setAlphaAndBeta");
    if (m_amrLevel == 0) { m_factory->m_desc.alpha = a_alpha; m_factory->m_desc.beta = a_beta; }
}
void VCAMRNonLinearPoissonOpHIP::setBC(const suhmo_bc_t &a_bc) { chk(suhmo_level_set_bc(h(), &a_bc), "setBC"); }
void VCAMRNonLinearPoissonOpHIP::getFlux(FArrayBox &a_flux, const FArrayBox &a_data, const FluxBox &a_bCoef, const Box &a_facebox,
                                         int a_dir, int a_ref) const
{
    const suhmo_level_desc_t &d = m_factory->m_desc;
    const Real dxd = (a_dir == 0 ? d.dx : d.dy) * (1 << m_depth) / (m_amrLevel == 1 ? 2.0 : 1.0);
    const Real scale = d.beta * a_ref / dxd;                                                       // m_beta * a_ref / m_dx_vect[a_dir]
    a_flux.define(a_facebox, 1);
    const FArrayBox &b = a_bCoef[a_dir];
    for (int j = a_facebox.lo[1]; j <= a_facebox.hi[1]; j++)
        for (int i = a_facebox.lo[0]; i <= a_facebox.hi[0]; i++) {
            Real phihi = a_data(i, j), philo = a_dir == 0 ? a_data(i - 1, j) : a_data(i, j - 1);
            Real gradphi = (phihi - philo) * scale;
            a_flux(i, j) = -b(i, j) * gradphi;
        }
}
void VCAMRNonLinearPoissonOpHIP::finerOperatorChanged(const VCAMRNonLinearPoissonOpHIP &a_operator, int a_coarseningFactor)
{
    if (a_coarseningFactor == 1) return;                                                           // only the exchanges remain (:1431-1438)
    if (a_operator.m_amrLevel == m_amrLevel)                                                       // next finer multigrid depth of the same level
        chk(suhmo_level_build_mg_coefficients(h(), nullptr), "finerOperatorChanged (multigrid depths)");
    else {
        if (!(m_amrLevel == 0 && a_operator.m_amrLevel == 1 && a_coarseningFactor == 2)) MayDay::Error("finerOperatorChanged: base operator <- fine operator, ratio 2");
        chk(suhmo_amr2_finer_operator_changed(m_factory->m_level, m_factory->m_fine, nullptr), "finerOperatorChanged (AMR levels)");
    }
}
void VCAMRNonLinearPoissonOpHIP::AMROperatorNF(LevelData<FArrayBox> &a_LofPhi, const LevelData<FArrayBox> &a_phi, const LevelData<FArrayBox> &a_phiCoarse,
                                               bool a_homogeneousPhysBC)
{
    if (a_homogeneousPhysBC) MayDay::Abort("VCAMRNonLinearPoissonOp::applyOpI homogeneous AMR");
    if (m_amrLevel != 1) MayDay::Error("AMROperatorNF: fine operator only");
    put(SUHMO_F_PHI, a_phi, 0);
    m_factory->m_ops[0]->put(SUHMO_F_PHI, a_phiCoarse, 0);
    chk(suhmo_amr2_cf_interp(m_factory->m_level, m_factory->m_fine, SUHMO_F_PHI, SUHMO_F_PHI, nullptr), "coarseFineInterp");
    chk(suhmo_level_apply_op(h(), 0, 0, nullptr), "AMROperatorNF");
    get(SUHMO_F_LPHI, a_LofPhi, 0);
}
void VCAMRNonLinearPoissonOpHIP::AMROperatorNC(LevelData<FArrayBox> &a_LofPhi, const LevelData<FArrayBox> &a_phiFine, const LevelData<FArrayBox> &a_phi,
                                               bool a_homogeneousPhysBC, VCAMRNonLinearPoissonOpHIP *a_finerOp)
{
    if (a_homogeneousPhysBC) MayDay::Abort("VCAMRNonLinearPoissonOp::applyOpI homogeneous AMR");
    if (m_amrLevel != 0) MayDay::Error("AMROperatorNC: base operator");
    put(SUHMO_F_PHI, a_phi, 0);
    chk(suhmo_level_apply_op(h(), 0, 0, nullptr), "AMROperatorNC: applyOpI");
    if (a_phiFine.size() > 0) {                                                                    // a_phiFine.isDefined()
        if (a_finerOp == nullptr || a_finerOp->m_amrLevel != 1) MayDay::Error("AMROperatorNC: finer operator");
        a_finerOp->put(SUHMO_F_PHI, a_phiFine, 0);
        chk(suhmo_amr2_reflux(m_factory->m_level, m_factory->m_fine, SUHMO_F_LPHI, nullptr), "reflux");
    }
    get(SUHMO_F_LPHI, a_LofPhi, 0);
}
void VCAMRNonLinearPoissonOpHIP::AMROperator(LevelData<FArrayBox> &a_LofPhi, const LevelData<FArrayBox> &a_phiFine, const LevelData<FArrayBox> &a_phi,
                                             const LevelData<FArrayBox> &a_phiCoarse, bool a_homogeneousPhysBC, VCAMRNonLinearPoissonOpHIP *a_finerOp)
{
    if (a_phiCoarse.size() == 0) { AMROperatorNC(a_LofPhi, a_phiFine, a_phi, a_homogeneousPhysBC, a_finerOp); return; }
    if (a_phiFine.size() == 0) { AMROperatorNF(a_LofPhi, a_phi, a_phiCoarse, a_homogeneousPhysBC); return; }
    MayDay::Error("AMROperator on a middle level: the two-level mirror has none (suhmo_amr_* of the C-ABI handle N levels)");
}
void VCAMRNonLinearPoissonOpHIP::reflux(const LevelData<FArrayBox> &a_phiFine, const LevelData<FArrayBox> &a_phi, LevelData<FArrayBox> &a_residual,
                                        VCAMRNonLinearPoissonOpHIP *a_finerOp)
{
    if (m_amrLevel != 0 || a_finerOp == nullptr || a_finerOp->m_amrLevel != 1) MayDay::Error("reflux: base operator with its finer operator");
    put(SUHMO_F_PHI, a_phi, 0); put(SUHMO_F_LPHI, a_residual, 0);
    a_finerOp->put(SUHMO_F_PHI, a_phiFine, 0);
    chk(suhmo_amr2_reflux(m_factory->m_level, m_factory->m_fine, SUHMO_F_LPHI, nullptr), "reflux");
    get(SUHMO_F_LPHI, a_residual, 0);
}
void VCAMRNonLinearPoissonOpHIP::AMRRestrict(LevelData<FArrayBox> &a_resCoarse, const LevelData<FArrayBox> &a_residual, const LevelData<FArrayBox> &a_correction,
                                             const LevelData<FArrayBox> &a_coarseCorrection, bool a_skip_res)
{
    LevelData<FArrayBox> r;
    create(r, a_residual);
    AMRRestrictS(a_resCoarse, a_residual, a_correction, a_coarseCorrection, r, a_skip_res);
}
void VCAMRNonLinearPoissonOpHIP::AMRProlong(LevelData<FArrayBox> &a_correction, const LevelData<FArrayBox> &a_coarseCorrection)
{
    if (m_amrLevel != 1) MayDay::Error("AMRProlong: fine operator only");
    put(SUHMO_F_PHI, a_correction, 0);
    m_factory->m_ops[0]->put(SUHMO_F_CORR, a_coarseCorrection, 0);
    chk(suhmo_amr2_prolong_pc(m_factory->m_level, m_factory->m_fine, SUHMO_F_CORR, nullptr), "AMRProlong");
    LevelData<FArrayBox> tmp(a_correction.disjointBoxLayout(), 1, 0);
    get(SUHMO_F_PHI, tmp, 0);
    for (int k = 0; k < tmp.size(); k++) {
        const Box &b = tmp[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_correction[k](i, j) = tmp[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::AMRProlongS(LevelData<FArrayBox> &a_correction, const LevelData<FArrayBox> &a_coarseCorrection,
                                             LevelData<FArrayBox> &, const Copier &)
{ AMRProlong(a_correction, a_coarseCorrection); }                                                  // temp / copier: the device reads the coarse level directly
void VCAMRNonLinearPoissonOpHIP::AMRUpdateResidual(LevelData<FArrayBox> &a_residual, const LevelData<FArrayBox> &a_correction,
                                                   const LevelData<FArrayBox> &a_coarseCorrection)
{ AMRResidualNF(a_residual, a_correction, a_coarseCorrection, a_residual, false); }

void VCAMRNonLinearPoissonOpHIP::createCoarsened(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs, const int &a_refRat)
{
    if (!a_rhs.disjointBoxLayout().coarsenable(a_refRat)) MayDay::Error("createCoarsened: layout not coarsenable");
    a_lhs.define(a_rhs.disjointBoxLayout().coarsened(a_refRat), 1, a_rhs.ghost());
}
void VCAMRNonLinearPoissonOpHIP::assignLocal(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs)
{
    for (int k = 0; k < a_lhs.size(); k++) {
        const Box &b = a_lhs[k].box(), &r = a_rhs[k].box();
        for (int j = std::max(b.lo[1], r.lo[1]); j <= std::min(b.hi[1], r.hi[1]); j++)
            for (int i = std::max(b.lo[0], r.lo[0]); i <= std::min(b.hi[0], r.hi[0]); i++) a_lhs[k](i, j) = a_rhs[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::buildCopier(Copier &a_copier, const LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs)
{ a_copier.define(a_rhs.disjointBoxLayout(), a_lhs.disjointBoxLayout()); }
template <class F> static void for_overlaps(const DisjointBoxLayout &src, const DisjointBoxLayout &dst, F f)
{
    for (int kd = 0; kd < dst.size(); kd++)
        for (int ks = 0; ks < src.size(); ks++) {
            const Box &a = dst[kd], &b = src[ks];
            Box o(std::max(a.lo[0], b.lo[0]), std::max(a.lo[1], b.lo[1]), std::min(a.hi[0], b.hi[0]), std::min(a.hi[1], b.hi[1]));
            if (o.size(0) > 0 && o.size(1) > 0) f(kd, ks, o);
        }
}
void VCAMRNonLinearPoissonOpHIP::assignCopier(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs, const Copier &a_copier)
{
    for_overlaps(a_copier.src(), a_copier.dst(), [&](int kd, int ks, const Box &o) {
        for (int j = o.lo[1]; j <= o.hi[1]; j++) for (int i = o.lo[0]; i <= o.hi[0]; i++) a_lhs[kd](i, j) = a_rhs[ks](i, j); });
}
void VCAMRNonLinearPoissonOpHIP::zeroCovered(LevelData<FArrayBox> &a_lhs, LevelData<FArrayBox> &, const Copier &a_copier)
{
    for_overlaps(a_copier.src(), a_copier.dst(), [&](int kd, int, const Box &o) {
        for (int j = o.lo[1]; j <= o.hi[1]; j++) for (int i = o.lo[0]; i <= o.hi[0]; i++) a_lhs[kd](i, j) = 0.0; });
}
Real VCAMRNonLinearPoissonOpHIP::dotProduct(const LevelData<FArrayBox> &a_1, const LevelData<FArrayBox> &a_2)
{
    Real sum = 0.0;                                                                                // box by box, Fortran order inside a box
    for (int k = 0; k < a_1.size(); k++) {
        const Box &v = a_1.disjointBoxLayout()[k];
        Real s = 0.0;
        for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) s += a_1[k](i, j) * a_2[k](i, j);
        sum += s;
    }
    return sum;
}
void VCAMRNonLinearPoissonOpHIP::mDotProduct(const LevelData<FArrayBox> &a_1, const int a_sz, const LevelData<FArrayBox> a_2[], Real a_mdots[])
{ for (int q = 0; q < a_sz; q++) a_mdots[q] = dotProduct(a_1, a_2[q]); }
void VCAMRNonLinearPoissonOpHIP::scale(LevelData<FArrayBox> &a_lhs, const Real &a_scale)
{
    for (int k = 0; k < a_lhs.size(); k++) {
        const Box &b = a_lhs[k].box();
        for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) a_lhs[k](i, j) *= a_scale;
    }
}
Real VCAMRNonLinearPoissonOpHIP::localMaxNorm(const LevelData<FArrayBox> &a_x)
{
    Real m = 0.0;
    for (int k = 0; k < a_x.size(); k++) {
        const Box &v = a_x.disjointBoxLayout()[k];
        for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) m = std::max(m, std::fabs(a_x[k](i, j)));
    }
    return m;
}

void VCAMRNonLinearPoissonOpHIP::create(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs)
{ a_lhs.define(a_rhs.disjointBoxLayout(), 1, a_rhs.ghost()); }
void VCAMRNonLinearPoissonOpHIP::createCoarser(LevelData<FArrayBox> &a_coarse, const LevelData<FArrayBox> &a_fine, bool)
{ a_coarse.define(m_factory->m_grids[m_depth + 1], 1, a_fine.ghost()); }                          // AMRNL...cpp:753-766
void VCAMRNonLinearPoissonOpHIP::assign(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_rhs)
{ axby(a_lhs, a_rhs, a_rhs, 1.0, 0.0); }
void VCAMRNonLinearPoissonOpHIP::incr(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_x, Real a_scale)
{ axby(a_lhs, a_lhs, a_x, 1.0, a_scale); }
void VCAMRNonLinearPoissonOpHIP::axby(LevelData<FArrayBox> &a_lhs, const LevelData<FArrayBox> &a_x, const LevelData<FArrayBox> &a_y, Real a, Real b)
{
    for (int k = 0; k < a_lhs.size(); k++) {
        const Box &v = a_lhs.disjointBoxLayout()[k];
        for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++)
            a_lhs[k](i, j) = (b == 0.0) ? a * a_x[k](i, j) : a * a_x[k](i, j) + b * a_y[k](i, j);
    }
}
void VCAMRNonLinearPoissonOpHIP::setToZero(LevelData<FArrayBox> &a_lhs) { for (int k = 0; k < a_lhs.size(); k++) a_lhs[k].setVal(0.0); }

} // namespace suhmo_host
