// test_mirror.cpp -- drives the C++ host mirror (suhmo_amd/host) exactly the way Chombo's
// multigrid drives VCAMRNonLinearPoissonOp (MGnewOp, relax, restrictResidual, restrictR,
// applyOpMg, prolongIncrement, UpdateOperator, AverageOperator, norm, solve) on box-decomposed
// LevelData, and checks every result BITWISE against the CPU oracle (test infrastructure).
// Built and run by tests/test_gpu_host_mirror.py on the GPU box.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../suhmo_amd/host/VCAMRNonLinearPoissonOpHIP.H"
#include "../../oracle/level_shim.h"

using namespace suhmo_host;

// oracle/amr2.c (test infrastructure; no public header)
extern "C" {
typedef struct OrAmr2 OrAmr2;
OrAmr2 *or_amr2_create(OrLevel *coarse, int nxc, int nyc, double dxc, double dyc, const OrBC *bc, const OrPhys *ph,
                       double alpha, double beta, int ci0, int cj0, int ci1, int cj1);
void or_amr2_destroy(OrAmr2 *A);
void or_amr2_fine_io(OrAmr2 *A, int field, double *g, int ghosted, int set);
int or_amr2_solve(OrAmr2 *A, const OrSolverParams *sp, double *hist);
void or_amr2_cf_interp_phi(OrAmr2 *A);
void or_amr2_fine_gsrb(OrAmr2 *A, int sweeps);
void or_amr2_fine_update_operator(OrAmr2 *A);
void or_amr2_fine_apply_op(OrAmr2 *A, int homogeneous);
double or_amr2_residual(OrAmr2 *A);
const double *or_amr2_coarse_residual(const OrAmr2 *A);
}

static int g_fail = 0;
// oracle/amrm.c (test infrastructure; no public header): hierarchies whose levels are unions of boxes
struct OrAmrM;
extern "C" {
OrAmrM *or_amrm_create(OrLevel *base, int nx0, int ny0, double dx0, double dy0, const OrBC *bc, const OrPhys *ph,
                       double alpha, double beta, int nlev, const int *nbox, const int *boxes);
void or_amrm_destroy(OrAmrM *A);
void or_amrm_box_io(OrAmrM *A, int l, int k, int field, double *g, int ghosted, int set);
int or_amrm_solve(OrAmrM *A, const OrSolverParams *sp, double *hist);
}

#define CHECK(cond, msg) do { if (!(cond)) { printf("FAIL: %s\n", msg); g_fail++; } else printf("ok:   %s\n", msg); } while (0)

static double hashv(int k, double a, double b) { return a + (b - a) * (double)(((unsigned)k * 2654435761u) % 1000003u) / 1000003.0; }

// LevelData (valid cells) <-> global row-major array
static std::vector<double> to_global(const LevelData<FArrayBox> &ld, int nx, int ny, int g)
{
    std::vector<double> a((size_t)(nx + 2 * g) * (ny + 2 * g), 0.0);
    for (int k = 0; k < ld.size(); k++) {
        const Box &v = ld.disjointBoxLayout()[k];
        Box b = v;
        if (g) { if (v.lo[0] == 0) b.lo[0] -= 1; if (v.hi[0] == nx - 1) b.hi[0] += 1; if (v.lo[1] == 0) b.lo[1] -= 1; if (v.hi[1] == ny - 1) b.hi[1] += 1; }
        for (int j = b.lo[1]; j <= b.hi[1]; j++)
            for (int i = b.lo[0]; i <= b.hi[0]; i++)
                if ((i >= 0 && i < nx) || (j >= 0 && j < ny))       // no corner ghosts
                    a[(size_t)(j + g) * (nx + 2 * g) + (i + g)] = ld[k](i, j);
    }
    return a;
}
static bool same_valid(const LevelData<FArrayBox> &ld, const std::vector<double> &glob, int nx)
{
    for (int k = 0; k < ld.size(); k++) {
        const Box &v = ld.disjointBoxLayout()[k];
        for (int j = v.lo[1]; j <= v.hi[1]; j++)
            for (int i = v.lo[0]; i <= v.hi[0]; i++) {
                double a = ld[k](i, j), b = glob[(size_t)j * nx + i];
                if (memcmp(&a, &b, 8) != 0) { printf("   mismatch at (%d,%d): %.17g vs %.17g\n", i, j, a, b); return false; }
            }
    }
    return true;
}

int main()
{
    const int nx = 128, ny = 64, mb = 32;
    const double dxv = 781.25, dyv = 312.5;
    ProblemDomain dom; dom.dom = Box(0, 0, nx - 1, ny - 1); dom.periodic[0] = false; dom.periodic[1] = false;
    std::vector<Box> bx;
    for (int bj = 0; bj < ny / mb; bj++) for (int bi = 0; bi < nx / mb; bi++) bx.push_back(Box(bi * mb, bj * mb, bi * mb + mb - 1, bj * mb + mb - 1));
    DisjointBoxLayout grids(bx, dom);
    RealVect dx; dx[0] = dxv; dx[1] = dyv;

    suhmo_bc_t bc; memset(&bc, 0, sizeof(bc));
    bc.type[0][0] = 0; bc.type[0][1] = 1; bc.type[1][0] = 1; bc.type[1][1] = 1;       // A-suite BCs
    suhmo_phys_t ph = {5e-25, 1e-3, 1.787e-6, 0.0, 1.0e4, 9800.0, 9.8, 0, 1, 0};

    LevelData<FArrayBox> phi(grids, 1, 1), rhs(grids, 1, 0), aCoef(grids, 1, 0), B(grids, 1, 1), Pi(grids, 1, 1), zb(grids, 1, 1), mask(grids, 1, 1);
    LevelData<FluxBox> bCoef(grids, 1, 0);
    for (int k = 0; k < grids.size(); k++) {
        const Box g = grids[k].grown(1);
        for (int j = g.lo[1]; j <= g.hi[1]; j++)
            for (int i = g.lo[0]; i <= g.hi[0]; i++) {
                double x = (i + 0.5) * dxv;
                double H = 6.0 * (std::sqrt(x + 5000.0) - std::sqrt(5000.0)) + 1.0; if (H < 0) H = 0;
                B[k](i, j) = 0.01 * (1.0 + 0.4 * std::sin(0.37 * i) * std::cos(0.23 * j));
                Pi[k](i, j) = 910.0 * 9.8 * H; zb[k](i, j) = 0.0; mask[k](i, j) = 1.0;
                phi[k](i, j) = 101325.0 / 9800.0 + 1e-3 * hashv(i * 1000 + j, -1.0, 1.0);
            }
        const Box &v = grids[k];
        for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { rhs[k](i, j) = 5.79e-9; aCoef[k](i, j) = 0.0; }
    }

    // ---- oracle twin
    OrBC obc; memcpy(&obc, &bc, sizeof(obc));
    OrPhys oph; memcpy(&oph, &ph, sizeof(oph));
    OrLevel *O = or_level_create(nx, ny, dxv, dyv, mb, &obc, &oph, 0.0, -1.0, 2);
    { auto g = to_global(phi, nx, ny, 0); or_level_set(O, 0, OR_F_PHI, g.data(), 0); }
    { auto g = to_global(rhs, nx, ny, 0); or_level_set(O, 0, OR_F_RHS, g.data(), 0); }
    { auto g = to_global(aCoef, nx, ny, 0); or_level_set(O, 0, OR_F_ACOEF, g.data(), 0); }
    { auto g = to_global(B, nx, ny, 1); or_level_set(O, 0, OR_F_B, g.data(), 1); }
    { auto g = to_global(Pi, nx, ny, 1); or_level_set(O, 0, OR_F_PI, g.data(), 1); }
    { auto g = to_global(zb, nx, ny, 1); or_level_set(O, 0, OR_F_ZB, g.data(), 1); }
    { auto g = to_global(mask, nx, ny, 1); or_level_set(O, 0, OR_F_MASK, g.data(), 1); }

    // ---- factory / operators, as AmrHydro::SolveForHead_nl + AMRFASMultiGrid::define do
    VCAMRNonLinearPoissonOpHIPFactory fac;
    fac.define(dom, grids, dx, bc, 0.0, aCoef, -1.0, bCoef, ph, B, Pi, zb, mask, true);
    CHECK(fac.numDepths() == or_level_num_depths(O), "MGnewOp depth rule: coarsenable(2^d * s_maxCoarse)");
    CHECK(fac.MGnewOp(dom, fac.numDepths()) == nullptr, "MGnewOp returns NULL past the coarsest depth");
    VCAMRNonLinearPoissonOpHIP *op = fac.AMRnewOp(dom);
    VCAMRNonLinearPoissonOpHIP *opC = fac.MGnewOp(dom, 1);
    std::vector<double> og((size_t)nx * ny), ogc((size_t)nx * ny / 4);

    op->UpdateOperator(phi, nullptr, 0, 0, false);
    or_level_update_operator(O, 0);
    or_level_build_mg_coefficients(O);
    opC->AverageOperator(*op, 1);

    op->relax(phi, rhs, 4, 0, 0);
    or_level_gsrb(O, 0, 4); or_level_get(O, 0, OR_F_PHI, og.data(), 0);
    CHECK(same_valid(phi, og, nx), "relax(4) == oracle levelGSRB x4 (bitwise)");

    LevelData<FArrayBox> res, resC, phiC, LphiC;
    op->create(res, rhs); op->createCoarser(resC, rhs, false); op->createCoarser(phiC, phi, true); opC->create(LphiC, resC);
    op->residualI(res, phi, rhs, false);
    or_level_residual(O, 0); or_level_get(O, 0, OR_F_RES, og.data(), 0);
    CHECK(same_valid(res, og, nx), "residualI == oracle");
    CHECK(op->norm(res, 0) == or_level_norm(O, 0, OR_F_RES, 0), "norm(res, 0) == oracle max-norm");

    op->restrictResidual(resC, phi, nullptr, rhs, false);
    or_level_restrict_residual(O, 0); or_level_get(O, 1, OR_F_RES, ogc.data(), 0);
    CHECK(same_valid(resC, ogc, nx / 2), "restrictResidual == oracle");
    op->restrictR(phiC, phi);
    or_level_restrict_r(O, 0); or_level_get(O, 1, OR_F_PHI, ogc.data(), 0);
    CHECK(same_valid(phiC, ogc, nx / 2), "restrictR == oracle");
    opC->applyOpMg(LphiC, phiC, nullptr, false);
    or_level_apply_op(O, 1, 0); or_level_get(O, 1, OR_F_LPHI, ogc.data(), 0);
    CHECK(same_valid(LphiC, ogc, nx / 2), "applyOpMg on depth 1 == oracle");

    // prolongIncrement with the restricted phi as 'correction'
    { auto g = to_global(phiC, nx / 2, ny / 2, 0); or_level_prolong_increment(O, 0, g.data()); }
    op->prolongIncrement(phi, phiC);
    or_level_get(O, 0, OR_F_PHI, og.data(), 0);
    CHECK(same_valid(phi, og, nx), "prolongIncrement == oracle");

    // whole solve, parameters of SolveForHead_nl for m_cur_step >= 50
    HeadSolverParameters sp(100, true);
    OrSolverParams osp; memcpy(&osp, static_cast<suhmo_solver_params_t *>(&sp), sizeof(osp));
    std::vector<double> hist, ohist(sp.max_iter + 2);
    int n = fac.solve(phi, rhs, sp, &hist);
    int on = or_level_solve(O, &osp, ohist.data());
    or_level_get(O, 0, OR_F_PHI, og.data(), 0);
    CHECK(n == on, "solve: same number of V-cycles as the oracle");
    CHECK(same_valid(phi, og, nx), "solve: converged head == oracle (bitwise)");
    printf("V-cycles %d, residual %.3e -> %.3e\n", n, hist.front(), hist.back());

    // ================= the rest of the MGLevelOp interface on the base level (a second oracle level carries the
    // changed alpha / beta / boundary values)
    {
        LevelData<FArrayBox> L1(grids, 1, 0), L2(grids, 1, 0), r1(grids, 1, 0), r2(grids, 1, 0);
        op->UpdateOperator(phi, nullptr, 0, 0, false);
        { auto g = to_global(phi, nx, ny, 0); or_level_set(O, 0, OR_F_PHI, g.data(), 0); }
        or_level_update_operator(O, 0);
        op->applyOp(L1, phi, false);
        or_level_apply_op(O, 0, 0); or_level_get(O, 0, OR_F_LPHI, og.data(), 0);
        CHECK(same_valid(L1, og, nx), "applyOp == oracle applyOpI");
        op->applyOpNoBoundary(L2, phi);
        CHECK(same_valid(L2, og, nx), "applyOpNoBoundary (after the BC fill) == oracle");
        op->residual(r1, phi, rhs, false); op->residualNF(r2, phi, nullptr, rhs, false);
        or_level_residual(O, 0); or_level_get(O, 0, OR_F_RES, og.data(), 0);
        CHECK(same_valid(r1, og, nx) && same_valid(r2, og, nx), "residual / residualNF(no coarser level) == oracle residualI");
        // getFlux on one box: phi's ghost cells were filled by applyOp (as the reference's const-cast does)
        LevelData<FluxBox> cb(grids, 1, 0);
        op->getBCoef(cb);
        std::vector<double> fl((size_t)(nx + 1) * ny);
        suhmo_level_get_flux(fac.handle(), 0, 0, 1, fl.data(), nullptr);
        FArrayBox flux;
        const Box fbx0 = grids[0].surroundingNodes(0);
        op->getFlux(flux, phi[0], cb[0], fbx0, 0, 1);
        bool okfl = true;
        for (int j = fbx0.lo[1]; j <= fbx0.hi[1]; j++) for (int i = fbx0.lo[0]; i <= fbx0.hi[0]; i++) { double a = flux(i, j), b = fl[(size_t)j * (nx + 1) + i]; if (memcmp(&a, &b, 8)) okfl = false; }
        CHECK(okfl, "getFlux(box 0, dir 0) == the level-wide device flux (bitwise)");
        // preCond: correction = residual / lambda, then two sweeps
        LevelData<FArrayBox> corr(grids, 1, 1);
        op->preCond(corr, r1);
        or_level_reset_lambda(O, 0);
        std::vector<double> lam((size_t)nx * ny), p0((size_t)nx * ny), rg = to_global(r1, nx, ny, 0);
        or_level_get(O, 0, OR_F_LAMBDA, lam.data(), 0);
        for (size_t q = 0; q < p0.size(); q++) p0[q] = rg[q] / lam[q];
        or_level_set(O, 0, OR_F_PHI, p0.data(), 0); or_level_set(O, 0, OR_F_RHS, rg.data(), 0);
        or_level_gsrb(O, 0, 2); or_level_get(O, 0, OR_F_PHI, og.data(), 0);
        CHECK(same_valid(corr, og, nx), "preCond == rhs / lambda + relax(2) of the oracle");
        // setAlphaAndBeta + setBC against an oracle level created with those values
        suhmo_bc_t bc2 = bc; bc2.value[0][0] = 3.0; bc2.type[1][1] = 0; bc2.value[1][1] = 11.0;
        OrBC obc2; memcpy(&obc2, &bc2, sizeof(obc2));
        OrLevel *O2 = or_level_create(nx, ny, dxv, dyv, mb, &obc2, &oph, 0.0, -2.0, 2);
        { auto g = to_global(phi, nx, ny, 0); or_level_set(O2, 0, OR_F_PHI, g.data(), 0); }
        { auto g = to_global(rhs, nx, ny, 0); or_level_set(O2, 0, OR_F_RHS, g.data(), 0); }
        { auto g = to_global(aCoef, nx, ny, 0); or_level_set(O2, 0, OR_F_ACOEF, g.data(), 0); }
        { auto g = to_global(B, nx, ny, 1); or_level_set(O2, 0, OR_F_B, g.data(), 1); }
        { auto g = to_global(Pi, nx, ny, 1); or_level_set(O2, 0, OR_F_PI, g.data(), 1); }
        { auto g = to_global(zb, nx, ny, 1); or_level_set(O2, 0, OR_F_ZB, g.data(), 1); }
        { auto g = to_global(mask, nx, ny, 1); or_level_set(O2, 0, OR_F_MASK, g.data(), 1); }
        op->setAlphaAndBeta(0.0, -2.0); op->setBC(bc2);
        op->UpdateOperator(phi, nullptr, 0, 0, false); or_level_update_operator(O2, 0);
        op->applyOp(L1, phi, false);
        or_level_apply_op(O2, 0, 0); or_level_get(O2, 0, OR_F_LPHI, og.data(), 0);
        CHECK(same_valid(L1, og, nx), "setAlphaAndBeta(0, -2) + setBC(new values, Dirichlet on y-hi): applyOp == oracle level created that way");
        op->relax(phi, rhs, 2, 0, 0);
        or_level_gsrb(O2, 0, 2); or_level_get(O2, 0, OR_F_PHI, og.data(), 0);
        CHECK(same_valid(phi, og, nx), "... and relax(2) too");
        op->setAlphaAndBeta(0.0, -1.0); op->setBC(bc);
        or_level_destroy(O2);
        // LevelDataOps on the host containers
        LevelData<FArrayBox> a1, a2, cz;
        op->create(a1, phi); op->assignLocal(a1, phi); op->create(a2, phi); op->assign(a2, phi);
        Real dp = op->dotProduct(a1, a2), manual = 0.0;
        for (int k = 0; k < grids.size(); k++) { const Box &v = grids[k]; Real sb = 0.0; for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) sb += phi[k](i, j) * phi[k](i, j); manual += sb; }
        LevelData<FArrayBox> two[2]; op->create(two[0], phi); op->assign(two[0], phi); op->create(two[1], phi); op->assign(two[1], phi); op->scale(two[1], 2.0);
        Real md[2]; op->mDotProduct(a1, 2, two, md);
        CHECK(dp == manual && md[0] == dp && md[1] == 2.0 * dp, "dotProduct / mDotProduct / scale");
        CHECK(op->localMaxNorm(a1) == op->norm(a1, 0), "localMaxNorm == norm(x, 0) on one rank");
        op->createCoarsened(cz, phi, 2);
        CHECK(cz.size() == grids.size() && cz[0].box().size(0) == mb / 2 + 2, "createCoarsened: coarsened layout, same ghosts");
        Copier cp; LevelData<FArrayBox> part(DisjointBoxLayout(std::vector<Box>{Box(8, 8, 39, 23)}, dom), 1, 0);
        part[0].setVal(7.0);
        op->buildCopier(cp, a1, part); op->assignCopier(a1, part, cp);
        bool okcp = a1[0](8, 8) == 7.0 && a1[1](39, 23) == 7.0 && a1[0](7, 8) == phi[0](7, 8);
        op->zeroCovered(a2, part, cp);
        okcp = okcp && a2[0](8, 8) == 0.0 && a2[1](39, 23) == 0.0 && a2[0](7, 8) == phi[0](7, 8) && a2[1](40, 23) == phi[1](40, 23);
        CHECK(okcp, "buildCopier / assignCopier / zeroCovered on overlapping layouts");
        op->setTime(3600.0); op->diagonalScale(a1, true); op->divideByIdentityCoef(a1);
        // the oracle level continues from the same state as the mirror
        { auto g = to_global(phi, nx, ny, 0); or_level_set(O, 0, OR_F_PHI, g.data(), 0); }
        { auto g = to_global(rhs, nx, ny, 0); or_level_set(O, 0, OR_F_RHS, g.data(), 0); }
        op->UpdateOperator(phi, nullptr, 0, 0, false); or_level_update_operator(O, 0);     // bCoef of the restored BC, both sides
        HeadSolverParameters sp1(100, true); sp1.max_iter = 2;
        OrSolverParams osp1; memcpy(&osp1, static_cast<suhmo_solver_params_t *>(&sp1), sizeof(osp1));
        std::vector<double> oh1(sp1.max_iter + 2);
        int n1 = fac.solve(phi, rhs, sp1, nullptr), on1 = or_level_solve(O, &osp1, oh1.data());
        or_level_get(O, 0, OR_F_PHI, og.data(), 0);
        CHECK(n1 == on1 && same_valid(phi, og, nx), "after the setAlphaAndBeta / setBC round trip: two V-cycles == oracle (bitwise)");
    }

    // ================= two AMR levels: the base level above + a fine patch (coarse cells [32..95] x [16..47])
    {
        const int ci0 = 32, cj0 = 16, ci1 = 95, cj1 = 47, fnx = 2 * (ci1 - ci0 + 1), fny = 2 * (cj1 - cj0 + 1), fmb = 32;
        ProblemDomain fdom; fdom.dom = Box(0, 0, 2 * nx - 1, 2 * ny - 1); fdom.periodic[0] = false; fdom.periodic[1] = false;
        std::vector<Box> fbx;
        for (int bj = 0; bj < fny / fmb; bj++) for (int bi = 0; bi < fnx / fmb; bi++)
            fbx.push_back(Box(2 * ci0 + bi * fmb, 2 * cj0 + bj * fmb, 2 * ci0 + bi * fmb + fmb - 1, 2 * cj0 + bj * fmb + fmb - 1));
        DisjointBoxLayout fgrids(fbx, fdom);
        LevelData<FArrayBox> fphi(fgrids, 1, 1), frhs(fgrids, 1, 0), faCoef(fgrids, 1, 0), fB(fgrids, 1, 1), fPi(fgrids, 1, 1), fzb(fgrids, 1, 1), fmask(fgrids, 1, 1);
        LevelData<FluxBox> fbCoef(fgrids, 1, 0);
        const double fdx = dxv / 2, fdy = dyv / 2;
        for (int k = 0; k < fgrids.size(); k++) {
            const Box g = fgrids[k].grown(1);
            for (int j = g.lo[1]; j <= g.hi[1]; j++)
                for (int i = g.lo[0]; i <= g.hi[0]; i++) {
                    double x = (i + 0.5) * fdx;
                    double H = 6.0 * (std::sqrt(x + 5000.0) - std::sqrt(5000.0)) + 1.0; if (H < 0) H = 0;
                    fB[k](i, j) = 0.01 * (1.0 + 0.4 * std::sin(0.185 * i) * std::cos(0.115 * j));
                    fPi[k](i, j) = 910.0 * 9.8 * H; fzb[k](i, j) = 0.0; fmask[k](i, j) = 1.0;
                    fphi[k](i, j) = 101325.0 / 9800.0 + 1e-3 * hashv(i * 977 + j, -1.0, 1.0);
                }
            const Box &v = fgrids[k];
            for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { frhs[k](i, j) = 5.79e-9; faCoef[k](i, j) = 0.0; }
        }
        // patch-sized global arrays for the oracle (offset 2 ci0, 2 cj0)
        auto patch = [&](const LevelData<FArrayBox> &ld, int g) {
            std::vector<double> a((size_t)(fnx + 2 * g) * (fny + 2 * g), 0.0);
            for (int k = 0; k < ld.size(); k++) {
                Box b = ld.disjointBoxLayout()[k].grown(g);
                for (int j = b.lo[1]; j <= b.hi[1]; j++) for (int i = b.lo[0]; i <= b.hi[0]; i++) {
                    int li = i - 2 * ci0 + g, lj = j - 2 * cj0 + g;
                    if (li >= 0 && li < fnx + 2 * g && lj >= 0 && lj < fny + 2 * g) a[(size_t)lj * (fnx + 2 * g) + li] = ld[k](i, j);
                }
            }
            return a;
        };
        OrAmr2 *A = or_amr2_create(O, nx, ny, dxv, dyv, &obc, &oph, 0.0, -1.0, ci0, cj0, ci1, cj1);
        { auto g = patch(fphi, 0); or_amr2_fine_io(A, OR_F_PHI, g.data(), 0, 1); }
        { auto g = patch(frhs, 0); or_amr2_fine_io(A, OR_F_RHS, g.data(), 0, 1); }
        { auto g = patch(faCoef, 0); or_amr2_fine_io(A, OR_F_ACOEF, g.data(), 0, 1); }
        { auto g = patch(fB, 1); or_amr2_fine_io(A, OR_F_B, g.data(), 1, 1); }
        { auto g = patch(fPi, 1); or_amr2_fine_io(A, OR_F_PI, g.data(), 1, 1); }
        { auto g = patch(fzb, 1); or_amr2_fine_io(A, OR_F_ZB, g.data(), 1, 1); }
        { auto g = patch(fmask, 1); or_amr2_fine_io(A, OR_F_MASK, g.data(), 1, 1); }
        fac.defineFineLevel(fdom, fgrids, faCoef, fbCoef, fB, fPi, fzb, fmask);
        VCAMRNonLinearPoissonOpHIP *fop = fac.AMRnewOp(fdom);
        CHECK(fop != op && fop != nullptr, "AMRnewOp(fine domain) returns the fine-level operator");
        auto same_patch = [&](const LevelData<FArrayBox> &ld, const std::vector<double> &glob) {
            for (int k = 0; k < ld.size(); k++) {
                const Box &v = ld.disjointBoxLayout()[k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) {
                    double a = ld[k](i, j), b = glob[(size_t)(j - 2 * cj0) * fnx + (i - 2 * ci0)];
                    if (memcmp(&a, &b, 8) != 0) { printf("   mismatch at (%d,%d): %.17g vs %.17g\n", i, j, a, b); return false; }
                }
            }
            return true;
        };
        std::vector<double> fg((size_t)fnx * fny);
        // coarse-fine ghosts of the head (relaxNF with 0 sweeps), UpdateOperator with the coarser level, then
        // relaxNF: coarse-fine interpolation from the current base-level head + 2 sweeps
        fop->relaxNF(fphi, &phi, frhs, 0, 0, 0);
        or_amr2_cf_interp_phi(A);
        fop->UpdateOperator(fphi, &phi, 0, 0, false);
        or_amr2_fine_update_operator(A);
        fop->relaxNF(fphi, &phi, frhs, 2, 0, 0);
        or_amr2_cf_interp_phi(A); or_amr2_fine_gsrb(A, 2); or_amr2_fine_io(A, OR_F_PHI, fg.data(), 0, 0);
        CHECK(same_patch(fphi, fg), "relaxNF (coarseFineInterp + levelGSRB x2 on the patch) == oracle");
        // the two-level solve
        HeadSolverParameters sp2(100, true); sp2.max_iter = 5;
        OrSolverParams osp2; memcpy(&osp2, static_cast<suhmo_solver_params_t *>(&sp2), sizeof(osp2));
        std::vector<LevelData<FArrayBox> *> vphi = {&phi, &fphi};
        std::vector<LevelData<FArrayBox> *> vrhs = {&rhs, &frhs};
        std::vector<double> h2, oh2(sp2.max_iter + 2);
        int n2 = fac.solveAMR(vphi, vrhs, sp2, &h2);
        int on2 = or_amr2_solve(A, &osp2, oh2.data());
        or_level_get(O, 0, OR_F_PHI, og.data(), 0); or_amr2_fine_io(A, OR_F_PHI, fg.data(), 0, 0);
        CHECK(n2 == on2 && h2.back() == oh2[on2], "solveAMR: V-cycle count and composite residual == oracle");
        CHECK(same_valid(phi, og, nx), "solveAMR: base-level head == oracle (bitwise)");
        CHECK(same_patch(fphi, fg), "solveAMR: fine-level head == oracle (bitwise)");
        printf("AMR V-cycles %d, composite residual %.3e -> %.3e\n", n2, h2.front(), h2.back());

        // ---- the remaining AMRLevelOp methods on the two levels
        {
            LevelData<FArrayBox> fL(fgrids, 1, 0), cL(grids, 1, 0), cR(grids, 1, 0);
            // AMROperatorNF: coarseFineInterp + applyOpI on the patch
            fop->AMROperatorNF(fL, fphi, phi, false);
            or_amr2_cf_interp_phi(A); or_amr2_fine_apply_op(A, 0); or_amr2_fine_io(A, OR_F_LPHI, fg.data(), 0, 0);
            CHECK(same_patch(fL, fg), "AMROperatorNF == oracle (coarseFineInterp + applyOpI on the patch)");
            // AMROperatorNC = applyOpI + reflux; rhs - that == the oracle's composite residual outside the patch
            op->AMROperatorNC(cL, fphi, phi, false, fop);
            op->AMRResidualNC(cR, fphi, phi, rhs, false, fop);
            or_amr2_residual(A); memcpy(og.data(), or_amr2_coarse_residual(A), sizeof(double) * (size_t)nx * ny);
            bool okc = true, okr = true, changed = false;
            LevelData<FArrayBox> cPlain(grids, 1, 0);
            op->applyOp(cPlain, phi, false);
            for (int k = 0; k < grids.size(); k++) {
                const Box &v = grids[k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) {
                    bool covered = i >= ci0 && i <= ci1 && j >= cj0 && j <= cj1;
                    double mine = -1.0 * cL[k](i, j) + 1.0 * rhs[k](i, j), ref = og[(size_t)j * nx + i];
                    if (!covered && memcmp(&mine, &ref, 8) != 0) okc = false;
                    double r2 = cR[k](i, j);
                    if (!covered && memcmp(&r2, &ref, 8) != 0) okr = false;
                    if (cL[k](i, j) != cPlain[k](i, j)) changed = true;
                }
            }
            CHECK(okc, "AMROperatorNC (applyOpI + reflux): rhs - L == oracle composite residual outside the patch");
            CHECK(okr, "AMRResidualNC == oracle composite residual outside the patch");
            CHECK(changed, "reflux changes L(phi) next to the patch");
            // reflux() on a caller-provided L(phi)
            LevelData<FArrayBox> cL2(grids, 1, 0);
            op->assign(cL2, cPlain);
            op->reflux(fphi, phi, cL2, fop);
            bool okf = true;
            for (int k = 0; k < grids.size(); k++) {
                const Box &v = grids[k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { double a = cL2[k](i, j), b = cL[k](i, j); if (memcmp(&a, &b, 8)) okf = false; }
            }
            CHECK(okf, "applyOp + reflux == AMROperatorNC");
            // AMRProlong: piecewise-constant interpolation of a coarse correction
            LevelData<FArrayBox> fcorr(fgrids, 1, 1), ccorr(grids, 1, 1);
            for (int k = 0; k < grids.size(); k++) { const Box &v = grids[k]; for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) ccorr[k](i, j) = hashv(i * 31 + j, -1.0, 1.0); }
            for (int k = 0; k < fgrids.size(); k++) { const Box &v = fgrids[k]; for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) fcorr[k](i, j) = hashv(i * 17 + j, 0.0, 1.0); }
            LevelData<FArrayBox> fcorr0(fgrids, 1, 1); fop->assign(fcorr0, fcorr);
            fop->AMRProlong(fcorr, ccorr);
            bool okp = true;
            for (int k = 0; k < fgrids.size(); k++) {
                const Box &v = fgrids[k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) {
                    int I = i / 2, J = j / 2, kc = (J / mb) * (nx / mb) + I / mb;
                    double e = fcorr0[k](i, j) + ccorr[kc](I, J), a = fcorr[k](i, j);
                    if (memcmp(&a, &e, 8)) okp = false;
                }
            }
            CHECK(okp, "AMRProlong == PROLONGNL with ratio 2 (piecewise constant)");
            // AMRUpdateResidual: residual <- residual - L(correction) through AMRResidualNF
            LevelData<FArrayBox> fr1(fgrids, 1, 0), fr2(fgrids, 1, 0);
            fop->assign(fr1, frhs); fop->AMRUpdateResidual(fr1, fphi, phi);
            fop->AMRResidualNF(fr2, fphi, phi, frhs, false);
            bool oku = true;
            for (int k = 0; k < fgrids.size(); k++) { const Box &v = fgrids[k]; for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { double a = fr1[k](i, j), b = fr2[k](i, j); if (memcmp(&a, &b, 8)) oku = false; } }
            CHECK(oku, "AMRUpdateResidual == AMRResidualNF with the residual as right-hand side");
            // AMRRestrict (allocates its own scratch) == AMRRestrictS
            LevelData<FArrayBox> rc1(grids, 1, 0), rc2(grids, 1, 0), scr;
            fop->create(scr, frhs);
            fop->AMRRestrict(rc1, frhs, fphi, phi, false);
            fop->AMRRestrictS(rc2, frhs, fphi, phi, scr, false);
            bool oka = true;
            for (int k = 0; k < grids.size(); k++) { const Box &v = grids[k]; for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { double a = rc1[k](i, j), b = rc2[k](i, j); if (memcmp(&a, &b, 8)) oka = false; } }
            CHECK(oka, "AMRRestrict == AMRRestrictS");
            // finerOperatorChanged across the AMR levels: coarse B and bCoef under the patch = averages of the fine ones
            LevelData<FluxBox> fb(fgrids, 1, 0), cb(grids, 1, 0);
            fop->getBCoef(fb);
            op->finerOperatorChanged(*fop, 2);
            op->getBCoef(cb);
            bool okb = true;
            for (int k = 0; k < grids.size(); k++) {
                const Box &v = grids[k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0] + 1; i++) {
                    if (!(i >= ci0 && i <= ci1 + 1 && j >= cj0 && j <= cj1)) continue;
                    int fi = 2 * i, fj = 2 * j, bi = std::min((fi - 2 * ci0) / fmb, fnx / fmb - 1), kf = ((fj - 2 * cj0) / fmb) * (fnx / fmb) + bi;
                    double sm = 0.0; sm = sm + fb[kf][0](fi, fj); sm = sm + fb[kf][0](fi, fj + 1);
                    double e = sm / 2.0, a = cb[k][0](i, j);
                    if (memcmp(&a, &e, 8)) okb = false;
                }
            }
            CHECK(okb, "finerOperatorChanged(fine operator, 2): coarse x-face bCoef under the patch = mean of the two fine faces");
            CHECK(fac.refToFiner(dom) == 2 && fac.refToFiner(fdom) == 1 && fop->refToCoarser() == 2, "refToFiner / refToCoarser");
        }
        or_amr2_destroy(A);
    }


    // ================= all AMR levels at once, as the reference's factory takes them: base + two levels that are unions of boxes
    {
        const int nlev = 3;
        std::vector<std::vector<Box> > lb(nlev);
        lb[0] = bx;
        lb[1] = {Box(64, 32, 127, 95), Box(128, 32, 191, 63), Box(0, 8, 47, 55)};                 // an L-shaped union and a box on the domain side
        lb[2] = {Box(144, 80, 239, 111), Box(144, 112, 207, 175), Box(16, 32, 63, 79)};
        std::vector<DisjointBoxLayout> hg(nlev);
        std::vector<LevelData<FArrayBox> > hphi(nlev), hrhs(nlev), ha(nlev), hB(nlev), hPi(nlev), hzb(nlev), hmask(nlev);
        for (int l = 0; l < nlev; l++) {
            ProblemDomain d; d.dom = Box(0, 0, (nx << l) - 1, (ny << l) - 1); d.periodic[0] = d.periodic[1] = false;
            hg[l] = DisjointBoxLayout(lb[l], d);
            hphi[l].define(hg[l], 1, 1); hrhs[l].define(hg[l], 1, 0); ha[l].define(hg[l], 1, 0);
            hB[l].define(hg[l], 1, 1); hPi[l].define(hg[l], 1, 1); hzb[l].define(hg[l], 1, 1); hmask[l].define(hg[l], 1, 1);
            const double ldx = dxv / (1 << l);
            for (int k = 0; k < hg[l].size(); k++) {
                const Box g = hg[l][k].grown(1);
                for (int j = g.lo[1]; j <= g.hi[1]; j++)
                    for (int i = g.lo[0]; i <= g.hi[0]; i++) {
                        double x = (i + 0.5) * ldx;
                        double H = 6.0 * (std::sqrt(x + 5000.0) - std::sqrt(5000.0)) + 1.0; if (H < 0) H = 0;
                        hB[l][k](i, j) = 0.01 * (1.0 + 0.4 * std::sin(0.37 * i / (1 << l)) * std::cos(0.23 * j / (1 << l)));
                        hPi[l][k](i, j) = 910.0 * 9.8 * H; hzb[l][k](i, j) = 0.0; hmask[l][k](i, j) = 1.0;
                        hphi[l][k](i, j) = 101325.0 / 9800.0 + 1e-3 * hashv(i * (1000 - 23 * l) + j, -1.0, 1.0);
                    }
                const Box &v = hg[l][k];
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) { hrhs[l][k](i, j) = 5.79e-9; ha[l][k](i, j) = 0.0; }
            }
        }
        auto ptrs = [&](std::vector<LevelData<FArrayBox> > &v) { std::vector<LevelData<FArrayBox> *> p; for (auto &x : v) p.push_back(&x); return p; };
        VCAMRNonLinearPoissonOpHIPFactory hf;
        hf.defineHierarchy(dom, hg, dx, bc, 0.0, -1.0, ph, ptrs(ha), ptrs(hB), ptrs(hPi), ptrs(hzb), ptrs(hmask));
        // oracle twin: a fresh base level + the boxes
        OrLevel *O2 = or_level_create(nx, ny, dxv, dyv, mb, &obc, &oph, 0.0, -1.0, 2);
        { auto g = to_global(hphi[0], nx, ny, 0); or_level_set(O2, 0, OR_F_PHI, g.data(), 0); }
        { auto g = to_global(hrhs[0], nx, ny, 0); or_level_set(O2, 0, OR_F_RHS, g.data(), 0); }
        { auto g = to_global(ha[0], nx, ny, 0); or_level_set(O2, 0, OR_F_ACOEF, g.data(), 0); }
        { auto g = to_global(hB[0], nx, ny, 1); or_level_set(O2, 0, OR_F_B, g.data(), 1); }
        { auto g = to_global(hPi[0], nx, ny, 1); or_level_set(O2, 0, OR_F_PI, g.data(), 1); }
        { auto g = to_global(hzb[0], nx, ny, 1); or_level_set(O2, 0, OR_F_ZB, g.data(), 1); }
        { auto g = to_global(hmask[0], nx, ny, 1); or_level_set(O2, 0, OR_F_MASK, g.data(), 1); }
        or_level_build_mg_coefficients(O2);
        std::vector<int> nbox = {0, (int)lb[1].size(), (int)lb[2].size()}, flat;
        for (int l = 1; l < nlev; l++) for (auto &b : lb[l]) flat.insert(flat.end(), {b.lo[0], b.lo[1], b.hi[0], b.hi[1]});
        OrAmrM *M = or_amrm_create(O2, nx, ny, dxv, dyv, &obc, &oph, 0.0, -1.0, nlev, nbox.data(), flat.data());
        for (int l = 1; l < nlev; l++)
            for (int k = 0; k < hg[l].size(); k++) {
                or_amrm_box_io(M, l, k, OR_F_PHI, hphi[l][k].dataPtr(), 1, 1); or_amrm_box_io(M, l, k, OR_F_B, hB[l][k].dataPtr(), 1, 1);
                or_amrm_box_io(M, l, k, OR_F_PI, hPi[l][k].dataPtr(), 1, 1); or_amrm_box_io(M, l, k, OR_F_ZB, hzb[l][k].dataPtr(), 1, 1);
                or_amrm_box_io(M, l, k, OR_F_MASK, hmask[l][k].dataPtr(), 1, 1);
                or_amrm_box_io(M, l, k, OR_F_RHS, hrhs[l][k].dataPtr(), 0, 1); or_amrm_box_io(M, l, k, OR_F_ACOEF, ha[l][k].dataPtr(), 0, 1);
            }
        HeadSolverParameters hsp(100, true);
        hsp.max_iter = 6;
        OrSolverParams hosp; memcpy(&hosp, static_cast<suhmo_solver_params_t *>(&hsp), sizeof(hosp));
        std::vector<Real> hh; std::vector<double> ohh(hsp.max_iter + 2, 0.0);
        auto pphi = ptrs(hphi); std::vector<LevelData<FArrayBox> *> prhs = ptrs(hrhs);
        int hn = hf.solveHierarchy(pphi, prhs, hsp, &hh);
        int ohn = or_amrm_solve(M, &hosp, ohh.data());
        CHECK(hn == ohn && hh.back() == ohh[ohn], "solveHierarchy (base + 2 levels of 3 boxes): V-cycle count and composite residual == oracle");
        { std::vector<double> g0((size_t)nx * ny); or_level_get(O2, 0, OR_F_PHI, g0.data(), 0); CHECK(same_valid(hphi[0], g0, nx), "solveHierarchy: base-level head == oracle (bitwise)"); }
        bool okb = true;
        for (int l = 1; l < nlev; l++)
            for (int k = 0; k < hg[l].size(); k++) {
                const Box &v = hg[l][k];
                std::vector<double> g((size_t)v.size(0) * v.size(1));
                or_amrm_box_io(M, l, k, OR_F_PHI, g.data(), 0, 0);
                for (int j = v.lo[1]; j <= v.hi[1]; j++) for (int i = v.lo[0]; i <= v.hi[0]; i++) {
                    double a = hphi[l][k](i, j), b = g[(size_t)(j - v.lo[1]) * v.size(0) + (i - v.lo[0])];
                    if (memcmp(&a, &b, 8)) okb = false;
                }
            }
        CHECK(okb, "solveHierarchy: the head of every box of levels 1 and 2 == oracle (bitwise)");
        or_amrm_destroy(M);
        or_level_destroy(O2);
    }

    or_level_destroy(O);
    printf(g_fail ? "RESULT: FAIL (%d)\n" : "RESULT: PASS\n", g_fail);
    return g_fail ? 1 : 0;
}
