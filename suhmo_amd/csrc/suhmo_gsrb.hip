// suhmo_gsrb.hip -- nonlinear variable-coefficient Gauss-Seidel red-black relaxation.
//
// Replaces VCAMRNonLinearPoissonOp::levelGSRB (src/VCAMRNonLinearPoissonOp.cpp:654-760):
//   per colour pass: exchange -> mixBCValues -> NonLinear_level -> GSRBHELMHOLTZVCNL2D
// with one fused kernel per colour pass over the whole level canvas: the physical BC is
// evaluated on the fly (common.h phiW/E/S/N), the nonlinear term and its derivative
// (COMPUTENONLINEARTERMS) and the relaxation coefficient lambda (SUMFACESNL) are
// recomputed in registers, so per sweep only phi, rhs, bx, by, B, Pi, zb, mask stream
// through HBM.  Colour rule: cell (i,j) is updated in pass p iff (i + j_global + p) is even
// (src/VCAMRNonLinearPoissonOpF.ChF:121-129).
#include "suhmo_common.h"

// ---- variant 0: one thread per active-colour cell (reference kernel for the others) ----
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_gsrb_pass_simple(DV v, FP fp, suhmo_phys_t ph, int pass)
{
    int j = blockIdx.y * blockDim.y + threadIdx.y;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.ny) return;
    int i = 2 * t + ((j + v.j0 + pass) & 1);
    if (i >= v.nx) return;
    double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    double c = phi[idx];
    double e = phiE(v, phi, idx, i, c, false), w = phiW(v, phi, idx, i, c, false);
    double n = phiN(v, phi, idx, j, c, false), s = phiS(v, phi, idx, j, c, false);
    double bxW = fp.f[SUHMO_F_BX][idx], bxE = fp.f[SUHMO_F_BX][idx + 1];
    double byS = fp.f[SUHMO_F_BY][idx], byN = fp.f[SUHMO_F_BY][idx + v.P];
    double nl, dnl;
    nl_terms(ph, c, fp.f[SUHMO_F_B][idx], fp.f[SUHMO_F_PI][idx], fp.f[SUHMO_F_ZB][idx],
             fp.f[SUHMO_F_MASK][idx], nl, dnl);
    double aterm = HAS_ALPHA ? v.alpha * fp.f[SUHMO_F_ACOEF][idx] : v.alpha;
    double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, byN, byS, nl);
    double lam = lambda_cell(v, aterm, bxE, bxW, byN, byS);
    double denom = 1.0e-16 + lam + dnl;                      // ...OpF.ChF:154
    phi[idx] = c + (fp.f[SUHMO_F_RHS][idx] - lofphi) / denom; // :156
}

static int launch_simple(suhmo_level *L, int depth, int pass, hipStream_t st)
{
    Depth &D = L->d[depth];
    dim3 blk(64, 4), grd(((D.v.nx + 1) / 2 + 63) / 64, (D.v.ny + 3) / 4);
    if (D.v.alpha != 0.0)
        hipLaunchKernelGGL(k_gsrb_pass_simple<true>, grd, blk, 0, st, D.v, D.fp, L->ph, pass);
    else
        hipLaunchKernelGGL(k_gsrb_pass_simple<false>, grd, blk, 0, st, D.v, D.fp, L->ph, pass);
    return 0;
}

int suhmo_launch_gsrb(suhmo_level *L, int depth, int sweeps, hipStream_t st)
{
    Depth &D = L->d[depth];
    for (int it = 0; it < sweeps; it++) {
        ProfEv pe{};
        bool prof = L->prof_on && depth == 0;
        if (prof) {
            HIPCHK(hipEventCreate(&pe.a)); HIPCHK(hipEventCreate(&pe.b));
            HIPCHK(hipEventRecord(pe.a, st));
        }
        for (int pass = 0; pass < 2; pass++) {
            if (L->ex && (D.v.ext[0] || D.v.ext[1])) {
                int rc = L->ex(L->user, L, depth, SUHMO_F_PHI, (suhmo_stream_t)st);
                if (rc) return rc;
            }
            launch_simple(L, depth, pass, st);
        }
        if (prof) {
            HIPCHK(hipEventRecord(pe.b, st));
            pe.cells = (long)D.v.nx * D.v.ny;
            L->prof.push_back(pe);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}
