// suhmo_b2.hip -- boundary B2: per-box Chombo-Fortran kernel symbols (include/suhmo_chf.h).
//
// Each symbol has the name and argument order of the reference's FORT_* call site, takes
// HOST pointers to Fortran-order fabs, stages them through HBM, runs one HIP kernel over the
// box and copies the written fab back.  Arithmetic = the 2-D arm of the cited .ChF
// subroutine, same association (-ffp-contract=off).  Compatibility path only: the hot
// path is the level-batched ABI (suhmo_level.hip / suhmo_gsrb.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../include/suhmo_chf.h"

namespace {

void default_handler(const char *m) { fprintf(stderr, "MAYDAYERROR: %s\n", m); abort(); }
void (*g_handler)(const char *) = default_handler;
#define B2CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof b_, "%s -> %s", #x, hipGetErrorString(e_)); g_handler(b_); return; } } while (0)

struct DF { double *d; int lo0, lo1, n0, n1, nc; };
__device__ __forceinline__ double &A(const DF &f, int i, int j, int n = 0)
{
    return f.d[(size_t)(i - f.lo0) + (size_t)f.n0 * ((size_t)(j - f.lo1) + (size_t)f.n1 * n)];
}
struct BX { int lo0, lo1, n0, n1; };

// bump arena in device memory, reset at the start of every call
struct Arena {
    char *base = nullptr; size_t cap = 0, used = 0;
    double *take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (used + bytes > cap) return nullptr;
        double *p = (double *)(base + used); used += bytes; return p;
    }
    bool reserve(size_t bytes)
    {
        used = 0;
        if (bytes <= cap) return true;
        if (base) (void)hipFree(base);
        cap = bytes * 2;
        if (hipMalloc(&base, cap) != hipSuccess) { base = nullptr; cap = 0; return false; }
        return true;
    }
};
thread_local Arena g_arena;

struct HF { const double *h; int lo0, lo1, hi0, hi1, nc; size_t bytes() const { return (size_t)(hi0 - lo0 + 1) * (hi1 - lo1 + 1) * nc * 8; } };
#define HFAB(a) HF{a, *i##a##lo0, *i##a##lo1, *i##a##hi0, *i##a##hi1, *n##a##comp}
#define HFAB1(a) HF{a, *i##a##lo0, *i##a##lo1, *i##a##hi0, *i##a##hi1, 1}
#define HBOX(b) BX{*i##b##lo0, *i##b##lo1, *i##b##hi0 - *i##b##lo0 + 1, *i##b##hi1 - *i##b##lo1 + 1}

bool stage(std::vector<HF> fabs, DF *out)
{
    size_t tot = 0;
    for (auto &f : fabs) tot += ((f.bytes() + 255) & ~(size_t)255);
    if (!g_arena.reserve(tot)) { g_handler("device arena allocation failed"); return false; }
    for (size_t k = 0; k < fabs.size(); k++) {
        const HF &f = fabs[k];
        double *d = g_arena.take(f.bytes());
        if (hipMemcpy(d, f.h, f.bytes(), hipMemcpyHostToDevice) != hipSuccess) { g_handler("H2D copy failed"); return false; }
        out[k] = DF{d, f.lo0, f.lo1, f.hi0 - f.lo0 + 1, f.hi1 - f.lo1 + 1, f.nc};
    }
    return true;
}
void unstage(const DF &d, double *h) { (void)hipMemcpy(h, d.d, (size_t)d.n0 * d.n1 * d.nc * 8, hipMemcpyDeviceToHost); }
inline dim3 grid(const BX &b) { return dim3((b.n0 + 63) / 64, (b.n1 + 3) / 4); }
#define BLK dim3(64, 4)
#define CELL(b) int i = b.lo0 + blockIdx.x * blockDim.x + threadIdx.x, j = b.lo1 + blockIdx.y * blockDim.y + threadIdx.y; \
                if (i >= b.lo0 + b.n0 || j >= b.lo1 + b.n1) return

__device__ __forceinline__ double lof(const DF &phi, const DF &a, const DF &b0, const DF &b1, const DF &nl, int i, int j, int n,
                                      double alpha, double beta, double rdx, double rdy)
{
    return alpha * A(a, i, j, n) * A(phi, i, j, n)
           - beta * (A(b0, i + 1, j, n) * (A(phi, i + 1, j, n) - A(phi, i, j, n)) * rdx
                     - A(b0, i, j, n) * (A(phi, i, j, n) - A(phi, i - 1, j, n)) * rdx
                     + A(b1, i, j + 1, n) * (A(phi, i, j + 1, n) - A(phi, i, j, n)) * rdy
                     - A(b1, i, j, n) * (A(phi, i, j, n) - A(phi, i, j - 1, n)) * rdy)
           + A(nl, i, j, n);
}

__global__ void kb_gsrb(DF phi, DF rhs, BX r, double rdx, double rdy, double alpha, DF a, double beta, DF b0, DF b1, DF nl, DF dnl, DF lam, int rb)
{
    CELL(r);
    if (((i + j + rb) & 1) != 0) return;          // imin + j + redBlack even, ...OpF.ChF:121-129
    for (int n = 0; n < phi.nc; n++) {
        double l = lof(phi, a, b0, b1, nl, i, j, n, alpha, beta, rdx, rdy);
        double denom = 1.0e-16 + A(lam, i, j, n) + A(dnl, i, j, n);
        A(phi, i, j, n) = A(phi, i, j, n) + (A(rhs, i, j, n) - l) / denom;
    }
}
__global__ void kb_op(DF out, DF phi, DF rhs, int mode, double alpha, DF a, double beta, DF b0, DF b1, DF nl, BX r, double rdx, double rdy)
{
    CELL(r);
    for (int n = 0; n < phi.nc; n++) {
        double l = lof(phi, a, b0, b1, nl, i, j, n, alpha, beta, rdx, rdy);
        A(out, i, j, n) = mode == 0 ? l : A(rhs, i, j, n) - l;
    }
}
// coarse cell owner thread accumulates its 4 fine cells in the Fortran loop order
__global__ void kb_restrictres(DF res, DF phi, DF rhs, double alpha, DF a, double beta, DF b0, DF b1, DF nl, BX rc, BX rf, double rdx, double rdy)
{
    CELL(rc);
    for (int n = 0; n < phi.nc; n++) {
        double acc = A(res, i, j, n);
        for (int b = 0; b < 2; b++) for (int c = 0; c < 2; c++) {
            int fi = 2 * i + c, fj = 2 * j + b;
            if (fi < rf.lo0 || fi >= rf.lo0 + rf.n0 || fj < rf.lo1 || fj >= rf.lo1 + rf.n1) continue;
            acc = acc + (A(rhs, fi, fj, n) - lof(phi, a, b0, b1, nl, fi, fj, n, alpha, beta, rdx, rdy)) / 4.0;
        }
        A(res, i, j, n) = acc;
    }
}
__global__ void kb_restrict(DF c, DF f, BX rc, BX rf)
{
    CELL(rc);
    for (int n = 0; n < f.nc; n++) {
        double acc = A(c, i, j, n);
        for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
            int fi = 2 * i + a, fj = 2 * j + b;
            if (fi < rf.lo0 || fi >= rf.lo0 + rf.n0 || fj < rf.lo1 || fj >= rf.lo1 + rf.n1) continue;
            acc = acc + A(f, fi, fj, n) / 4.0;
        }
        A(c, i, j, n) = acc;
    }
}
__global__ void kb_sumfaces(DF lhs, double beta, DF b, BX r, int dir, double scale)
{
    CELL(r);
    int ii = dir == 0, jj = dir == 1;
    for (int n = 0; n < lhs.nc; n++) {
        double sumVal = A(b, i + ii, j + jj, n) + A(b, i, j, n);
        A(lhs, i, j, n) = A(lhs, i, j, n) + scale * beta * sumVal;
    }
}
__global__ void kb_prolong(DF phi, DF c, BX r, int m)
{
    CELL(r);
    for (int n = 0; n < phi.nc; n++) A(phi, i, j, n) = A(phi, i, j, n) + A(c, i / m, j / m, n);
}
__global__ void kb_prolong2(DF phi, DF c, BX r, int m)
{
    CELL(r);
    const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
    int ic = i / m, jc = j / m, o1 = 2 * (i % 2) - 1, o2 = 2 * (j % 2) - 1;
    for (int n = 0; n < phi.nc; n++) {
        double p = A(phi, i, j, n);
        p = p + fx2 * A(c, ic, jc, n) + f0 * A(c, ic + o1, jc + o2, n);
        p = p + fx1 * (A(c, ic + o1, jc, n) + A(c, ic, jc + o2, n));
        A(phi, i, j, n) = p;
    }
}
__global__ void kb_getflux(DF flux, DF phi, BX r, double beta_dx, int idir)
{
    CELL(r);
    int ii = idir == 0, jj = idir == 1;
    for (int n = 0; n < phi.nc; n++) A(flux, i, j, n) = -(A(phi, i, j, n) - A(phi, i - ii, j - jj, n)) * beta_dx;
}
__global__ void kb_nl(DF phi, DF B, DF IM, DF Pi, DF zb, BX r, DF nlf, DF dnlf, double Ap, double br, double brMax)
{
    CELL(r);
    if (A(IM, i, j) < 0.0) { A(nlf, i, j) = 0.0; A(dnlf, i, j) = 0.0; return; }
    double b = A(B, i, j);
    double N = A(Pi, i, j) - 1000.0 * 9.8 * (A(phi, i, j) - A(zb, i, j));
    double nl = -Ap * b * N * N * N;
    double dnl = 3.0 * Ap * b * 1000.0 * 9.8 * N * N;
    if (br > b) { nl = nl * (1.0 - (br - b) / br); dnl = dnl * b / br; }
    if (brMax < b) { nl = nl * (1.0 - (brMax - b) / brMax); dnl = dnl * b / brMax; }
    A(nlf, i, j) = nl; A(dnlf, i, j) = dnl;
}
__global__ void kb_re(DF B, DF g, BX r, DF Re, double om, double nu)
{
    CELL(r);
    double s = sqrt(A(g, i, j, 0) * A(g, i, j, 0) + A(g, i, j, 1) * A(g, i, j, 1));
    double discr = 1.0 + 4.0 * om * (A(B, i, j) * A(B, i, j) * A(B, i, j) * 9.8 * s) / (12.0 * nu * nu);
    A(Re, i, j) = (-1.0 + sqrt(discr)) / (2.0 * om);
}
__global__ void kb_bcoeff(DF B, DF Re, BX r, DF bc, DF IM, double om, double nu, int cut)
{
    CELL(r);
    double num_q = -(A(B, i, j) * A(B, i, j) * A(B, i, j) * 9.8);
    double denom_q = 12.0 * nu * (1.0 + om * A(Re, i, j));
    A(bc, i, j) = (A(IM, i, j) < 0.0 && cut > 0) ? 0.0 : num_q / denom_q;
}
__global__ void kb_macgrad(DF eg, DF mask, DF phi, BX r, double factor, int dir, int hasMask)
{
    CELL(r);
    int ii = dir == 0, jj = dir == 1;
    double v = factor * (A(phi, i, j) - A(phi, i - ii, j - jj));
    if (hasMask > 0 && ((A(mask, i, j) < 1e-6) || (A(mask, i - ii, j - jj) < 1e-6))) v = 0.0;
    A(eg, i, j) = v;
}
__global__ void kb_bcfill(DF phi, BX r, int dir, int hiLo, int mode)
{
    CELL(r);
    int off = hiLo == 0 ? -1 : 1, i0 = off * (dir == 0), i1 = off * (dir == 1);
    for (int n = 0; n < phi.nc; n++) {
        double v;
        if (mode == 0) v = 2.0 * A(phi, i - i0, j - i1, n) - A(phi, i - 2 * i0, j - 2 * i1, n);
        else if (mode == 1) v = A(phi, i - i0, j - i1, n);
        else v = 0.0;
        A(phi, i, j, n) = v;
    }
}
__global__ void kb_div(DF u, DF div, BX r, double one_on_dx, int idir)
{
    CELL(r);
    int h0 = idir == 0, h1 = idir == 1;
    for (int n = 0; n < div.nc; n++) A(div, i, j, n) = A(div, i, j, n) + one_on_dx * (A(u, i + h0, j + h1, n) - A(u, i, j, n));
}

bool ncomp_ok(int a, int b) { if (a != b) { g_handler("ncomp mismatch"); return false; } return true; }
__global__ void kb_qw(DF aB, DF aRe, DF g, BX r, DF Qw, double omega, double nu)
{
    CELL(r);
    double b = A(aB, i, j);
    double num_q = -(b * b * b * 9.8 * A(g, i, j));
    double denom_q = 12.0 * nu * (1.0 + omega * A(aRe, i, j));
    A(Qw, i, j) = num_q / denom_q;
}
__global__ void kb_scaprod(DF a, DF b1, DF b2, BX r, DF p1, DF p2)
{
    CELL(r);
    A(p1, i, j) = A(a, i, j) * A(b1, i, j);
    A(p2, i, j) = A(a, i, j) * A(b2, i, j);
}
__global__ void kb_dcoeff(BX r, DF D, double rho, DF mr, DF b, DF im, int cutOffB)
{
    CELL(r);
    if (A(im, i, j) < 0.0 && cutOffB > 0) A(D, i, j) = 0.0;
    else A(D, i, j) = fmax(A(b, i, j) * A(mr, i, j) / rho, 5.0e-6);
}
__global__ void kb_difterm(DF phi, BX r, double dxinv0, double dxinv1, DF Dt, DF d0, DF d1)
{
    CELL(r);
    A(Dt, i, j) = (A(d0, i + 1, j) * (A(phi, i + 1, j) - A(phi, i, j)) * dxinv0 - A(d0, i, j) * (A(phi, i, j) - A(phi, i - 1, j)) * dxinv0
                   + A(d1, i, j + 1) * (A(phi, i, j + 1) - A(phi, i, j)) * dxinv1 - A(d1, i, j) * (A(phi, i, j) - A(phi, i, j - 1)) * dxinv1);
}
__global__ void kb_tvrecharge(DF zs, BX r, DF out, double TK, double bg)
{
    CELL(r);
    const double ddf = 0.01 / 86400., dT_dZ = -0.0075;
    A(out, i, j) = fmax(ddf * (TK + A(zs, i, j) * dT_dZ), 0.0) + bg;
}
} // namespace

extern "C" {

void suhmo_chf_set_error_handler(void (*h)(const char *)) { g_handler = h ? h : default_handler; }

void gsrbhelmholtzvcnl2d_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(rhs), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx),
                          SUHMO_CHF_CONST_REAL(alpha), SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta),
                          SUHMO_CHF_CONST_FRA(bCoef0), SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc),
                          SUHMO_CHF_CONST_FRA(nlDfunc), SUHMO_CHF_CONST_FRA(lambda), SUHMO_CHF_CONST_INT(redBlack))
{
    if (!ncomp_ok(*nphicomp, *nrhscomp) || !ncomp_ok(*nphicomp, *nbCoef0comp) || !ncomp_ok(*nphicomp, *nbCoef1comp)) return;  // :87-106
    DF d[8];
    if (!stage({HFAB(phi), HFAB(rhs), HFAB(aCoef), HFAB(bCoef0), HFAB(bCoef1), HFAB(nlfunc), HFAB(nlDfunc), HFAB(lambda)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_gsrb, grid(r), BLK, 0, 0, d[0], d[1], r, 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), *alpha, d[2], *beta,
                       d[3], d[4], d[5], d[6], d[7], *redBlack);
    unstage(d[0], phi);
}

void vcnlcomputeop2d_(SUHMO_CHF_FRA(lofphi), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_REAL(alpha), SUHMO_CHF_CONST_FRA(aCoef),
                      SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoef0), SUHMO_CHF_CONST_FRA(bCoef1),
                      SUHMO_CHF_CONST_FRA(nlfunc), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx))
{
    if (!ncomp_ok(*nphicomp, *nlofphicomp)) return;
    DF d[6];
    if (!stage({HFAB(lofphi), HFAB(phi), HFAB(aCoef), HFAB(bCoef0), HFAB(bCoef1), HFAB(nlfunc)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_op, grid(r), BLK, 0, 0, d[0], d[1], d[1], 0, *alpha, d[2], *beta, d[3], d[4], d[5], r,
                       1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]));
    unstage(d[0], lofphi);
}

void vcnlcomputeres2d_(SUHMO_CHF_FRA(res), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(rhs), SUHMO_CHF_CONST_REAL(alpha),
                       SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoef0),
                       SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx))
{
    if (!ncomp_ok(*nphicomp, *nrescomp)) return;
    DF d[7];
    if (!stage({HFAB(res), HFAB(phi), HFAB(rhs), HFAB(aCoef), HFAB(bCoef0), HFAB(bCoef1), HFAB(nlfunc)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_op, grid(r), BLK, 0, 0, d[0], d[1], d[2], 1, *alpha, d[3], *beta, d[4], d[5], d[6], r,
                       1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]));
    unstage(d[0], res);
}

static void restrict_impl(double *phiCoarse, const int *l0, const int *l1, const int *h0, const int *h1, const int *nc,
                          const double *phiFine, const int *fl0, const int *fl1, const int *fh0, const int *fh1, const int *fnc,
                          const int *r0, const int *r1, const int *r2, const int *r3)
{
    DF d[2];
    if (!stage({HF{phiCoarse, *l0, *l1, *h0, *h1, *nc}, HF{phiFine, *fl0, *fl1, *fh0, *fh1, *fnc}}, d)) return;
    BX rf{*r0, *r1, *r2 - *r0 + 1, *r3 - *r1 + 1};
    BX rc{*r0 / 2, *r1 / 2, *r2 / 2 - *r0 / 2 + 1, *r3 / 2 - *r1 / 2 + 1};
    hipLaunchKernelGGL(kb_restrict, grid(rc), BLK, 0, 0, d[0], d[1], rc, rf);
    unstage(d[0], phiCoarse);
}
void restrictvcnl_(SUHMO_CHF_FRA(phiCoarse), SUHMO_CHF_CONST_FRA(phiFine), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REAL(dx))
{
    (void)dx;
    restrict_impl(phiCoarse, iphiCoarselo0, iphiCoarselo1, iphiCoarsehi0, iphiCoarsehi1, nphiCoarsecomp, phiFine, iphiFinelo0,
                  iphiFinelo1, iphiFinehi0, iphiFinehi1, nphiFinecomp, iregionlo0, iregionlo1, iregionhi0, iregionhi1);
}
void restrictnl_(SUHMO_CHF_FRA(phiCoarse), SUHMO_CHF_CONST_FRA(phiFine), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REAL(dx))
{
    (void)dx;
    restrict_impl(phiCoarse, iphiCoarselo0, iphiCoarselo1, iphiCoarsehi0, iphiCoarsehi1, nphiCoarsecomp, phiFine, iphiFinelo0,
                  iphiFinelo1, iphiFinehi0, iphiFinehi1, nphiFinecomp, iregionlo0, iregionlo1, iregionhi0, iregionhi1);
}

void restrictresvcnl2d_(SUHMO_CHF_FRA(res), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(rhs), SUHMO_CHF_CONST_REAL(alpha),
                        SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoef0),
                        SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx))
{
    DF d[7];
    if (!stage({HFAB(res), HFAB(phi), HFAB(rhs), HFAB(aCoef), HFAB(bCoef0), HFAB(bCoef1), HFAB(nlfunc)}, d)) return;
    BX rf = HBOX(region);
    BX rc{rf.lo0 / 2, rf.lo1 / 2, (rf.lo0 + rf.n0 - 1) / 2 - rf.lo0 / 2 + 1, (rf.lo1 + rf.n1 - 1) / 2 - rf.lo1 / 2 + 1};
    hipLaunchKernelGGL(kb_restrictres, grid(rc), BLK, 0, 0, d[0], d[1], d[2], *alpha, d[3], *beta, d[4], d[5], d[6], rc, rf,
                       1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]));
    unstage(d[0], res);
}

void sumfacesnl_(SUHMO_CHF_FRA(lhs), SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoefs), SUHMO_CHF_BOX(box),
                 SUHMO_CHF_CONST_INT(dir), SUHMO_CHF_CONST_REAL(scale))
{
    DF d[2];
    if (!stage({HFAB(lhs), HFAB(bCoefs)}, d)) return;
    BX r = HBOX(box);
    hipLaunchKernelGGL(kb_sumfaces, grid(r), BLK, 0, 0, d[0], *beta, d[1], r, *dir, *scale);
    unstage(d[0], lhs);
}

void prolongnl_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(coarse), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_INT(m))
{
    DF d[2];
    if (!stage({HFAB(phi), HFAB(coarse)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_prolong, grid(r), BLK, 0, 0, d[0], d[1], r, *m);
    unstage(d[0], phi);
}
void prolong_2_nl_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(coarse), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_INT(m))
{
    DF d[2];
    if (!stage({HFAB(phi), HFAB(coarse)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_prolong2, grid(r), BLK, 0, 0, d[0], d[1], r, *m);
    unstage(d[0], phi);
}
void newgetfluxnl_(SUHMO_CHF_FRA(flux), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_BOX(box), SUHMO_CHF_CONST_REAL(beta_dx), SUHMO_CHF_CONST_INT(a_idir))
{
    DF d[2];
    if (!stage({HFAB(flux), HFAB(phi)}, d)) return;
    BX r = HBOX(box);
    hipLaunchKernelGGL(kb_getflux, grid(r), BLK, 0, 0, d[0], d[1], r, *beta_dx, *a_idir);
    unstage(d[0], flux);
}

void computenonlinearterms_(SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(IM), SUHMO_CHF_CONST_FRA(aPi),
                            SUHMO_CHF_CONST_FRA(aZb), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(nlfunc), SUHMO_CHF_FRA(dnlfunc),
                            SUHMO_CHF_CONST_REAL(Aparam), SUHMO_CHF_CONST_REAL(brparam), SUHMO_CHF_CONST_REAL(brparamMax))
{
    DF d[7];
    if (!stage({HFAB(phi), HFAB(aB), HFAB(IM), HFAB(aPi), HFAB(aZb), HFAB(nlfunc), HFAB(dnlfunc)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_nl, grid(r), BLK, 0, 0, d[0], d[1], d[2], d[3], d[4], r, d[5], d[6], *Aparam, *brparam, *brparamMax);
    unstage(d[5], nlfunc); unstage(d[6], dnlfunc);
}
void computeqw_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(aRe), SUHMO_CHF_CONST_FRA(agradH), SUHMO_CHF_BOX(region),
                SUHMO_CHF_FRA(Qw), SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam))
{
    DF d[4];
    if (!stage({HFAB(aB), HFAB(aRe), HFAB(agradH), HFAB(Qw)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_qw, grid(r), BLK, 0, 0, d[0], d[1], d[2], r, d[3], *omegaparam, *nuparam);
    unstage(d[3], Qw);
}
void computescaprod_(SUHMO_CHF_CONST_FRA(vara), SUHMO_CHF_CONST_FRA(var1b), SUHMO_CHF_CONST_FRA(var2b), SUHMO_CHF_BOX(region),
                     SUHMO_CHF_FRA(prod1), SUHMO_CHF_FRA(prod2))
{
    DF d[5];
    if (!stage({HFAB(vara), HFAB(var1b), HFAB(var2b), HFAB(prod1), HFAB(prod2)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_scaprod, grid(r), BLK, 0, 0, d[0], d[1], d[2], r, d[3], d[4]);
    unstage(d[3], prod1); unstage(d[4], prod2);
}
void computedcoeff_(SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Dcoeff), SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_CONST_REAL(rho),
                    SUHMO_CHF_FRA(MRec), SUHMO_CHF_FRA(Bec), SUHMO_CHF_FRA(IMec), SUHMO_CHF_INT(cutOffB))
{
    (void)dx;
    DF d[4];
    if (!stage({HFAB(Dcoeff), HFAB(MRec), HFAB(Bec), HFAB(IMec)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_dcoeff, grid(r), BLK, 0, 0, r, d[0], *rho, d[1], d[2], d[3], *cutOffB);
    unstage(d[0], Dcoeff);
}
void computedifterm2d_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_FRA(Dterm),
                       SUHMO_CHF_CONST_FRA(Dcoef0), SUHMO_CHF_CONST_FRA(Dcoef1))
{
    DF d[4];
    if (!stage({HFAB(phi), HFAB(Dterm), HFAB(Dcoef0), HFAB(Dcoef1)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_difterm, grid(r), BLK, 0, 0, d[0], r, 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), d[1], d[2], d[3]);
    unstage(d[1], Dterm);
}
void compute_timevaryingrecharge_(SUHMO_CHF_CONST_FRA(aZs), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Recharge),
                                  SUHMO_CHF_CONST_REAL(TK), SUHMO_CHF_CONST_REAL(BackgroundInput))
{
    DF d[2];
    if (!stage({HFAB(aZs), HFAB(Recharge)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_tvrecharge, grid(r), BLK, 0, 0, d[0], r, d[1], *TK, *BackgroundInput);
    unstage(d[1], Recharge);
}
void computere_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(agradH), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Re),
                SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam))
{
    DF d[3];
    if (!stage({HFAB(aB), HFAB(agradH), HFAB(Re)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_re, grid(r), BLK, 0, 0, d[0], d[1], r, d[2], *omegaparam, *nuparam);
    unstage(d[2], Re);
}
void computebcoeff_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(aRe), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Bcoeff),
                    SUHMO_CHF_CONST_FRA(IMec), SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam), SUHMO_CHF_INT(cutOffB))
{
    DF d[4];
    if (!stage({HFAB(aB), HFAB(aRe), HFAB(Bcoeff), HFAB(IMec)}, d)) return;
    BX r = HBOX(region);
    hipLaunchKernelGGL(kb_bcoeff, grid(r), BLK, 0, 0, d[0], d[1], r, d[2], d[3], *omegaparam, *nuparam, *cutOffB);
    unstage(d[2], Bcoeff);
}
void newmacgrad_(SUHMO_CHF_FRA1(edgeGrad), SUHMO_CHF_FRA1(mask), SUHMO_CHF_FRA1(phi), SUHMO_CHF_BOX(edgeGrid),
                 SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hasMask), SUHMO_CHF_INT(edgeDir))
{
    if (*dir != *edgeDir) { g_handler("newmacgrad_: only the normal-derivative branch (dir == edgeDir) is on the hot path"); return; }
    DF d[3];
    if (!stage({HFAB1(edgeGrad), HFAB1(mask), HFAB1(phi)}, d)) return;
    BX r = HBOX(edgeGrid);
    hipLaunchKernelGGL(kb_macgrad, grid(r), BLK, 0, 0, d[0], d[1], d[2], r, 1.0 / dx[*dir], *dir, *hasMask);
    unstage(d[0], edgeGrad);
}
static void bc_impl(double *phi, const int *l0, const int *l1, const int *h0, const int *h1, const int *nc, const int *b0,
                    const int *b1, const int *b2, const int *b3, int dir, int hiLo, int mode)
{
    DF d[1];
    if (!stage({HF{phi, *l0, *l1, *h0, *h1, *nc}}, d)) return;
    BX r{*b0, *b1, *b2 - *b0 + 1, *b3 - *b1 + 1};
    hipLaunchKernelGGL(kb_bcfill, grid(r), BLK, 0, 0, d[0], r, dir, hiLo, mode);
    unstage(d[0], phi);
}
void simpleextrapbc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo))
{ bc_impl(phi, iphilo0, iphilo1, iphihi0, iphihi1, nphicomp, ibcboxlo0, ibcboxlo1, ibcboxhi0, ibcboxhi1, *dir, *hiLo, 0); }
void simplecopybc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo))
{ bc_impl(phi, iphilo0, iphilo1, iphihi0, iphihi1, nphicomp, ibcboxlo0, ibcboxlo1, ibcboxhi0, ibcboxhi1, *dir, *hiLo, 1); }
void nullbc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo))
{ bc_impl(phi, iphilo0, iphilo1, iphihi0, iphihi1, nphicomp, ibcboxlo0, ibcboxlo1, ibcboxhi0, ibcboxhi1, *dir, *hiLo, 2); }
void divergence_(SUHMO_CHF_CONST_FRA(uEdge), SUHMO_CHF_FRA(div), SUHMO_CHF_BOX(gridInt), SUHMO_CHF_CONST_REAL(dx), SUHMO_CHF_INT(idir))
{
    DF d[2];
    if (!stage({HFAB(uEdge), HFAB(div)}, d)) return;
    BX r = HBOX(gridInt);
    hipLaunchKernelGGL(kb_div, grid(r), BLK, 0, 0, d[0], d[1], r, 1.0 / *dx, *idir);
    unstage(d[1], div);
}

} // extern "C"
