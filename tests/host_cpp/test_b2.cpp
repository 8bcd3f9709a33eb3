// test_b2.cpp -- boundary B2: every per-box Chombo-Fortran symbol exported by
// libsuhmo_hip.so (include/suhmo_chf.h) is called the way Chombo's FORT_* macros call it
// (pointers to scalars, fab = pointer + lo/hi + ncomp) on boxes with non-zero, negative
// offsets, and compared BITWISE with the oracle's restatement of the same subroutine.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/suhmo_chf.h"
#include "../../oracle/suhmo_oracle.h"

static int g_fail = 0;
static unsigned g_seed = 12345u;
static double rnd(double a, double b) { g_seed = g_seed * 1664525u + 1013904223u; return a + (b - a) * (double)(g_seed >> 8) / 16777216.0; }

struct Fab {
    std::vector<double> v; int lo0, lo1, hi0, hi1, nc;
    Fab(int l0, int l1, int h0, int h1, int n, double a, double b) : lo0(l0), lo1(l1), hi0(h0), hi1(h1), nc(n)
    { v.resize((size_t)(h0 - l0 + 1) * (h1 - l1 + 1) * n); for (auto &x : v) x = rnd(a, b); }
    OrFab o() { return OrFab{v.data(), lo0, lo1, hi0, hi1, nc}; }
};
#define F(f) f.v.data(), &f.lo0, &f.lo1, &f.hi0, &f.hi1, &f.nc
#define F1(f) f.v.data(), &f.lo0, &f.lo1, &f.hi0, &f.hi1
#define BOXP(b) &b.lo0, &b.lo1, &b.hi0, &b.hi1
static void cmp(const char *name, const Fab &a, const Fab &b)
{
    bool ok = a.v.size() == b.v.size() && memcmp(a.v.data(), b.v.data(), a.v.size() * 8) == 0;
    printf("%s %s\n", ok ? "ok:  " : "FAIL:", name);
    if (!ok) g_fail++;
}

int main()
{
    // a 24 x 18 box at offset (-5, 7), 1 ghost; faces surroundingNodes
    OrBox reg{-5, 7, 18, 24};
    const int g = 1;
    double dx[2] = {3.0, 2.0}, alpha = 0.7, beta = -1.0;
    Fab phi(reg.lo0 - g, reg.lo1 - g, reg.hi0 + g, reg.hi1 + g, 1, 5.0, 900.0);
    Fab rhs(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, -1e-5, 1e-5), a(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0.0, 1.0);
    Fab b0(reg.lo0, reg.lo1, reg.hi0 + 1, reg.hi1, 1, -1.0, -0.05), b1(reg.lo0, reg.lo1, reg.hi0, reg.hi1 + 1, 1, -1.0, -0.05);
    Fab nl(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, -1e-5, 1e-5), dnl(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0.0, 1e-7);
    Fab lam(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0.01, 1.0);

    for (int rb = 0; rb < 2; rb++) {
        Fab p1 = phi, p2 = phi; OrFab op = p2.o(), orhs = rhs.o(), oa = a.o(), ob0 = b0.o(), ob1 = b1.o(), onl = nl.o(), odnl = dnl.o(), olam = lam.o();
        gsrbhelmholtzvcnl2d_(F(p1), F(rhs), BOXP(reg), dx, &alpha, F(a), &beta, F(b0), F(b1), F(nl), F(dnl), F(lam), &rb);
        or_gsrbhelmholtzvcnl2d(&op, &orhs, reg, dx, alpha, &oa, beta, &ob0, &ob1, &onl, &odnl, &olam, rb);
        cmp(rb ? "gsrbhelmholtzvcnl2d_ (black)" : "gsrbhelmholtzvcnl2d_ (red)", p1, p2);
    }
    {
        Fab l1(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0, 0), l2 = l1; OrFab ol = l2.o(), op = phi.o(), oa = a.o(), ob0 = b0.o(), ob1 = b1.o(), onl = nl.o(), orhs = rhs.o();
        vcnlcomputeop2d_(F(l1), F(phi), &alpha, F(a), &beta, F(b0), F(b1), F(nl), BOXP(reg), dx);
        or_vcnlcomputeop2d(&ol, &op, alpha, &oa, beta, &ob0, &ob1, &onl, reg, dx);
        cmp("vcnlcomputeop2d_", l1, l2);
        vcnlcomputeres2d_(F(l1), F(phi), F(rhs), &alpha, F(a), &beta, F(b0), F(b1), F(nl), BOXP(reg), dx);
        or_vcnlcomputeres2d(&ol, &op, &orhs, alpha, &oa, beta, &ob0, &ob1, &onl, reg, dx);
        cmp("vcnlcomputeres2d_", l1, l2);
        Fab s1 = lam, s2 = lam; OrFab os = s2.o(); int dir = 1; double scale = 1.0 / (dx[1] * dx[1]);
        sumfacesnl_(F(s1), &beta, F(b1), BOXP(reg), &dir, &scale);
        or_sumfacesnl(&os, beta, &ob1, reg, dir, scale);
        cmp("sumfacesnl_", s1, s2);
    }
    {   // restriction kernels work in the shifted (0-origin) index space, CHF_FRA_SHIFT
        OrBox r0{0, 0, 23, 17};
        Fab pf(-1, -1, 24, 18, 1, 5.0, 900.0), rf(0, 0, 23, 17, 1, -1e-5, 1e-5), af(0, 0, 23, 17, 1, 0, 1), nf(0, 0, 23, 17, 1, -1e-5, 1e-5);
        Fab c0(0, 0, 24, 17, 1, -1, -0.05), c1(0, 0, 23, 18, 1, -1, -0.05);
        Fab rc1(0, 0, 11, 8, 1, 0, 0), rc2 = rc1; double dxs = dx[0];
        OrFab orc = rc2.o(), opf = pf.o(), orf = rf.o(), oaf = af.o(), oc0 = c0.o(), oc1 = c1.o(), onf = nf.o();
        restrictresvcnl2d_(F(rc1), F(pf), F(rf), &alpha, F(af), &beta, F(c0), F(c1), F(nf), BOXP(r0), dx);
        or_restrictresvcnl2d(&orc, &opf, &orf, alpha, &oaf, beta, &oc0, &oc1, &onf, r0, dx);
        cmp("restrictresvcnl2d_", rc1, rc2);
        Fab q1(-1, -1, 12, 9, 1, 0, 0), q2 = q1; OrFab oq = q2.o();
        restrictvcnl_(F(q1), F(pf), BOXP(r0), &dxs);
        or_restrictvcnl(&oq, &opf, r0);
        cmp("restrictvcnl_", q1, q2);
        Fab q3(-1, -1, 12, 9, 1, 0, 0), q4 = q3; OrFab oq4 = q4.o();
        restrictnl_(F(q3), F(pf), BOXP(r0), &dxs);
        or_restrictvcnl(&oq4, &opf, r0);
        cmp("restrictnl_", q3, q4);
        Fab cc(-1, -1, 12, 9, 1, -1.0, 1.0); OrFab occ = cc.o(); int m = 2;
        Fab f1 = pf, f2 = pf; OrFab of2 = f2.o();
        prolongnl_(F(f1), F(cc), BOXP(r0), &m); or_prolongnl(&of2, &occ, r0, m);
        cmp("prolongnl_", f1, f2);
        prolong_2_nl_(F(f1), F(cc), BOXP(r0), &m); or_prolong_2_nl(&of2, &occ, r0, m);
        cmp("prolong_2_nl_", f1, f2);
    }
    {
        OrBox fb{reg.lo0, reg.lo1, reg.hi0 + 1, reg.hi1};
        Fab fl1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), fl2 = fl1; OrFab ofl = fl2.o(), op = phi.o(); double bdx = -1.0 / 3.0; int idir = 0;
        newgetfluxnl_(F(fl1), F(phi), BOXP(fb), &bdx, &idir); or_newgetfluxnl(&ofl, &op, fb, bdx, idir);
        cmp("newgetfluxnl_", fl1, fl2);
    }
    {
        OrPhys ph = {5e-25, 1e-3, 1.787e-6, 0.0125, 0.03, 9800.0, 9.8, 1, 1, 1};
        Fab B(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, 0.002, 0.05), Pi(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, 1e5, 1.3e7);
        Fab zb(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, 0.0, 50.0), IM(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, -0.2, 1.0);
        Fab n1(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0, 0), d1 = n1, n2 = n1, d2 = n1;
        OrFab op = phi.o(), oB = B.o(), oIM = IM.o(), oPi = Pi.o(), ozb = zb.o(), on2 = n2.o(), od2 = d2.o();
        computenonlinearterms_(F(phi), F(B), F(IM), F(Pi), F(zb), BOXP(reg), F(n1), F(d1), &ph.A, &ph.cutOffbr, &ph.maxOffbr);
        or_computenonlinearterms(&op, &oB, &oIM, &oPi, &ozb, reg, &on2, &od2, &ph);
        cmp("computenonlinearterms_ (nl)", n1, n2); cmp("computenonlinearterms_ (dnl)", d1, d2);
        OrBox gb{reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1};
        Fab gH(gb.lo0, gb.lo1, gb.hi0, gb.hi1, 2, -1e-2, 1e-2), R1(gb.lo0, gb.lo1, gb.hi0, gb.hi1, 1, 0, 0), R2 = R1;
        OrFab ogH = gH.o(), oR2 = R2.o();
        computere_(F(B), F(gH), BOXP(gb), F(R1), &ph.omega, &ph.nu); or_computere(&oB, &ogH, gb, &oR2, &ph);
        cmp("computere_", R1, R2);
        OrBox fb{reg.lo0, reg.lo1, reg.hi0 + 1, reg.hi1};
        Fab Bec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0.002, 0.05), Rec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0.0, 4000.0), Mec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, -1.0, 1.0);
        Fab bc1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), bc2 = bc1; OrFab oBec = Bec.o(), oRec = Rec.o(), oMec = Mec.o(), obc2 = bc2.o();
        computebcoeff_(F(Bec), F(Rec), BOXP(fb), F(bc1), F(Mec), &ph.omega, &ph.nu, &ph.cutOffB);
        or_computebcoeff(&oBec, &oRec, fb, &obc2, &oMec, &ph);
        cmp("computebcoeff_", bc1, bc2);
        for (int hasMask = 0; hasMask < 2; hasMask++) {
            Fab e1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), e2 = e1; OrFab oe2 = e2.o(); int dir = 0, edgeDir = 0, hm = hasMask;
            newmacgrad_(F1(e1), F1(IM), F1(phi), BOXP(fb), dx, &dir, &hm, &edgeDir);
            or_newmacgrad(&oe2, &oIM, &op, fb, dx, dir, hasMask);
            cmp(hasMask ? "newmacgrad_ (masked)" : "newmacgrad_", e1, e2);
        }
    }
    {
        Fab p1 = phi, p2 = phi; OrFab op2 = p2.o();
        OrBox lo{reg.lo0 - 1, reg.lo1 - 1, reg.lo0 - 1, reg.hi1 + 1}, hi{reg.lo0, reg.hi1 + 1, reg.hi0, reg.hi1 + 1};
        int d0 = 0, d1 = 1, s0 = 0, s1 = 1;
        simpleextrapbc_(F(p1), BOXP(lo), &d0, &s0); or_simpleextrapbc(&op2, lo, 0, 0);
        simpleextrapbc_(F(p1), BOXP(hi), &d1, &s1); or_simpleextrapbc(&op2, hi, 1, 1);
        cmp("simpleextrapbc_", p1, p2);
        simplecopybc_(F(p1), BOXP(lo), &d0, &s0); or_simplecopybc(&op2, lo, 0, 0);
        cmp("simplecopybc_", p1, p2);
        nullbc_(F(p1), BOXP(hi), &d1, &s1); or_nullbc(&op2, hi, 1, 1);
        cmp("nullbc_", p1, p2);
        Fab v1(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, -1.0, 1.0), v2 = v1; OrFab ov2 = v2.o(), ob0 = b0.o(), ob1 = b1.o();
        int i0 = 0, i1 = 1;
        divergence_(F(b0), F(v1), BOXP(reg), &dx[0], &i0); or_divergence(&ob0, &ov2, reg, dx[0], 0);
        divergence_(F(b1), F(v1), BOXP(reg), &dx[1], &i1); or_divergence(&ob1, &ov2, reg, dx[1], 1);
        cmp("divergence_", v1, v2);
    }
    {   // the time-step kernels (src/AmrHydroF.ChF) on the x-faces of the box
        OrBox fb{reg.lo0, reg.lo1, reg.hi0 + 1, reg.hi1};
        Fab Bec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 1e-4, 0.2), Rec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0.0, 4000.0), gH(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, -0.05, 0.05);
        Fab gZ(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, -0.02, 0.02), MRec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0.0, 1e-6), IMec(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, -1.0, 1.0);
        double omega = 1e-3, nu = 1.787e-6, rho = 910.0;
        Fab q1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), q2 = q1;
        OrFab oB = Bec.o(), oR = Rec.o(), oG = gH.o(), oZ = gZ.o(), oM = MRec.o(), oI = IMec.o(), oq2 = q2.o();
        computeqw_(F(Bec), F(Rec), F(gH), BOXP(fb), F(q1), &omega, &nu);
        or_computeqw(&oB, &oR, &oG, fb, &oq2, omega, nu);
        cmp("computeqw_", q1, q2);
        Fab p1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), p2 = p1, r1 = p1, r2 = p1; OrFab op2 = p2.o(), or2 = r2.o();
        computescaprod_(F(q1), F(gH), F(gZ), BOXP(fb), F(p1), F(r1));
        or_computescaprod(&oq2, &oG, &oZ, fb, &op2, &or2);
        cmp("computescaprod_ (Qw grad h)", p1, p2); cmp("computescaprod_ (Qw grad zb)", r1, r2);
        for (int cut = 0; cut < 2; cut++) {
            Fab d1(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 0, 0), d2 = d1; OrFab od2 = d2.o(); int c = cut;
            computedcoeff_(BOXP(fb), F(d1), dx, &rho, F(MRec), F(Bec), F(IMec), &c);
            or_computedcoeff(fb, &od2, rho, &oM, &oB, &oI, cut);
            cmp(cut ? "computedcoeff_ (cutOffB)" : "computedcoeff_", d1, d2);
        }
        OrBox fby{reg.lo0, reg.lo1, reg.hi0, reg.hi1 + 1};
        Fab D0(fb.lo0, fb.lo1, fb.hi0, fb.hi1, 1, 5e-6, 1e-3), D1(fby.lo0, fby.lo1, fby.hi0, fby.hi1, 1, 5e-6, 1e-3);
        Fab t1(reg.lo0, reg.lo1, reg.hi0, reg.hi1, 1, 0, 0), t2 = t1; OrFab ot2 = t2.o(), oD0 = D0.o(), oD1 = D1.o(), oph = phi.o();
        computedifterm2d_(F(phi), BOXP(reg), dx, F(t1), F(D0), F(D1));
        or_computedifterm2d(&oph, reg, dx, &ot2, &oD0, &oD1);
        cmp("computedifterm2d_", t1, t2);
        Fab zs(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, 0.0, 2000.0), w1(reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1, 1, 0, 0), w2 = w1;
        OrBox gb{reg.lo0 - 1, reg.lo1 - 1, reg.hi0 + 1, reg.hi1 + 1}; OrFab ozs = zs.o(), ow2 = w2.o();
        double TK = 9.5, bg = 7.93e-11;
        compute_timevaryingrecharge_(F(zs), BOXP(gb), F(w1), &TK, &bg);
        or_compute_timevaryingrecharge(&ozs, gb, &ow2, TK, bg);
        cmp("compute_timevaryingrecharge_", w1, w2);
    }
    printf(g_fail ? "RESULT: FAIL (%d)\n" : "RESULT: PASS\n", g_fail);
    return g_fail ? 1 : 0;
}
