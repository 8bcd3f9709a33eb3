/*
 * suhmo_oracle.c -- TEST INFRASTRUCTURE ONLY.  See suhmo_oracle.h.
 * Parity status: suhmo_oracle.h (no kernel-level reference vectors exist; pinned end-to-end).
 *
 * Each function restates the 2D (CH_SPACEDIM == 2), ncomp-general arm of one
 * Chombo-Fortran subroutine of the reference, same loop nest order (component
 * outermost, j, then i fastest) and same left-to-right expression association.
 * Compile with -ffp-contract=off.
 */
#include "suhmo_oracle.h"
#include <math.h>
#include <stdlib.h>

#define AT(f, i, j, n) (*or_at((f), (i), (j), (n)))

/* src/VCAMRNonLinearPoissonOpF.ChF:108-165 */
void or_gsrbhelmholtzvcnl2d(OrFab *phi, const OrFab *rhs, OrBox region, const double dx[2],
                            double alpha, const OrFab *aCoef, double beta,
                            const OrFab *bCoef0, const OrFab *bCoef1,
                            const OrFab *nlfunc, const OrFab *nlDfunc,
                            const OrFab *lambda, int redBlack)
{
    double dxinv[2];
    for (int idir = 0; idir < 2; idir++) dxinv[idir] = 1.0 / (dx[idir] * dx[idir]); /* :110 */
    int ncomp = phi->ncomp;
    for (int n = 0; n < ncomp; n++) {
        for (int j = region.lo1; j <= region.hi1; j++) {
            int imin = region.lo0;
            int indtot = imin + j;                      /* :122 */
            imin = imin + abs((indtot + redBlack) % 2); /* :127 */
            int imax = region.hi0;
            for (int i = imin; i <= imax; i += 2) {
                double lofphi =
                    alpha * AT(aCoef, i, j, n) * AT(phi, i, j, n)
                    - beta *
                      (  AT(bCoef0, i + 1, j, n) * (AT(phi, i + 1, j, n) - AT(phi, i, j, n)) * dxinv[0]
                       - AT(bCoef0, i, j, n) * (AT(phi, i, j, n) - AT(phi, i - 1, j, n)) * dxinv[0]
                       + AT(bCoef1, i, j + 1, n) * (AT(phi, i, j + 1, n) - AT(phi, i, j, n)) * dxinv[1]
                       - AT(bCoef1, i, j, n) * (AT(phi, i, j, n) - AT(phi, i, j - 1, n)) * dxinv[1])
                    + AT(nlfunc, i, j, n);                              /* :130-152 */
                double denom = 1.0e-16 + AT(lambda, i, j, n) + AT(nlDfunc, i, j, n); /* :154 */
                AT(phi, i, j, n) = AT(phi, i, j, n) + (AT(rhs, i, j, n) - lofphi) / denom; /* :156 */
            }
        }
    }
}

/* src/VCAMRNonLinearPoissonOpF.ChF:252-281 */
void or_vcnlcomputeop2d(OrFab *lofphi, const OrFab *phi, double alpha, const OrFab *aCoef,
                        double beta, const OrFab *bCoef0, const OrFab *bCoef1,
                        const OrFab *nlfunc, OrBox region, const double dx[2])
{
    double dxinv[2];
    for (int idir = 0; idir < 2; idir++) dxinv[idir] = 1.0 / (dx[idir] * dx[idir]);
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) {
                AT(lofphi, i, j, n) =
                    alpha * AT(aCoef, i, j, n) * AT(phi, i, j, n)
                    - beta *
                      (  AT(bCoef0, i + 1, j, n) * (AT(phi, i + 1, j, n) - AT(phi, i, j, n)) * dxinv[0]
                       - AT(bCoef0, i, j, n) * (AT(phi, i, j, n) - AT(phi, i - 1, j, n)) * dxinv[0]
                       + AT(bCoef1, i, j + 1, n) * (AT(phi, i, j + 1, n) - AT(phi, i, j, n)) * dxinv[1]
                       - AT(bCoef1, i, j, n) * (AT(phi, i, j, n) - AT(phi, i, j - 1, n)) * dxinv[1])
                    + AT(nlfunc, i, j, n);
            }
}

/* src/VCAMRNonLinearPoissonOpF.ChF:373-403 */
void or_vcnlcomputeres2d(OrFab *res, const OrFab *phi, const OrFab *rhs, double alpha,
                         const OrFab *aCoef, double beta, const OrFab *bCoef0,
                         const OrFab *bCoef1, const OrFab *nlfunc, OrBox region,
                         const double dx[2])
{
    double dxinv[2];
    for (int idir = 0; idir < 2; idir++) dxinv[idir] = 1.0 / (dx[idir] * dx[idir]);
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) {
                AT(res, i, j, n) =
                    AT(rhs, i, j, n)
                    - (alpha * AT(aCoef, i, j, n) * AT(phi, i, j, n)
                       - beta *
                         (  AT(bCoef0, i + 1, j, n) * (AT(phi, i + 1, j, n) - AT(phi, i, j, n)) * dxinv[0]
                          - AT(bCoef0, i, j, n) * (AT(phi, i, j, n) - AT(phi, i - 1, j, n)) * dxinv[0]
                          + AT(bCoef1, i, j + 1, n) * (AT(phi, i, j + 1, n) - AT(phi, i, j, n)) * dxinv[1]
                          - AT(bCoef1, i, j, n) * (AT(phi, i, j, n) - AT(phi, i, j - 1, n)) * dxinv[1])
                       + AT(nlfunc, i, j, n));
            }
}

/* src/VCAMRNonLinearPoissonOpF.ChF:432-446; identical body in
 * src/AMRNonLinearPoissonOpF.ChF:505-519 (RESTRICTNL) */
void or_restrictvcnl(OrFab *phiCoarse, const OrFab *phiFine, OrBox region)
{
    double denom = 2 * 2; /* D_TERM(2, *2, *2) */
    for (int n = 0; n < phiFine->ncomp; n++)
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) {
                int ii = i / 2, jj = j / 2;
                AT(phiCoarse, ii, jj, n) = AT(phiCoarse, ii, jj, n) + AT(phiFine, i, j, n) / denom;
            }
}

/* src/VCAMRNonLinearPoissonOpF.ChF:516-558 */
void or_restrictresvcnl2d(OrFab *res, const OrFab *phi, const OrFab *rhs, double alpha,
                          const OrFab *aCoef, double beta, const OrFab *bCoef0,
                          const OrFab *bCoef1, const OrFab *nlfunc, OrBox region,
                          const double dx[2])
{
    double dxinv[2];
    for (int idir = 0; idir < 2; idir++) dxinv[idir] = 1.0 / (dx[idir] * dx[idir]);
    double denom = 2 * 2;
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) {
                int ii = i / 2, jj = j / 2;
                double lofphi =
                    alpha * AT(aCoef, i, j, n) * AT(phi, i, j, n)
                    - beta *
                      (  AT(bCoef0, i + 1, j, n) * (AT(phi, i + 1, j, n) - AT(phi, i, j, n)) * dxinv[0]
                       - AT(bCoef0, i, j, n) * (AT(phi, i, j, n) - AT(phi, i - 1, j, n)) * dxinv[0]
                       + AT(bCoef1, i, j + 1, n) * (AT(phi, i, j + 1, n) - AT(phi, i, j, n)) * dxinv[1]
                       - AT(bCoef1, i, j, n) * (AT(phi, i, j, n) - AT(phi, i, j - 1, n)) * dxinv[1])
                    + AT(nlfunc, i, j, n);
                AT(res, ii, jj, n) = AT(res, ii, jj, n) + (AT(rhs, i, j, n) - lofphi) / denom;
            }
}

/* src/VCAMRNonLinearPoissonOpF.ChF:586-598 */
void or_sumfacesnl(OrFab *lhs, double beta, const OrFab *bCoefs, OrBox box, int dir,
                   double scale)
{
    int ii = (dir == 0), jj = (dir == 1);
    for (int n = 0; n < lhs->ncomp; n++)
        for (int j = box.lo1; j <= box.hi1; j++)
            for (int i = box.lo0; i <= box.hi0; i++) {
                double sumVal = AT(bCoefs, i + ii, j + jj, n) + AT(bCoefs, i, j, n);
                AT(lhs, i, j, n) = AT(lhs, i, j, n) + scale * beta * sumVal;
            }
}

/* src/AMRNonLinearPoissonOpF.ChF:617-628 */
void or_prolongnl(OrFab *phi, const OrFab *coarse, OrBox region, int m)
{
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = region.lo1; j <= region.hi1; j++)
            for (int i = region.lo0; i <= region.hi0; i++) {
                int ii = i / m, jj = j / m;
                AT(phi, i, j, n) = AT(phi, i, j, n) + AT(coarse, ii, jj, n);
            }
}

/* src/AMRNonLinearPoissonOpF.ChF:660-705 (2D arm) */
void or_prolong_2_nl(OrFab *phi, const OrFab *coarse, OrBox region, int m)
{
    double den = 1.0 / 16.0; /* one/(4**CH_SPACEDIM) :660 */
    double fx1 = 3.0 * den;  /* :662 */
    double fx2 = 9.0 * den;  /* three**2*den :663 */
    double f0 = 1.0 * den;   /* :665 */
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            int ic = i / m, jc = j / m;
            int o1 = 2 * (i % 2) - 1; /* :677 */
            int o2 = 2 * (j % 2) - 1;
            for (int n = 0; n < phi->ncomp; n++) {
                AT(phi, i, j, n) = AT(phi, i, j, n)
                    + fx2 * AT(coarse, ic, jc, n)
                    + f0 * AT(coarse, ic + o1, jc + o2, n);       /* :683-686 */
                AT(phi, i, j, n) = AT(phi, i, j, n)
                    + fx1 * (AT(coarse, ic + o1, jc, n) + AT(coarse, ic, jc + o2, n)); /* :689-694 */
            }
        }
}

/* src/AMRNonLinearPoissonOpF.ChF:722-737 */
void or_newgetfluxnl(OrFab *flux, const OrFab *phi, OrBox box, double beta_dx, int idir)
{
    int ii = (idir == 0), jj = (idir == 1);
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = box.lo1; j <= box.hi1; j++)
            for (int i = box.lo0; i <= box.hi0; i++)
                AT(flux, i, j, n) = -(AT(phi, i, j, n) - AT(phi, i - ii, j - jj, n)) * beta_dx;
}

/* src/VCAMRNonLinearPoissonOp.cpp:820-840 */
void or_vc_getflux(OrFab *flux, const OrFab *phi, const OrFab *bCoefDir, OrBox facebox,
                   int dir, double beta, double dx_dir, int ref)
{
    int ii = (dir == 0), jj = (dir == 1);
    double scale = beta * ref / dx_dir; /* :820 */
    for (int j = facebox.lo1; j <= facebox.hi1; j++)
        for (int i = facebox.lo0; i <= facebox.hi0; i++)
            for (int n = 0; n < phi->ncomp; n++) {
                double phihi = AT(phi, i, j, n);
                double philo = AT(phi, i - ii, j - jj, n);
                double gradphi = (phihi - philo) * scale;
                AT(flux, i, j, n) = -AT(bCoefDir, i, j, n) * gradphi;
            }
}

/* src/AmrHydroF.ChF:38-65.  The effective-pressure factor is written out three
 * times in the reference; it is a pure function of the cell so one evaluation is
 * bit-identical. */
void or_computenonlinearterms(const OrFab *phi, const OrFab *aB, const OrFab *IM,
                              const OrFab *aPi, const OrFab *aZb, OrBox region,
                              OrFab *nlfunc, OrFab *dnlfunc, const OrPhys *ph)
{
    const double Aparam = ph->A, brparam = ph->cutOffbr, brparamMax = ph->maxOffbr;
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            if (AT(IM, i, j, 0) < 0.0) { /* :40 */
                AT(nlfunc, i, j, 0) = 0.0;
                AT(dnlfunc, i, j, 0) = 0.0;
            } else {
                double B = AT(aB, i, j, 0);
                double N = AT(aPi, i, j, 0) - ph->rho_w_g * (AT(phi, i, j, 0) - AT(aZb, i, j, 0));
                AT(nlfunc, i, j, 0) = -Aparam * B * N * N * N;                      /* :44-47 */
                AT(dnlfunc, i, j, 0) = 3.0 * Aparam * B * 1000.0 * ph->grav * N * N; /* :49-52 */
                if (brparam > B) {                                                  /* :54 */
                    AT(nlfunc, i, j, 0) = AT(nlfunc, i, j, 0) * (1.0 - (brparam - B) / brparam);
                    AT(dnlfunc, i, j, 0) = AT(dnlfunc, i, j, 0) * B / brparam;
                }
                if (brparamMax < B) {                                               /* :59 */
                    AT(nlfunc, i, j, 0) = AT(nlfunc, i, j, 0) * (1.0 - (brparamMax - B) / brparamMax);
                    AT(dnlfunc, i, j, 0) = AT(dnlfunc, i, j, 0) * B / brparamMax;
                }
            }
        }
}

/* src/AmrHydroF.ChF:92-109 */
void or_computere(const OrFab *aB, const OrFab *agradH, OrBox region, OrFab *Re,
                  const OrPhys *ph)
{
    const double omegaparam = ph->omega, nuparam = ph->nu;
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            double sqrt_gradH_cc = sqrt(AT(agradH, i, j, 0) * AT(agradH, i, j, 0)
                                        + AT(agradH, i, j, 1) * AT(agradH, i, j, 1));
            double discr = 1.0 + 4.0 * omegaparam *
                               (AT(aB, i, j, 0) * AT(aB, i, j, 0) * AT(aB, i, j, 0)
                                * ph->grav * sqrt_gradH_cc) /
                               (12.0 * nuparam * nuparam);
            AT(Re, i, j, 0) = (-1.0 + sqrt(discr)) / (2.0 * omegaparam);
        }
}

/* src/AmrHydroF.ChF:212-228 */
void or_computebcoeff(const OrFab *aB, const OrFab *aRe, OrBox region, OrFab *Bcoeff,
                      const OrFab *IMec, const OrPhys *ph)
{
    const double omegaparam = ph->omega, nuparam = ph->nu;
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            double num_q = -(AT(aB, i, j, 0) * AT(aB, i, j, 0) * AT(aB, i, j, 0) * ph->grav);
            double denom_q = 12.0 * nuparam * (1.0 + omegaparam * AT(aRe, i, j, 0));
            if ((AT(IMec, i, j, 0) < 0.0) && (ph->cutOffB > 0))
                AT(Bcoeff, i, j, 0) = 0.0;
            else
                AT(Bcoeff, i, j, 0) = num_q / denom_q;
        }
}

/* src/AmrHydroF.ChF:313-340 */
void or_computedifterm2d(const OrFab *phi, OrBox region, const double dx[2], OrFab *Dterm,
                         const OrFab *Dcoef0, const OrFab *Dcoef1)
{
    double dxinv[2];
    for (int idir = 0; idir < 2; idir++) dxinv[idir] = 1.0 / (dx[idir] * dx[idir]);
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            AT(Dterm, i, j, 0) =
                (  AT(Dcoef0, i + 1, j, 0) * (AT(phi, i + 1, j, 0) - AT(phi, i, j, 0)) * dxinv[0]
                 - AT(Dcoef0, i, j, 0) * (AT(phi, i, j, 0) - AT(phi, i - 1, j, 0)) * dxinv[0]
                 + AT(Dcoef1, i, j + 1, 0) * (AT(phi, i, j + 1, 0) - AT(phi, i, j, 0)) * dxinv[1]
                 - AT(Dcoef1, i, j, 0) * (AT(phi, i, j, 0) - AT(phi, i, j - 1, 0)) * dxinv[1]);
        }
}

/* src/AmrHydroF.ChF:113-150 */
void or_computeqw(const OrFab *aB, const OrFab *aRe, const OrFab *agradH, OrBox region, OrFab *Qw, double omega, double nu)
{
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            double num_q = -(AT(aB, i, j, 0) * AT(aB, i, j, 0) * AT(aB, i, j, 0) * 9.8 * AT(agradH, i, j, 0));
            double denom_q = 12.0 * nu * (1.0 + omega * AT(aRe, i, j, 0));
            AT(Qw, i, j, 0) = num_q / denom_q;
        }
}
/* src/AmrHydroF.ChF:162-186 */
void or_computescaprod(const OrFab *vara, const OrFab *var1b, const OrFab *var2b, OrBox region, OrFab *prod1, OrFab *prod2)
{
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            AT(prod1, i, j, 0) = AT(vara, i, j, 0) * AT(var1b, i, j, 0);
            AT(prod2, i, j, 0) = AT(vara, i, j, 0) * AT(var2b, i, j, 0);
        }
}
/* src/AmrHydroF.ChF:241-265 */
void or_computedcoeff(OrBox region, OrFab *Dcoeff, double rho, const OrFab *MRec, const OrFab *Bec, const OrFab *IMec, int cutOffB)
{
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++) {
            if (AT(IMec, i, j, 0) < 0.0 && cutOffB > 0) AT(Dcoeff, i, j, 0) = 0.0;
            else AT(Dcoeff, i, j, 0) = fmax(AT(Bec, i, j, 0) * AT(MRec, i, j, 0) / rho, 5.0e-6);
        }
}
/* src/AmrHydroF.ChF:346-373 */
void or_compute_timevaryingrecharge(const OrFab *aZs, OrBox region, OrFab *Recharge, double TK, double BackgroundInput)
{
    const double ddf = 0.01 / 86400., dT_dZ = -0.0075;
    for (int j = region.lo1; j <= region.hi1; j++)
        for (int i = region.lo0; i <= region.hi0; i++)
            AT(Recharge, i, j, 0) = fmax(ddf * (TK + AT(aZs, i, j, 0) * dT_dZ), 0.0) + BackgroundInput;
}

/* util/GradientF.ChF:55-70 (normal derivative; CHF_FRA1 = single component) */
void or_newmacgrad(OrFab *edgeGrad, const OrFab *mask, const OrFab *phi, OrBox edgeGrid,
                   const double dx[2], int dir, int hasMask)
{
    int ii = (dir == 0), jj = (dir == 1);
    double factor = 1.0 / dx[dir];
    for (int j = edgeGrid.lo1; j <= edgeGrid.hi1; j++)
        for (int i = edgeGrid.lo0; i <= edgeGrid.hi0; i++) {
            if (hasMask > 0) {
                if ((AT(mask, i, j, 0) < 1e-6) || (AT(mask, i - ii, j - jj, 0) < 1e-6))
                    AT(edgeGrad, i, j, 0) = 0.0;
                else
                    AT(edgeGrad, i, j, 0) = factor * (AT(phi, i, j, 0) - AT(phi, i - ii, j - jj, 0));
            } else {
                AT(edgeGrad, i, j, 0) = factor * (AT(phi, i, j, 0) - AT(phi, i - ii, j - jj, 0));
            }
        }
}

/* util/ExtrapBCF.ChF:17-29 */
void or_simpleextrapbc(OrFab *phi, OrBox bcbox, int dir, int hiLo)
{
    int offset = 1;
    if (hiLo == 0) offset = -1;
    int ii0 = offset * (dir == 0), ii1 = offset * (dir == 1);
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = bcbox.lo1; j <= bcbox.hi1; j++)
            for (int i = bcbox.lo0; i <= bcbox.hi0; i++)
                AT(phi, i, j, n) = 2.0 * AT(phi, i - ii0, j - ii1, n) - AT(phi, i - 2 * ii0, j - 2 * ii1, n);
}

/* util/ExtrapBCF.ChF:49-60 */
void or_simplecopybc(OrFab *phi, OrBox bcbox, int dir, int hiLo)
{
    int offset = 1;
    if (hiLo == 0) offset = -1;
    int ii0 = offset * (dir == 0), ii1 = offset * (dir == 1);
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = bcbox.lo1; j <= bcbox.hi1; j++)
            for (int i = bcbox.lo0; i <= bcbox.hi0; i++)
                AT(phi, i, j, n) = AT(phi, i - ii0, j - ii1, n);
}

/* util/ExtrapBCF.ChF:79-90 */
void or_nullbc(OrFab *phi, OrBox bcbox, int dir, int hiLo)
{
    (void)dir; (void)hiLo;
    for (int n = 0; n < phi->ncomp; n++)
        for (int j = bcbox.lo1; j <= bcbox.hi1; j++)
            for (int i = bcbox.lo0; i <= bcbox.hi0; i++)
                AT(phi, i, j, n) = 0.0;
}

/* util/DivergenceF.ChF:38-54 */
void or_divergence(const OrFab *uEdge, OrFab *div, OrBox gridInt, double dx, int idir)
{
    int h0 = (idir == 0), h1 = (idir == 1); /* c2fHi; c2fLo = 0 */
    double one_on_dx = 1.0 / dx;
    for (int comp = 0; comp < div->ncomp; comp++)
        for (int j = gridInt.lo1; j <= gridInt.hi1; j++)
            for (int i = gridInt.lo0; i <= gridInt.hi0; i++)
                AT(div, i, j, comp) = AT(div, i, j, comp)
                    + one_on_dx * (AT(uEdge, i + h0, j + h1, comp) - AT(uEdge, i, j, comp));
}
