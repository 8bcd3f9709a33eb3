/*
 * suhmo_chf.h -- boundary B2: the per-box Chombo-Fortran kernel ABI of the head solve.
 *
 * Chombo's FORT_X(...) macros call lower-case, underscore-suffixed symbols and expand their
 * argument macros (2-D) as below; evidence in the reference: the one vendored generated
 * header util/DivergenceF_F.H:17-27 and the expanded names used in util/ExtrapBCF.ChF:172-184.
 * libsuhmo_hip.so exports every kernel symbol on the hot path with exactly the argument order
 * of the reference's call sites, so a Chombo build can link it in place of the .ChF objects.
 * Pointers are HOST pointers (Chombo FArrayBox::dataPtr); each call stages the box through
 * HBM, runs the HIP kernel and copies the result back -- a compatibility path (one launch
 * and two PCIe trips per box); the performance path is the level-batched ABI in suhmo_hip.h.
 * (Exact macro text lives in Chombo's FORT_PROTO.H, which is not vendored: verify against a
 * real Chombo checkout before relying on link compatibility.)
 */
#ifndef SUHMO_CHF_H
#define SUHMO_CHF_H
#ifdef __cplusplus
extern "C" {
#endif

#define SUHMO_CHF_FRA(a)        double *a, const int *i##a##lo0, const int *i##a##lo1, const int *i##a##hi0, const int *i##a##hi1, const int *n##a##comp
#define SUHMO_CHF_CONST_FRA(a)  const double *a, const int *i##a##lo0, const int *i##a##lo1, const int *i##a##hi0, const int *i##a##hi1, const int *n##a##comp
#define SUHMO_CHF_FRA1(a)       double *a, const int *i##a##lo0, const int *i##a##lo1, const int *i##a##hi0, const int *i##a##hi1
#define SUHMO_CHF_BOX(b)        const int *i##b##lo0, const int *i##b##lo1, const int *i##b##hi0, const int *i##b##hi1
#define SUHMO_CHF_CONST_REAL(x)     const double *x
#define SUHMO_CHF_CONST_REALVECT(x) const double *x
#define SUHMO_CHF_CONST_INT(n)      const int *n
#define SUHMO_CHF_INT(n)            int *n

/* src/VCAMRNonLinearPoissonOpF.ChF:46-59; call site src/VCAMRNonLinearPoissonOp.cpp:722-744 */
void gsrbhelmholtzvcnl2d_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(rhs), SUHMO_CHF_BOX(region),
                          SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_CONST_REAL(alpha), SUHMO_CHF_CONST_FRA(aCoef),
                          SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoef0), SUHMO_CHF_CONST_FRA(bCoef1),
                          SUHMO_CHF_CONST_FRA(nlfunc), SUHMO_CHF_CONST_FRA(nlDfunc), SUHMO_CHF_CONST_FRA(lambda),
                          SUHMO_CHF_CONST_INT(redBlack));
/* ...OpF.ChF:201-211; call site .cpp:324-343 */
void vcnlcomputeop2d_(SUHMO_CHF_FRA(lofphi), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_REAL(alpha),
                      SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoef0),
                      SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc), SUHMO_CHF_BOX(region),
                      SUHMO_CHF_CONST_REALVECT(dx));
/* ...OpF.ChF:320-331; call site .cpp:145-165 */
void vcnlcomputeres2d_(SUHMO_CHF_FRA(res), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(rhs),
                       SUHMO_CHF_CONST_REAL(alpha), SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta),
                       SUHMO_CHF_CONST_FRA(bCoef0), SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc),
                       SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx));
/* ...OpF.ChF:419-423; call site .cpp:367-370 (and RESTRICTNL, AMRNonLinearPoissonOpF.ChF:491-495, .cpp:788-791) */
void restrictvcnl_(SUHMO_CHF_FRA(phiCoarse), SUHMO_CHF_CONST_FRA(phiFine), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REAL(dx));
void restrictnl_(SUHMO_CHF_FRA(phiCoarse), SUHMO_CHF_CONST_FRA(phiFine), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REAL(dx));
/* ...OpF.ChF:480-491; call site .cpp:438-458 */
void restrictresvcnl2d_(SUHMO_CHF_FRA(res), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(rhs),
                        SUHMO_CHF_CONST_REAL(alpha), SUHMO_CHF_CONST_FRA(aCoef), SUHMO_CHF_CONST_REAL(beta),
                        SUHMO_CHF_CONST_FRA(bCoef0), SUHMO_CHF_CONST_FRA(bCoef1), SUHMO_CHF_CONST_FRA(nlfunc),
                        SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx));
/* ...OpF.ChF:574-579; call site .cpp:522-527 */
void sumfacesnl_(SUHMO_CHF_FRA(lhs), SUHMO_CHF_CONST_REAL(beta), SUHMO_CHF_CONST_FRA(bCoefs), SUHMO_CHF_BOX(box),
                 SUHMO_CHF_CONST_INT(dir), SUHMO_CHF_CONST_REAL(scale));
/* src/AMRNonLinearPoissonOpF.ChF:607-611 / 646-650; call sites src/AMRNonLinearPoissonOp.cpp:880-883, 1194-1198 */
void prolongnl_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(coarse), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_INT(m));
void prolong_2_nl_(SUHMO_CHF_FRA(phi), SUHMO_CHF_CONST_FRA(coarse), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_INT(m));
/* ...OpF.ChF:711-716; call site .cpp:1817-1821 */
void newgetfluxnl_(SUHMO_CHF_FRA(flux), SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_BOX(box), SUHMO_CHF_CONST_REAL(beta_dx),
                   SUHMO_CHF_CONST_INT(a_idir));
/* src/AmrHydroF.ChF:23-34; call site src/AmrHydro.cpp:1560-1570 */
void computenonlinearterms_(SUHMO_CHF_CONST_FRA(phi), SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(IM),
                            SUHMO_CHF_CONST_FRA(aPi), SUHMO_CHF_CONST_FRA(aZb), SUHMO_CHF_BOX(region),
                            SUHMO_CHF_FRA(nlfunc), SUHMO_CHF_FRA(dnlfunc), SUHMO_CHF_CONST_REAL(Aparam),
                            SUHMO_CHF_CONST_REAL(brparam), SUHMO_CHF_CONST_REAL(brparamMax));
/* src/AmrHydroF.ChF:81-87; call site src/AmrHydro.cpp:1499-1504 */
void computere_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(agradH), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Re),
                SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam));
/* src/AmrHydroF.ChF:199-207; call site src/AmrHydro.cpp:1528-1535 */
void computebcoeff_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(aRe), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Bcoeff),
                    SUHMO_CHF_CONST_FRA(IMec), SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam),
                    SUHMO_CHF_INT(cutOffB));
/* util/GradientF.ChF:30-37; call site util/Gradient.cpp:791-798 */
void newmacgrad_(SUHMO_CHF_FRA1(edgeGrad), SUHMO_CHF_FRA1(mask), SUHMO_CHF_FRA1(phi), SUHMO_CHF_BOX(edgeGrid),
                 SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hasMask), SUHMO_CHF_INT(edgeDir));
/* util/ExtrapBCF.ChF:7-10, 39-42, 69-72; call sites util/ExtrapGhostCells.cpp:142, 230, 320 */
void simpleextrapbc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo));
void simplecopybc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo));
void nullbc_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(bcbox), SUHMO_CHF_INT(dir), SUHMO_CHF_INT(hiLo));
/* util/DivergenceF.ChF:23-27; prototype util/DivergenceF_F.H:17-27 */
void divergence_(SUHMO_CHF_CONST_FRA(uEdge), SUHMO_CHF_FRA(div), SUHMO_CHF_BOX(gridInt), SUHMO_CHF_CONST_REAL(dx),
                 SUHMO_CHF_INT(idir));
/* ---- the caller of the solve (AmrHydro::timeStepFAS): src/AmrHydroF.ChF
 * COMPUTEQW :113-150 (call site src/AmrHydro.cpp:1697-1705), COMPUTESCAPROD :162-186 (:2962-2967), COMPUTEDCOEFF :241-265
 * (:1851-1858), COMPUTEDIFTERM2D :289-343 (:2986-2991), COMPUTE_TIMEVARYINGRECHARGE :346-373 (:2853-2857) */
void computeqw_(SUHMO_CHF_CONST_FRA(aB), SUHMO_CHF_CONST_FRA(aRe), SUHMO_CHF_CONST_FRA(agradH), SUHMO_CHF_BOX(region),
                SUHMO_CHF_FRA(Qw), SUHMO_CHF_CONST_REAL(omegaparam), SUHMO_CHF_CONST_REAL(nuparam));
void computescaprod_(SUHMO_CHF_CONST_FRA(vara), SUHMO_CHF_CONST_FRA(var1b), SUHMO_CHF_CONST_FRA(var2b), SUHMO_CHF_BOX(region),
                     SUHMO_CHF_FRA(prod1), SUHMO_CHF_FRA(prod2));
void computedcoeff_(SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Dcoeff), SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_CONST_REAL(rho),
                    SUHMO_CHF_FRA(MRec), SUHMO_CHF_FRA(Bec), SUHMO_CHF_FRA(IMec), SUHMO_CHF_INT(cutOffB));
void computedifterm2d_(SUHMO_CHF_FRA(phi), SUHMO_CHF_BOX(region), SUHMO_CHF_CONST_REALVECT(dx), SUHMO_CHF_FRA(Dterm),
                       SUHMO_CHF_CONST_FRA(Dcoef0), SUHMO_CHF_CONST_FRA(Dcoef1));
void compute_timevaryingrecharge_(SUHMO_CHF_CONST_FRA(aZs), SUHMO_CHF_BOX(region), SUHMO_CHF_FRA(Recharge),
                                  SUHMO_CHF_CONST_REAL(TK), SUHMO_CHF_CONST_REAL(BackgroundInput));
/* the Fortran kernels call MAYDAYERROR() on a component mismatch (...OpF.ChF:87-106); the
 * replacement calls this hook (default: message + abort) */
void suhmo_chf_set_error_handler(void (*handler)(const char *msg));

#ifdef __cplusplus
}
#endif
#endif
