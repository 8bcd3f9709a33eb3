// suhmo_gsrb.hip -- nonlinear variable-coefficient Gauss-Seidel red-black relaxation.
//
// Replaces VCAMRNonLinearPoissonOp::levelGSRB (src/VCAMRNonLinearPoissonOp.cpp:654-760):
//   per colour pass: exchange -> mixBCValues -> NonLinear_level -> GSRBHELMHOLTZVCNL2D
// The physical BC is evaluated on the fly, the nonlinear term and its derivative
// (COMPUTENONLINEARTERMS, src/AmrHydroF.ChF:23-68) and the relaxation coefficient lambda
// (SUMFACESNL, ...OpF.ChF:574-601) are recomputed in registers, so per sweep only phi, rhs,
// bx, by, B, Pi, zb, mask stream through HBM (72 B per cell per sweep).
// Colour rule: cell (i,j) is updated in pass p iff (i + j_global + p) is even
// (src/VCAMRNonLinearPoissonOpF.ChF:121-129).  GSRB is colour-Jacobi, so the result does not
// depend on the box decomposition: the kernels work on the whole level canvas.
//
// Two kernels:
//   k_gsrb_pass_simple   one launch per colour pass, one thread per active cell (small /
//                        odd-sized depths; also the structural reference for the fused one)
//   k_gsrb_fused<K>      K full sweeps (2K colour passes) in ONE pass over HBM: each
//                        workgroup owns a column strip x row chunk and marches down the rows
//                        with a (2K+3)-row ring of phi in LDS; row r-m is advanced from
//                        half-sweep m-1 to m while row r+1 is being prefetched, and row r-2K
//                        (all 2K half-sweeps done) is streamed out.  Halos of 2K cells are
//                        recomputed redundantly, output goes to a second phi buffer (ping-
//                        pong) so no workgroup ever reads a neighbour's updated cell.
#include "suhmo_hier.h"
#include <type_traits>

// ---- variant 0: one thread per active-colour cell ----
// PUSH (boxes of a multi-box level): a cell on the side of its box also stores its new value into the ghost cell of the box
// across that side (push[pbase + side cell] = {box, canvas offset}, box < 0: none), so the next colour pass needs no
// Copier::exchange launch in between: a ghost cell of colour c is read only by the pass of the other colour.
template <bool HAS_ALPHA>
__device__ __forceinline__ void d_gsrb_pass_simple(const DV &v, const FP &fp, suhmo_phys_t ph, int pass, int jlo, int jhi,
                                                   const FP *__restrict__ ft = nullptr, const int2 *__restrict__ push = nullptr)
{
    // rows [jlo, jhi]: the strip's own rows plus, on rank boundaries, the halo rows that are
    // still fresh enough to be advanced redundantly (one exchange then feeds several passes)
    int j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > jhi) return;
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
    const double pnew = c + (fp.f[SUHMO_F_RHS][idx] - lofphi) / denom; // :156
    phi[idx] = pnew;
    if (push) {                                               // sides in the order W, E (ny cells each), S, N (nx cells each)
        if (i == 0) { int2 q = push[j]; if (q.x >= 0) ft[q.x].f[SUHMO_F_PHI][q.y] = pnew; }
        if (i == v.nx - 1) { int2 q = push[v.ny + j]; if (q.x >= 0) ft[q.x].f[SUHMO_F_PHI][q.y] = pnew; }
        if (j == 0) { int2 q = push[2 * v.ny + i]; if (q.x >= 0) ft[q.x].f[SUHMO_F_PHI][q.y] = pnew; }
        if (j == v.ny - 1) { int2 q = push[2 * v.ny + v.nx + i]; if (q.x >= 0) ft[q.x].f[SUHMO_F_PHI][q.y] = pnew; }
    }
}
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_gsrb_pass_simple(DV v, FP fp, suhmo_phys_t ph, int pass, int jlo, int jhi)
{
    d_gsrb_pass_simple<HAS_ALPHA>(v, fp, ph, pass, jlo, jhi);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_gsrb_pass_simple_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph, int pass,
                                                            const int2 *__restrict__ push, const int *__restrict__ pbase)
{
    d_gsrb_pass_simple<HAS_ALPHA>(vt[blockIdx.z], ft[blockIdx.z], ph, pass, 0, vt[blockIdx.z].ny - 1, ft, push ? push + pbase[blockIdx.z] : nullptr);
}

static void launch_simple(suhmo_level *L, int depth, int pass, int ext_rows, hipStream_t st)
{
    Depth &D = L->d[depth];
    int jlo = D.v.rk[0] ? -ext_rows : 0, jhi = D.v.ny - 1 + (D.v.rk[1] ? ext_rows : 0);
    dim3 blk(64, 4), grd(((D.v.nx + 1) / 2 + 63) / 64, (jhi - jlo + 1 + 3) / 4);
    if (D.v.alpha != 0.0)
        hipLaunchKernelGGL(k_gsrb_pass_simple<true>, grd, blk, 0, st, D.v, D.fp, L->ph, pass, jlo, jhi);
    else
        hipLaunchKernelGGL(k_gsrb_pass_simple<false>, grd, blk, 0, st, D.v, D.fp, L->ph, pass, jlo, jhi);
}

// one colour pass of one box of a multi-box AMR level (suhmo_hier.hip): in place, the ghost ring holds exchanged /
// interpolated data
int suhmo_gsrb_colour_pass(suhmo_level *L, int depth, int pass, hipStream_t st)
{
    launch_simple(L, depth, pass, 0, st);
    HIPCHK(hipGetLastError());
    L->d[depth].phi_fresh = 0;
    return 0;
}

int suhmo_multi_colour_pass(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, int pass, hipStream_t st, bool push)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    dim3 blk(64, 4), grd(((m.maxnx + 1) / 2 + 63) / 64, (m.maxny + 3) / 4, m.nbox);
    const int2 *pp = push ? (const int2 *)m.push : nullptr;
    if (has_alpha) hipLaunchKernelGGL(k_gsrb_pass_simple_m<true>, grd, blk, 0, st, m.dv, m.fp, ph, pass, pp, m.pbase);
    else hipLaunchKernelGGL(k_gsrb_pass_simple_m<false>, grd, blk, 0, st, m.dv, m.fp, ph, pass, pp, m.pbase);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- levels that are UNIONS OF BOXES: several sweeps per launch (relaxNF, src/AMRNonLinearPoissonOp.cpp:690-750; levelGSRB :654-760).
// A workgroup owns a tile of a box and loads it with a halo of G = 4 or 8 cells -- the cells of the neighbouring boxes, taken from THEIR canvases (plan
// `halo`: for every position of the extended box the box that holds the cell and its canvas offset, or none) -- into LDS, advances the halo
// cells redundantly (pass m keeps what lies within 3 - m cells of the box current) and after 2 sweeps = 4 colour passes writes the box to the
// second canvas of the head.  What a cell reads beyond the cells of the level is what its OWN box's ghost ring says: the stored coarse-fine
// ghost (interpolated before the relaxation, constant during it) or the physical boundary condition of its side, evaluated with that box's
// view -- at a re-entrant corner of the union a position is the x-ghost of one box and the y-ghost of another, and each reader gets its own.
// Every update is d_gsrb_pass_simple's expression on the same operands in the same order, so the result is the colour passes' with an
// exchange before each, bit for bit; a launch reads canvases no workgroup of the launch writes (PHI -> PHI2, then PHI2 -> PHI).
// G: the halo the launch advances through = the sweeps it can do (G = 4: two sweeps; G = 8: FOUR sweeps -- a whole pre- or post-smoothing of the
// reference's num_smooth = 4 -- in one launch: the eight passes cost less than two launches' fixed parts).  T: the tile edge at most (a box is cut
// into ceil(n / T) tiles of equal width per direction).  Threads keep S positions of EACH colour of the tile's image.  What runs is T = 16, S = 1
// (320 / 512 threads): many small workgroups.  Measured and not kept: T = 32, S = 2 (48 x 48 image on 576 threads, 168 VGPRs) -- a 28 x 28 box as
// four tiles with their halos is 3600 positions and as one tile 1936, but cfg5's ~200 boxes per level do not fill 256 CUs with one workgroup each
// and a wave with two updates per pass has twice the dependent chain: 20.5 against 16.8 ms per step; on one box of 1024 x 256 cells 3.22 against
// 3.32 ms per step, within the box-to-box noise (`profiles/r04_w_box_tile_ab.txt`).  The plan `halo` is laid out for SUHMO_BOX_HALO = 8 cells around every box.
template <int G, int T, int S> struct BoxGeom { static constexpr int LW = T + 2 * G, NT = (((LW * LW / 2 + S - 1) / S + 63) / 64) * 64; };
template <bool HAS_ALPHA, int G, int T, int S>
__global__ __launch_bounds__((BoxGeom<G, T, S>::NT)) __attribute__((amdgpu_waves_per_eu(1, 3))) void k_gsrb_box_m(const DV *__restrict__ vt,
    const FP *__restrict__ ft, const int2 *__restrict__ halo, const int *__restrict__ hbase,
                                                                     suhmo_phys_t ph, int fsrc, int fdst, int npass, int bcg)
{
    constexpr int BOXG = G, BOXNT = BoxGeom<G, T, S>::NT, HG = SUHMO_BOX_HALO;
    constexpr int LWmax = T + 2 * BOXG;
    __shared__ double pl[LWmax * LWmax];
    __shared__ int own[LWmax * LWmax];
    const int k = blockIdx.y, tid = threadIdx.x;
    const DV &v = vt[k];                                               // (uniform: scalar loads, not a copy per lane)
    const int tiles_x = (v.nx + T - 1) / T, tiles_y = (v.ny + T - 1) / T;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int tj = blockIdx.x / tiles_x, ti = blockIdx.x - tj * tiles_x;
    const int twb = (v.nx + tiles_x - 1) / tiles_x, thb = (v.ny + tiles_y - 1) / tiles_y;      // tiles of equal size (the last one may be smaller)
    const int x0 = ti * twb, y0 = tj * thb;                            // the tile's first cell in the box = its halo's first position in the extended box
    const int tw = min(twb, v.nx - x0), th = min(thb, v.ny - y0);
    if (tw <= 0 || th <= 0) return;
    const int LW = tw + 2 * BOXG, LH = th + 2 * BOXG, EW = v.nx + 2 * HG, HW = (LW + 1) / 2;
    const int2 *__restrict__ hk = halo + hbase[k];
    const FP &fk = ft[k];
    // A thread keeps S positions of each colour of the tile's image through all passes (slot [s][c]: (i + j) & 1 == c, global indices; the colour
    // of a position follows from where the image lies, a periodic image keeps its parity), with everything about it that does not change: the
    // cell's box and offset, its coefficients, how far from the tile it lies.  A pass runs the slots of its colour: no lane idles for its colour.
    // One load phase of three dependent steps: the plan entry -> the box's view and field pointers (its own box: none) -> head and coefficients.
    const int par0 = (v.i0 + x0 - BOXG + v.j0 + y0 - BOXG) & 1;          // colour of the image's position (0, 0)
    int cq[S][2], cb[S][2], co[S][2], cd[S][2];    // position in the image (-1: none), box (-1: not advanced), canvas offset, distance from the tile
    double c_rhs[S][2], c_bxW[S][2], c_bxE[S][2], c_byS[S][2], c_byN[S][2], c_B[S][2], c_Pi[S][2], c_zb[S][2], c_mk[S][2], c_at[S][2];
    int2 hq[S][2];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            cq[s][c] = -1; hq[s][c] = int2{-1, 0};
            const int p = tid + s * BOXNT;
            if (p < LH * HW) {
                const int lj = p / HW, li = 2 * (p - lj * HW) + ((lj + par0 + c) & 1);
                if (li < LW) { cq[s][c] = lj * LW + li; hq[s][c] = hk[(y0 + lj + HG - BOXG) * EW + x0 + li + HG - BOXG]; }
            }
        }
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int q = cq[s][c];
            cb[s][c] = -1; co[s][c] = 0; cd[s][c] = 0;
            c_rhs[s][c] = c_bxW[s][c] = c_bxE[s][c] = c_byS[s][c] = c_byN[s][c] = c_B[s][c] = c_Pi[s][c] = c_zb[s][c] = c_mk[s][c] = c_at[s][c] = 0.0;
            if (q < 0) continue;
            const int b = hq[s][c].x, off = hq[s][c].y;
            own[q] = b;
            if (b < 0) { pl[q] = 0.0; continue; }
            const int lj = q / LW, li = q - lj * LW;
            const int dx = li < BOXG ? BOXG - li : (li >= BOXG + tw ? li - (BOXG + tw - 1) : 0), dy = lj < BOXG ? BOXG - lj : (lj >= BOXG + th ? lj - (BOXG + th - 1) : 0);
            const int d = max(dx, dy);
            const FP &fb = b == k ? fk : ft[b];
            if (d >= BOXG) { pl[q] = fb.f[fsrc][off]; continue; }        // (the outermost ring is only read)
            const int Pb = b == k ? v.P : vt[b].P;
            cb[s][c] = b; co[s][c] = off; cd[s][c] = d;
            pl[q] = fb.f[fsrc][off];
            c_rhs[s][c] = fb.f[SUHMO_F_RHS][off];
            c_bxW[s][c] = fb.f[SUHMO_F_BX][off]; c_bxE[s][c] = fb.f[SUHMO_F_BX][off + 1];
            c_byS[s][c] = fb.f[SUHMO_F_BY][off]; c_byN[s][c] = fb.f[SUHMO_F_BY][off + Pb];
            c_B[s][c] = fb.f[SUHMO_F_B][off]; c_Pi[s][c] = fb.f[SUHMO_F_PI][off]; c_zb[s][c] = fb.f[SUHMO_F_ZB][off]; c_mk[s][c] = fb.f[SUHMO_F_MASK][off];
            c_at[s][c] = HAS_ALPHA ? v.alpha * fb.f[SUHMO_F_ACOEF][off] : v.alpha;      // (alpha, beta, the cell sizes and the BC data are the level's: every box's view has them)
        }
    __syncthreads();
    // What a cell reads beyond the cells of the level does not change during the launch: its own box's stored ghost (coarse-fine: interpolated before
    // the relaxation) or a physical boundary condition of its own value.  Worked out ONCE per position -- the box's view, the cell's indices, the ghost
    // value: global loads and an index computation that used to sit in every pass of every wave with a cell next to the level's edge -- as a kind per
    // direction (2 bits each: 0 a cell of the level, read from LDS; 1 the constant g; 2 g - c, Dirichlet: g = 2 x value; 3 c + g, Neumann: g = +-dx x
    // value: phiW / phiE / phiS / phiN's expressions) and the four g.
    int nk[S][2];
    double hW[S][2], hE[S][2], hS[S][2], hN[S][2];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            nk[s][c] = 0; hW[s][c] = hE[s][c] = hS[s][c] = hN[s][c] = 0.0;
            const int q = cq[s][c], b = cb[s][c];
            if (b < 0) continue;
            const bool oW = own[q - 1] < 0, oE = own[q + 1] < 0, oS = own[q - LW] < 0, oN = own[q + LW] < 0;
            if (!(oW || oE || oS || oN)) continue;
            const DV &vb = b == k ? v : vt[b];
            const double *__restrict__ psrc = (b == k ? fk : ft[b]).f[fsrc];
            const int off = co[s][c];
            // (a neighbour that is not a cell of the level lies outside the cell's own box: i == 0 / nx - 1 / j == 0 / ny - 1 there; a box that is its own
            //  periodic neighbour has its images in the plan as cells of the level)
            if (oW) { if (vb.cfx[0]) { nk[s][c] |= 1; hW[s][c] = psrc[off - 1]; } else if (vb.bct[0][0] == 0) { nk[s][c] |= 2; hW[s][c] = vb.two_v[0][0];
                } else { nk[s][c] |= 3; hW[s][c] = vb.neu[0][0]; } }
            if (oE) { if (vb.cfx[1]) { nk[s][c] |= 1 << 2; hE[s][c] = psrc[off + 1]; } else if (vb.bct[0][1] == 0) { nk[s][c] |= 2 << 2; hE[s][c] = vb.two_v[0][1];
                } else { nk[s][c] |= 3 << 2; hE[s][c] = vb.neu[0][1]; } }
            if (oS) { if (vb.ext[0]) { nk[s][c] |= 1 << 4; hS[s][c] = psrc[off - vb.P]; } else if (vb.bct[1][0] == 0) { nk[s][c] |= 2 << 4;
                hS[s][c] = vb.two_v[1][0]; } else { nk[s][c] |= 3 << 4; hS[s][c] = vb.neu[1][0]; } }
            if (oN) { if (vb.ext[1]) { nk[s][c] |= 1 << 6; hN[s][c] = psrc[off + vb.P]; } else if (vb.bct[1][1] == 0) { nk[s][c] |= 2 << 6;
                hN[s][c] = vb.two_v[1][1]; } else { nk[s][c] |= 3 << 6; hN[s][c] = vb.neu[1][1]; } }
        }
    auto outside = [](int kind, double g, double c) { return kind == 1 ? g : (kind == 2 ? g - c : c + g); };
    // one update: d_gsrb_pass_simple's expression.  Called with CONSTANT slot indices from either branch of the (uniform) colour test below: a
    // `u ? x[1] : x[0]` on the slot arrays is turned into a variably indexed load by the optimiser, which keeps all of them in scratch
    auto update = [&](int q, int kinds, double g_w, double g_e, double g_s, double g_n, double rhs, double bxW, double bxE, double byS, double byN, double B,
        double Pi, double zb, double mk, double at) {
        const double c = pl[q];
        // a neighbour that is a cell of the level: its current value; else what the cell's own box holds there (coarse-fine ghost / physical BC)
        double w, e, sv, n;
        if (kinds == 0) { w = pl[q - 1]; e = pl[q + 1]; sv = pl[q - LW]; n = pl[q + LW]; }
        else {
            w = (kinds & 3) ? outside(kinds & 3, g_w, c) : pl[q - 1];
            e = ((kinds >> 2) & 3) ? outside((kinds >> 2) & 3, g_e, c) : pl[q + 1];
            sv = ((kinds >> 4) & 3) ? outside((kinds >> 4) & 3, g_s, c) : pl[q - LW];
            n = ((kinds >> 6) & 3) ? outside((kinds >> 6) & 3, g_n, c) : pl[q + LW];
        }
        double nl, dnl;
        nl_terms(ph, c, B, Pi, zb, mk, nl, dnl);
        const double lofphi = lofphi_cell(v, at, c, e, w, n, sv, bxE, bxW, byN, byS, nl);
        const double lam = lambda_cell(v, at, bxE, bxW, byN, byS);
        const double denom = 1.0e-16 + lam + dnl;                      // ...OpF.ChF:154
        pl[q] = c + (rhs - lofphi) / denom;                            // :156 (a cell of one colour reads cells of the other only: in place)
    };
    for (int m = 0; m < npass; m++) {
        const int reach = npass - 1 - m;                               // the pass advances the cells of colour m & 1 within `reach` of the tile
        if ((m & 1) == 0) {
#pragma unroll
            for (int s = 0; s < S; s++)
                if (cb[s][0] >= 0 && cd[s][0] <= reach)
                    update(cq[s][0], nk[s][0], hW[s][0], hE[s][0], hS[s][0], hN[s][0], c_rhs[s][0], c_bxW[s][0], c_bxE[s][0], c_byS[s][0], c_byN[s][0], c_B[s][0],
                        c_Pi[s][0], c_zb[s][0], c_mk[s][0], c_at[s][0]);
        } else {
#pragma unroll
            for (int s = 0; s < S; s++)
                if (cb[s][1] >= 0 && cd[s][1] <= reach)
                    update(cq[s][1], nk[s][1], hW[s][1], hE[s][1], hS[s][1], hN[s][1], c_rhs[s][1], c_bxW[s][1], c_bxE[s][1], c_byS[s][1], c_byN[s][1], c_B[s][1],
                        c_Pi[s][1], c_zb[s][1], c_mk[s][1], c_at[s][1]);
        }
        __syncthreads();
    }
    // the tile to the second canvas; the box's ghost ring goes along unchanged (first tile): the coarse-fine ghosts stay what they are, the
    // fine-fine ones are not read by this kernel and are refreshed by the next exchange
    const double *__restrict__ psrc = fk.f[fsrc];
    double *__restrict__ pdst = fk.f[fdst];
    // bcg: the ghost cells beyond the box's PHYSICAL sides get the homogeneous boundary condition of the final values as well (levelGSRB's closing
    // fill, src/AMRNonLinearPoissonOp.cpp:757-759; d_fill_ghosts' expression on the same operand) instead of a launch of their own; the caller sets
    // it only when no box of the level is its own periodic neighbour (that ghost would be another tile's cell)
    const bool gW = bcg && !v.cfx[0] && !v.per[0], gE = bcg && !v.cfx[1] && !v.per[0], gS = bcg && !v.ext[0] && !v.per[1], gN = bcg && !v.ext[1] && !v.per[1];
    for (int q = tid; q < tw * th; q += BOXNT) {
        const int jj = q / tw, ii = q - jj * tw, gi = x0 + ii, gj = y0 + jj, idx = cidx(v, gi, gj);
        const double c = pl[(BOXG + jj) * LW + BOXG + ii];
        pdst[idx] = c;
        if (gW && gi == 0) pdst[idx - 1] = phiW(v, pdst, idx, gi, c, true);
        if (gE && gi == v.nx - 1) pdst[idx + 1] = phiE(v, pdst, idx, gi, c, true);
        if (gS && gj == 0) pdst[idx - v.P] = phiS(v, pdst, idx, gj, c, true);
        if (gN && gj == v.ny - 1) pdst[idx + v.P] = phiN(v, pdst, idx, gj, c, true);
    }
    if (blockIdx.x == 0)
        for (int q = tid; q < 2 * (v.nx + 2) + 2 * v.ny; q += BOXNT) {
            int ii, jj;
            bool filled;                                                // (a side cell the loop above writes)
            if (q < v.nx + 2) { ii = q - 1; jj = -1; filled = gS && ii >= 0 && ii < v.nx; }
            else if (q < 2 * (v.nx + 2)) { ii = q - (v.nx + 2) - 1; jj = v.ny; filled = gN && ii >= 0 && ii < v.nx; }
            else if (q < 2 * (v.nx + 2) + v.ny) { ii = -1; jj = q - 2 * (v.nx + 2); filled = gW; }
            else { ii = v.nx; jj = q - 2 * (v.nx + 2) - v.ny; filled = gE; }
            if (filled) continue;
            const int idx = cidx(v, ii, jj);
            pdst[idx] = psrc[idx];
        }
}
// `npass` colour passes (up to 8 = 4 sweeps) of every box of the level in one launch, fsrc -> fdst
template <int G, int T, int S>
static void launch_box(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, const void *halo, const int *hbase, int fsrc, int fdst, int npass, int bc_ghosts, hipStream_t st)
{
    const dim3 grd(((m.maxnx + T - 1) / T) * ((m.maxny + T - 1) / T), m.nbox);
    if (has_alpha) hipLaunchKernelGGL((k_gsrb_box_m<true, G, T, S>), grd, dim3(BoxGeom<G, T, S>::NT), 0, st, m.dv, m.fp, (const int2 *)halo, hbase, ph, fsrc, fdst, npass, bc_ghosts);
    else hipLaunchKernelGGL((k_gsrb_box_m<false, G, T, S>), grd, dim3(BoxGeom<G, T, S>::NT), 0, st, m.dv, m.fp, (const int2 *)halo, hbase, ph, fsrc, fdst, npass, bc_ghosts);
}
int suhmo_multi_gsrb_box(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, const void *halo, const int *hbase, int fsrc, int fdst, int npass, int bc_ghosts, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    if (npass < 1 || npass > 2 * SUHMO_BOX_HALO) { suhmo_set_error("internal: %d colour passes in one box launch", npass); return -4; }
    if (npass > 4) launch_box<8, 16, 1>(m, ph, has_alpha, halo, hbase, fsrc, fdst, npass, bc_ghosts, st);
    else launch_box<4, 16, 1>(m, ph, has_alpha, halo, hbase, fsrc, fdst, npass, bc_ghosts, st);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- variant 1: K sweeps fused, streaming over rows ----
struct FusedGeom {
    int W;        // owned columns per strip (even)
    int Hc;       // owned rows per chunk
    int nstrips, nchunks, ntiles;
    int ylo, yhi; // rows that exist for loading / computing (strip-local j, inclusive)
    int jbeg, jend; // rows written: [jbeg, jend) = the strip's rows plus, on rank boundaries, the halo rows that
                    // stay valid after this launch (advanced redundantly so the next launch needs no exchange)
    int wrap_y;   // rows outside [0, ny) are periodic images (single-rank periodic y)
    const unsigned *negflag; unsigned mask_epoch;   // *negflag == mask_epoch: the ice mask has a negative cell (k_bcoef_fused of this V-cycle); NULL: read the mask
    int nchl, chunk0, chunk1, edges;   // chunks of THIS launch: nchl of them starting at chunk0, or (edges) all but [chunk0, chunk1) --
                               // a rank strip relaxes its inner chunks while the halo rows travel, the end chunks afterwards
    // FAS prolongIncrement fused into the load of phi (PROLONGNL, AMRNonLinearPoissonOpF.ChF:619-627):
    // phi(i,j) += phi_c(i/2,j/2) - phi_c_old(i/2,j/2)
    const double *pc, *pco; int Pc, gyc;
    // RST: restrictResidual + restrictR fused into the launch that finishes the pre-smoothing: coarse RES / PHI canvases
    double *rres, *rphi; int rP, rgy;
    // FRHS: the first relaxation of a coarse FAS depth forms the depth's right-hand side rhs = res + L(R phi) itself (k_apply<., 2>'s
    // expressions on the rows as they are loaded) and stores it, L(phi) and the copy of R phi the prolongation subtracts later
    const double *fres; double *frhs, *flphi, *fphiold;
    // RM = 2: the launch that ends the cycle's post-smoothing leaves the residual of the final phi behind (the solve loop's residual
    // evaluation, k_apply<., 1> / <., 3>): RES = rtrue - L(phi), optionally L(phi) too; rtrue = the right-hand side the CALLER's residual is
    // about (inside an AMR cycle the level relaxes against its FAS right-hand side while the true one waits on the second canvas)
    const double *ortrue; double *ores, *olphi;
    double *onorm;      // ... and the chunk's max |RES| (one partial per strip x chunk: k_norm_partial's job, for k_norm_final); NULL: not asked for
};

struct RowCoef {          // per-thread coefficients of its column pair in one row
    double rhs[2], B[2], Pi[2], zb[2], mask[2], a[2], byS[2], byN[2];
    double bx0, bx1, bx2;
};
#define CP2(d, s, f) d.f[0] = s.f[0]; d.f[1] = s.f[1]
template <bool HAS_ALPHA>
__device__ __forceinline__ void copy_coef(RowCoef &d, const RowCoef &s)
{
    CP2(d, s, rhs); CP2(d, s, B); CP2(d, s, Pi); CP2(d, s, zb); CP2(d, s, mask);
    if (HAS_ALPHA) { CP2(d, s, a); }
    CP2(d, s, byS); CP2(d, s, byN);
    d.bx0 = s.bx0; d.bx1 = s.bx1; d.bx2 = s.bx2;
}

// RST = true: the launch also restricts.  Once row j-1, j, j+1 are final (all 2K half-sweeps), the residual rhs - L(phi) of
// row j is evaluated from the ring with the same expressions as k_restrict_residual and accumulated, in the reference's
// visiting order, into the coarse cell of the thread's column pair (RESTRICTRESVCNL2D + RESTRICTVCNL, VCAMR...OpF.ChF:480-561,
// 419-449): the separate pass over phi and the 8 coefficient arrays (75 B/cell) disappears.  Costs: one more ring row, one
// more coefficient row, one more final row above and below the chunk and two more halo columns per side.
template <int K, bool HAS_ALPHA, int NT, int RM = 0, bool FRHS = false>
__global__ __launch_bounds__(NT) void k_gsrb_fused(DV v, FP fp, const double *__restrict__ pin,
                                                   double *__restrict__ pout, suhmo_phys_t ph, FusedGeom g)
{
    // RM: what the launch does with the final rows besides storing them: 0 nothing, 1 restricts (RST), 2 stores their residual (ROUT)
    constexpr bool ROUT = RM == 2, RR = RM != 0;
    static_assert(!(FRHS && (RR || HAS_ALPHA)), "the right-hand side is formed in the plain launch of an alpha = 0 operator");
    constexpr int LW = 2 * NT, R = 2 * K + 3 + (RR ? 1 : 0);
    constexpr int HX = 2 * K + (RR ? 2 : 0), EY = RR ? 1 : 0;
    __shared__ double lds[R * LW];

    // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs; give every XCD a
    // contiguous range of tiles (adjacent chunks of one strip share 2K halo rows in its L2).
    int b = blockIdx.x, q = g.ntiles / 8, rem = g.ntiles % 8, xcd = b % 8;
    int tile = xcd * q + (xcd < rem ? xcd : rem) + b / 8;
    int strip = tile / g.nchl, chunk = tile % g.nchl;
    chunk = g.edges ? (chunk < g.chunk0 ? chunk : g.chunk1 + (chunk - g.chunk0)) : g.chunk0 + chunk;

    const int t = threadIdx.x;
    const int c0 = strip * g.W;
    const int xl = 2 * t;                      // position of the pair's first cell in an LDS row
    const int i0 = c0 - HX + xl;               // its global column (even)
    const bool in_row = xl < g.W + 2 * HX;
    int im = i0;
    if (v.per[0]) { if (im < 0) im += v.nx; else if (im >= v.nx) im -= v.nx; }
    const bool cval = in_row && im >= 0 && im < v.nx;                 // pair lies in the domain
    const int cend = (c0 + g.W < v.nx) ? c0 + g.W : v.nx;
    const bool own = cval && i0 >= c0 && i0 < cend;

    const int jA = g.jbeg + chunk * g.Hc;
    const int jB = (jA + g.Hc < g.jend) ? jA + g.Hc : g.jend;
    const int jmin = (jA - EY - 2 * K > g.ylo) ? jA - EY - 2 * K : g.ylo;
    const int jmax = (jB - 1 + EY + 2 * K < g.yhi) ? jB - 1 + EY + 2 * K : g.yhi;

    const double *__restrict__ f_rhs = fp.f[SUHMO_F_RHS], *__restrict__ f_B = fp.f[SUHMO_F_B];
    const double *__restrict__ f_Pi = fp.f[SUHMO_F_PI], *__restrict__ f_zb = fp.f[SUHMO_F_ZB];
    const double *__restrict__ f_mask = fp.f[SUHMO_F_MASK], *__restrict__ f_a = fp.f[SUHMO_F_ACOEF];
    const double *__restrict__ f_bx = fp.f[SUHMO_F_BX], *__restrict__ f_by = fp.f[SUHMO_F_BY];

    auto wrapj = [&](int j) { if (g.wrap_y) { if (j < 0) j += v.ny; else if (j >= v.ny) j -= v.ny; } return j; };
    auto ld2 = [&](const double *__restrict__ p, int idx) { return *reinterpret_cast<const double2 *>(p + idx); };

    // coefficient ring as NAMED variables (an indexed array ends up in scratch memory):
    // cf0 = row r (being prefetched), cfM = row r-M
    RowCoef cf0, cf1, cf2, cf3, cf4, cf5;
    double racc = 0.0, raccp = 0.0;        // RST: the coarse cell's sums after its first fine row
    double nmax = 0.0;                     // ROUT: max |RES| over this thread's cells of the chunk
    // physical-BC sides this tile can touch (uniform): skip the per-lane boundary tests elsewhere
    const bool xbc = !v.per[0] && (c0 - HX <= 0 || c0 + g.W + HX >= v.nx);
    const bool ybc = !v.per[1] && (jmin <= 0 || jmax >= v.ny - 1);
    double2 pnext = make_double2(0.0, 0.0);
    double2 pprev = make_double2(0.0, 0.0);    // FRHS: this thread's pair of the row below the one whose L(phi) is formed, as loaded
    // the mask array is 8 of the 80 bytes a cell costs per launch and only its sign is used: when this V-cycle's UpdateOperator saw
    // no negative cell the loads are skipped (uniform)
    const bool usemask = !g.negflag || *g.negflag == g.mask_epoch;

    int sr = 0;                            // LDS ring slot of row r
    // slot of an earlier row: sr - m, m <= R -- a compare instead of a division by 7 (the kernel issues nearly as many scalar as vector
    // instructions: 290.7 against 294.7 us per launch, profiles/r03_i_ring_index_ab.txt; keeping the seven offsets in rotating scalars
    // instead: fewer instructions, more spilled scalars, the same time)
    auto ring = [](int x) { return x < 0 ? x + R : x; };
    for (int r = jmin - 1; r <= jB - 1 + 2 * K + EY; r++) {
        // ---- 1. prefetch: phi of row r+1, coefficients of row r (both first used in step r+1)
        bool lphi = cval && (r + 1 >= jmin) && (r + 1 <= jmax);
        if (lphi) {
            const int jr = wrapj(r + 1);
            pnext = ld2(pin, cidx(v, im, jr));
            if (g.pc) {
                const int ic = ((jr >> 1) + g.gyc) * g.Pc + SUHMO_XOFF + (im >> 1);
                const double corr = 1.0 * g.pc[ic] + (-1.0) * g.pco[ic];      // axby(1, -1), then PROLONGNL
                pnext.x = pnext.x + corr; pnext.y = pnext.y + corr;
            }
        }
        bool lcf = cval && (r >= jmin) && (r <= jmax);
        if (lcf) {
            int idx = cidx(v, im, wrapj(r));
#define LD2(dst, p, ix) { double2 t_ = ld2(p, ix); dst[0] = t_.x; dst[1] = t_.y; }
            LD2(cf0.rhs, (FRHS ? g.fres : f_rhs), idx); LD2(cf0.B, f_B, idx); LD2(cf0.Pi, f_Pi, idx);
            LD2(cf0.zb, f_zb, idx);
            if (usemask) { LD2(cf0.mask, f_mask, idx); } else { cf0.mask[0] = 1.0; cf0.mask[1] = 1.0; }
            if (HAS_ALPHA) LD2(cf0.a, f_a, idx);
            LD2(cf0.byS, f_by, idx); LD2(cf0.byN, f_by, idx + v.P);
            double2 bxp = ld2(f_bx, idx);
            cf0.bx0 = bxp.x; cf0.bx1 = bxp.y; cf0.bx2 = f_bx[idx + 2];
        }
        __syncthreads();                   // row r (written at the end of step r-1) is visible

        // ---- 1b. FRHS: rows r-2 (this thread's own pair, kept in pprev), r-1 and r are still as loaded: L(phi) of row r-1, its
        // right-hand side (used by the half-sweeps below: cf1 IS row r-1), and the three stores of k_apply<., 2> for the chunk's rows
        if constexpr (FRHS) {
            const int jf = r - 1;
            if (cval && jf >= jmin && jf <= jmax) {
                const int s0 = ring(sr - 1), sN = sr;
                const double *row = lds + s0 * LW;
                double lo[2], cc[2];
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int x = xl + a, i = im + a;
                    double c = row[x];
                    double w = row[(a == 0 && xl == 0) ? 0 : x - 1], e = row[(a == 1 && xl == LW - 2) ? LW - 1 : x + 1];
                    double n = lds[sN * LW + x], s = a ? pprev.y : pprev.x;
                    if (xbc) {
                        if (i == 0) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                        if (i == v.nx - 1) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
                    }
                    if (ybc) {
                        if (jf == 0 && !v.ext[0]) s = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                        if (jf == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
                    }
                    double nl, dnl;
                    nl_terms(ph, c, cf1.B[a], cf1.Pi[a], cf1.zb[a], cf1.mask[a], nl, dnl);
                    const double bxW = a ? cf1.bx1 : cf1.bx0, bxE = a ? cf1.bx2 : cf1.bx1;
                    lo[a] = lofphi_cell(v, v.alpha, c, e, w, n, s, bxE, bxW, cf1.byN[a], cf1.byS[a], nl);
                    cc[a] = c;
                }
                cf1.rhs[0] = 1.0 * cf1.rhs[0] + 1.0 * lo[0]; cf1.rhs[1] = 1.0 * cf1.rhs[1] + 1.0 * lo[1];
                // rank strip: the end chunks also keep what they formed in the halo rows they only read (the later launches of the depth
                // advance halo rows too and the prolongation corrects them): the right-hand side wherever both row neighbours were loaded,
                // R phi in every loaded row -- the rows k_apply<., 2> serves with hcomp = halo - 1
                const bool lo_end = v.ext[0] && chunk == 0, hi_end = v.ext[1] && chunk == g.nchunks - 1;
                const int jP0 = lo_end ? jmin : jA, jP1 = hi_end ? jmax + 1 : jB;
                const int jR0 = lo_end ? jmin + 1 : jA, jR1 = hi_end ? jmax : jB;
                if (own && jf >= jP0 && jf < jP1) {
                    const int idx = cidx(v, i0, jf);
                    if (jf >= jR0 && jf < jR1) {
                        *reinterpret_cast<double2 *>(g.flphi + idx) = make_double2(lo[0], lo[1]);
                        *reinterpret_cast<double2 *>(g.frhs + idx) = make_double2(cf1.rhs[0], cf1.rhs[1]);
                    }
                    *reinterpret_cast<double2 *>(g.fphiold + idx) = make_double2(cc[0], cc[1]);
                }
                pprev = make_double2(cc[0], cc[1]);            // row r-1 as loaded: the south neighbours of row r in the next step
            }
            __syncthreads();                                   // the half-sweep below overwrites cells of row r-1 that the neighbours read above
        }

        // ---- 2. advance row r-m from half-sweep m-1 to m, m = 1..2K
        // One half-sweep of row r-m.  The colour offset `a` (which cell of the pair is updated)
        // depends only on the row, so it is WAVE-UNIFORM: branch on it once (scalar branch) and
        // address the pair's coefficients statically instead of selecting per lane.
        auto advance_a = [&](const int m, const RowCoef &q, auto a_tag) {
            constexpr int a = decltype(a_tag)::value;
            const int j = r - m;
            const int x = xl + a, i = im + a;
            const int s0 = ring(sr - m), sN = ring(sr - m + 1), sS = ring(sr - m - 1);
            const double *row = lds + s0 * LW;
            double c = row[x];
            double w = row[(a == 0 && xl == 0) ? 0 : x - 1], e = row[(a == 1 && xl == LW - 2) ? LW - 1 : x + 1];
            double n = lds[sN * LW + x], s = lds[sS * LW + x];
            if (xbc) {                                         // mixBCValues on the fly (strip touches a BC side)
                if (i == 0) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                if (i == v.nx - 1) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
            }
            if (ybc) {
                if (j == 0 && !v.ext[0]) s = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                if (j == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
            }
            double nl, dnl;
            nl_terms(ph, c, q.B[a], q.Pi[a], q.zb[a], q.mask[a], nl, dnl);
            const double bxW = a ? q.bx1 : q.bx0, bxE = a ? q.bx2 : q.bx1;
            double aterm = HAS_ALPHA ? v.alpha * q.a[a] : v.alpha;
            double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, q.byN[a], q.byS[a], nl);
            double lam = lambda_cell(v, aterm, bxE, bxW, q.byN[a], q.byS[a]);
            double denom = 1.0e-16 + lam + dnl;
            lds[s0 * LW + x] = c + (q.rhs[a] - lofphi) / denom;
        };
        auto advance = [&](const int m, const RowCoef &q) {
            const int j = r - m;
            if (j >= jmin && j <= jmax) {                      // uniform
                const int a = (j + v.j0 + ((m - 1) & 1)) & 1;  // uniform: colour offset of this row
                if (a) { if (cval) advance_a(m, q, std::integral_constant<int, 1>()); }
                else   { if (cval) advance_a(m, q, std::integral_constant<int, 0>()); }
            }
        };
        advance(1, cf1);
        __syncthreads();
        advance(2, cf2);
        if constexpr (K >= 2) {
            __syncthreads();
            advance(3, cf3);
            __syncthreads();
            advance(4, cf4);
        }

        // ---- 2b. RST: rows r-2K-2 .. r-2K are final: residual of row r-2K-1, restricted
        if constexpr (RR) {
            const int jr = r - 2 * K - 1;
            if (own && jr >= jA && jr < jB && jr >= 0 && jr < v.ny) {    // (rank strips: halo rows are advanced, not restricted)
                const RowCoef &q = (K >= 2) ? cf5 : cf3;
                const int s0 = ring(sr - (2 * K + 1)), sN = ring(sr - 2 * K), sS = ring(sr - (2 * K + 2));
                const double *row = lds + s0 * LW;
                double acc = (jr & 1) ? racc : 0.0, accp = (jr & 1) ? raccp : 0.0;
                double lo2[2] = {0.0, 0.0};
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int x = xl + a, i = im + a;
                    double c = row[x], w = row[x - 1], e = row[x + 1];       // own pairs never sit on the edge of the LDS row (HX >= 2)
                    double n = lds[sN * LW + x], s = lds[sS * LW + x];
                    if (xbc) {
                        if (i == 0) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                        if (i == v.nx - 1) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
                    }
                    if (ybc) {
                        if (jr == 0 && !v.ext[0]) s = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                        if (jr == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
                    }
                    double nl, dnl;
                    nl_terms(ph, c, q.B[a], q.Pi[a], q.zb[a], q.mask[a], nl, dnl);
                    const double bxW = a ? q.bx1 : q.bx0, bxE = a ? q.bx2 : q.bx1;
                    double aterm = HAS_ALPHA ? v.alpha * q.a[a] : v.alpha;
                    double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, q.byN[a], q.byS[a], nl);
                    if constexpr (ROUT) lo2[a] = lofphi;
                    else {
                        acc = acc + (q.rhs[a] - lofphi) / 4.0;
                        accp = accp + c / 4.0;
                    }
                }
                if constexpr (ROUT) {
                    const int idx = cidx(v, i0, jr);
                    const double2 rt = *reinterpret_cast<const double2 *>(g.ortrue + idx);
                    const double r0 = -1.0 * lo2[0] + 1.0 * rt.x, r1 = -1.0 * lo2[1] + 1.0 * rt.y;
                    *reinterpret_cast<double2 *>(g.ores + idx) = make_double2(r0, r1);
                    if (g.olphi) *reinterpret_cast<double2 *>(g.olphi + idx) = make_double2(lo2[0], lo2[1]);
                    nmax = fmax(nmax, fmax(fabs(r0), fabs(r1)));
                } else if (jr & 1) {
                    const int ic = ((jr >> 1) + g.rgy) * g.rP + SUHMO_XOFF + (i0 >> 1);
                    g.rres[ic] = acc; g.rphi[ic] = accp;
                } else { racc = acc; raccp = accp; }
            }
        }
        // ---- 3. row r-2K has all 2K half-sweeps: stream it out (own pair, written by this thread)
        {
            const int jo = r - 2 * K;
            if (own && jo >= jA && jo < jB) {
                const int so = ring(sr - 2 * K);
                double2 o = make_double2(lds[so * LW + xl], lds[so * LW + xl + 1]);
                *reinterpret_cast<double2 *>(pout + cidx(v, i0, jo)) = o;
            }
        }
        // ---- 4. row r+1 enters the ring (its slot held row r-2K-2: no longer read), rotate
        sr = sr + 1 == R ? 0 : sr + 1;
        if (in_row) { lds[sr * LW + xl] = pnext.x; lds[sr * LW + xl + 1] = pnext.y; }
        if constexpr (RR && K >= 2) copy_coef<HAS_ALPHA>(cf5, cf4);
        if constexpr (K >= 2) { copy_coef<HAS_ALPHA>(cf4, cf3); copy_coef<HAS_ALPHA>(cf3, cf2); }
        else if constexpr (RR) copy_coef<HAS_ALPHA>(cf3, cf2);
        copy_coef<HAS_ALPHA>(cf2, cf1);
        copy_coef<HAS_ALPHA>(cf1, cf0);
    }
    if constexpr (ROUT) {
        if (g.onorm) {                         // (uniform) the chunk's max norm: a maximum, so the order of the lanes does not matter
            static_assert(!ROUT || NT == 64, "one wave per workgroup");
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nmax = fmax(nmax, __shfl_xor(nmax, o));
            if (t == 0) g.onorm[strip * g.nchunks + chunk] = nmax;
        }
    }
}

static bool fused_ok(const suhmo_level *L, const Depth &D, int K)
{
    const DV &v = D.v;
    if (v.nx % 2 || v.nx < 64 || v.ny < 4 * K + 4) return false;
    if (v.cfx[0] || v.cfx[1] || L->desc.nx_global > 0) return false;   // AMR patch: colour-pass kernel (stored coarse-fine ghosts)
    if ((v.ext[0] || v.ext[1]) && (v.gy < 2 * K || v.ny < 2 * K)) return false;   // halo rows must be real data
    if (v.per[1] && !(v.ext[0] || v.ext[1]) && v.ny < 4 * K) return false;
    return true;
}

// part 0: the whole level.  Rank strips with the exchange in flight: part 1 = the chunks that read no halo row (returns 1 and
// launches nothing when the geometry has none), part 2 = the first and the last chunk, then the canvases trade places.
template <int K, int NT, int RM = 0>
static int launch_fused(suhmo_level *L, int depth, int ext_rows, hipStream_t st, int part = 0)
{
    constexpr bool RST = RM == 1, RR = RM != 0;
    Depth &D = L->d[depth];
    const DV &v = D.v;
    if (!D.phi_alt) {
        HIPCHK(hipMalloc(&D.phi_alt, D.elems * sizeof(double)));
        HIPCHK(hipMemsetAsync(D.phi_alt, 0, D.elems * sizeof(double), st));
    }
    FusedGeom g;
    constexpr int HX = 2 * K + (RR ? 2 : 0), EY = RR ? 1 : 0;
    const int maxW = 2 * NT - 2 * HX;
    g.nstrips = (v.nx + maxW - 1) / maxW;
    g.W = 2 * ((v.nx + 2 * g.nstrips - 1) / (2 * g.nstrips));
    g.nstrips = (v.nx + g.W - 1) / g.W;
    // rows per chunk: all tiles resident in ONE round (tiles <= workgroup slots of the chip;
    // a partial second round costs a full one), but never shorter than 16K rows so that the
    // 4K-row pipeline fill stays small; measured on MI355X: profiles/r01_b_hc_sweep.log
    {
        static int slots_k[3] = {0, 0, 0};     // per (K, NT, RST) instantiation
        if (!slots_k[K]) {
            int nb = 0, ncu = 0, dev = 0;
            HIPCHK(hipGetDevice(&dev));
            HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_gsrb_fused<K, false, NT, RM>), NT, 0));
            slots_k[K] = (nb > 0 ? nb : 1) * (ncu > 0 ? ncu : 256);
        }
        g.jbeg = v.ext[0] ? -ext_rows : 0;
        g.jend = v.ny + (v.ext[1] ? ext_rows : 0);
        const int nrows = g.jend - g.jbeg;
        int nch = slots_k[K] / g.nstrips;
        if (nch < 1) nch = 1;
        g.Hc = (nrows + nch - 1) / nch;
        // (not shorter than 8K rows: the 4K-row pipeline fill.  The one-round height itself is the optimum where it is allowed: 2048^2,
        //  19 rows: 45.1 us per sweep, 24 rows 51.4, 32 rows -- two rounds, the old lower bound -- 60.3, profiles/r03_stream2048.txt)
        if (g.Hc < 8 * K) g.Hc = 8 * K;
        // cache-resident depths (about 1 M cells) are latency-bound: many short chunks beat few long ones
        // (profiles/r01_k_sweep_small_hc.log: 1024^2, K = 2: 22.3 us per sweep at 6 rows vs 28.1 for the colour passes)
        if ((long)v.nx * v.ny < 2000000L) g.Hc = 6;
        if (L->fused_hc > 0) g.Hc = L->fused_hc;          // explicit: no lower clamp (tools/sweep_small_hc.sh)
        if (RST) g.Hc += g.Hc & 1;                        // a coarse cell's two fine rows stay in one chunk
        if (g.Hc > nrows) g.Hc = nrows;
        g.nchunks = (nrows + g.Hc - 1) / g.Hc;
    }
    g.nchl = g.nchunks; g.chunk0 = 0; g.chunk1 = g.nchunks; g.edges = 0;
    if (part) {
        // inner chunks [c0, c1): their rows and the 2K (+1) rows either side that they read are the strip's own (a side that is a
        // physical boundary has no halo rows to wait for)
        int c0 = 0, c1 = g.nchunks;
        if (v.ext[0]) while (c0 < g.nchunks && g.jbeg + c0 * g.Hc - 2 * K - EY < 0) c0++;
        if (v.ext[1]) while (c1 > c0 && std::min(g.jbeg + c1 * g.Hc, g.jend) - 1 + 2 * K + EY > v.ny - 1) c1--;
        const bool ok = c1 > c0 && c1 - c0 < g.nchunks;
        if (!ok) { if (part == 1) return 1; }
        else if (part == 1) { g.chunk0 = c0; g.chunk1 = c1; g.nchl = c1 - c0; }
        else { g.edges = 1; g.chunk0 = c0; g.chunk1 = c1; g.nchl = g.nchunks - (c1 - c0); }
    }
    g.ntiles = g.nstrips * g.nchl;
    bool selfper = v.per[1] && !(v.ext[0] || v.ext[1]);
    g.wrap_y = selfper;
    g.ylo = v.ext[0] ? g.jbeg - 2 * K - EY : (selfper ? -2 * K - EY : 0);
    g.yhi = v.ext[1] ? g.jend - 1 + 2 * K + EY : (selfper ? v.ny - 1 + 2 * K + EY : v.ny - 1);
    const double *pin = D.fp.f[SUHMO_F_PHI];
    g.pc = g.pco = nullptr; g.Pc = g.gyc = 0;
    if (D.prolong_pending) {
        Depth &C = L->d[depth + 1];
        g.pc = C.fp.f[SUHMO_F_PHI]; g.pco = C.fp.f[SUHMO_F_PHIOLD]; g.Pc = C.v.P; g.gyc = C.v.gy;
        if (part != 1) D.prolong_pending = 0;
    }
    g.negflag = nullptr; g.mask_epoch = 0;
    // (the report is about depth 0's mask and, on a rank strip, the stored halo rows of every depth; the coarse masks are averages of
    //  depth 0's, MGnewOp: none of them is negative either -- as long as nobody has written one since, coarse_mask_ok)
    if (L->skip_mask && L->maskflag_epoch && L->maskflag_epoch == L->mask_epoch && (depth == 0 || L->coarse_mask_ok)
        && (L->desc.nx_global == 0)) {
        g.negflag = (const unsigned *)(L->scratch + L->scratch_elems - 1); g.mask_epoch = L->mask_epoch;
    }
    g.rres = g.rphi = nullptr; g.rP = g.rgy = 0;
    if (RST) {
        Depth &C = L->d[depth + 1];
        g.rres = C.fp.f[SUHMO_F_RES]; g.rphi = C.fp.f[SUHMO_F_PHI]; g.rP = C.v.P; g.rgy = C.v.gy;
        C.phi_fresh = 0;
    }
    g.fres = nullptr; g.frhs = g.flphi = g.fphiold = nullptr;
    g.ortrue = nullptr; g.ores = g.olphi = nullptr;
    if (RM == 2) {
        if (v.alpha != 0.0) { suhmo_set_error("internal: residual output on a launch that cannot form it"); return -4; }
        g.ortrue = L->resout_rhs ? L->resout_rhs : D.fp.f[SUHMO_F_RHS];
        g.ores = D.fp.f[SUHMO_F_RES];
        if (L->resout_req & 2) { g.olphi = suhmo_field(L, depth, SUHMO_F_LPHI); if (!g.olphi) return -2; }
        g.onorm = nullptr;
        if ((L->resout_req & 4) && (size_t)g.nstrips * g.nchunks + 4 < L->scratch_elems) { g.onorm = L->scratch + 2; L->resout_np = g.nstrips * g.nchunks; }
        else L->resout_np = 0;
    }
    if (D.rhs_pending) {
        // (suhmo_gsrb_can_fuse_rhs said yes for exactly this launch: whole level, two sweeps, one-wave workgroups, alpha = 0)
        if constexpr (K == 2 && NT == 64 && !RR) {
            if (part || v.alpha != 0.0) { suhmo_set_error("internal: rhs_pending on a launch that cannot form it"); return -4; }
            g.fres = D.fp.f[SUHMO_F_RES]; g.frhs = D.fp.f[SUHMO_F_RHS]; g.flphi = suhmo_field(L, depth, SUHMO_F_LPHI); g.fphiold = suhmo_field(L, depth, SUHMO_F_PHIOLD);
            if (!g.flphi || !g.fphiold) return -2;
            D.rhs_pending = 0; L->frhs_stream++;
            hipLaunchKernelGGL((k_gsrb_fused<2, false, 64, false, true>), dim3(g.ntiles), dim3(NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
            { std::swap(D.fp.f[SUHMO_F_PHI], D.phi_alt); suhmo_fp_changed(); }
            return 0;
        } else { suhmo_set_error("internal: rhs_pending on a launch that cannot form it"); return -4; }
    }
    if (v.alpha != 0.0)
        hipLaunchKernelGGL((k_gsrb_fused<K, true, NT, false>), dim3(g.ntiles), dim3(NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
    else
        hipLaunchKernelGGL((k_gsrb_fused<K, false, NT, RM>), dim3(g.ntiles), dim3(NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
    if (part != 1) { std::swap(D.fp.f[SUHMO_F_PHI], D.phi_alt); suhmo_fp_changed(); }
    return 0;
}

// ---- variant 3: S sweeps of a T x T tile in LDS (cache-resident depths) ----
// Depths of up to about a million cells are latency-bound: a colour pass is one launch of ~5 us whatever its size (the
// dependent launch, the first touch of data the previous kernel wrote on another XCD), so a smoothing of 4 sweeps costs 8 of
// them.  Here a workgroup loads its tile plus a halo of 2S cells into LDS, keeps the coefficients of its cells in registers
// and runs all 2S colour passes between barriers; pass p is valid one cell further in from the halo's edge than pass p-1
// (sides that are physical boundaries do not shrink: the boundary condition is evaluated on the fly there), so after 2S
// passes exactly the tile is current.  Every update is the expression of k_gsrb_pass_simple on the same operands: the result
// is bitwise the same, the halo work is redundant.  Output goes to the second phi canvas (neighbour tiles read the old one).
struct TileGeom {
    int ntx, nty;
    const double *pc, *pco; int Pc, gyc;     // prolongIncrement fused into the load (as in FusedGeom)
    double *rres, *rphi; int rP, rgy;        // RST: coarse RES / PHI canvases
    int frhs;                                // first relaxation of a coarse FAS depth: rhs = res + L(phi) is formed here
    int jbeg, jend;                          // rows written: the level's, plus on a rank strip the halo rows that stay current for the next launch
    int chunks;                              // level = one tile: this many times S sweeps in the launch (halo images refreshed in LDS)
    int order;                               // workgroup -> tile: 0 as launched, 1 / 2 XCD-aware (see k_gsrb_tile)
};
// (the timing probes that switch parts of the tile kernel off -- wrong results on purpose -- live in a copy of this file:
//  tools/probes/suhmo_gsrb_tile_probe.hip, built by tools/probes/tile_probe.sh; nothing of them is in the product kernel)
// (of the ice mask COMPUTENONLINEARTERMS uses the sign only: mneg bit 0 / 1 = the first / second cell of the pair is without ice)
struct PairCoef { double rhs0, rhs1, B0, B1, Pi0, Pi1, zb0, zb1, a0, a1, byS0, byS1, byN0, byN1, bx0, bx1, bx2; int mneg; };

// RST = true: the launch also restricts (as k_gsrb_fused<.., RST>): one more ring of final values around the tile (two columns,
// pairs stay aligned), the residual quarters of the tile's cells go through LDS and one thread per coarse cell adds the four in
// the reference's visiting order (RESTRICTRESVCNL2D + RESTRICTVCNL).  The 32-wide tile is 28 rows high then: the same number
// of column pairs per thread, hence the same registers, as without the restriction.
template <int T, bool RST> struct TileShape { static constexpr int TX = T, TY = (RST && T == 32) ? 28 : T; };
// Threads per workgroup: 256 (5 column pairs per thread at T = 32, S = 4: 2 waves per SIMD).  Measured and dropped (profiles/r03_tile_*):
// 384 threads (3 pairs per thread, 161-168 VGPRs, 3 waves per SIMD on paper) -- the six waves of a workgroup land 2 + 2 + 1 + 1 on the
// four SIMDs, a second workgroup no longer fits beside the first, and the 2048^2 launch takes 266 us instead of 203.
template <int S, int T, bool RST> struct TileThreads { static constexpr int NT = 256; };
template <int S, int T, bool HAS_ALPHA, bool RST = false, bool CHUNKED = false>
__global__ __launch_bounds__((TileThreads<S, T, RST>::NT)) __attribute__((amdgpu_waves_per_eu((TileThreads<S, T, RST>::NT == 384 ? 3 : 2)))) void k_gsrb_tile(DV v,
    FP fp, const double *__restrict__ pin, double *__restrict__ pout,
                                                   suhmo_phys_t ph, TileGeom g)
{
    constexpr int TX = TileShape<T, RST>::TX, TY = TileShape<T, RST>::TY;
    constexpr int HX = 2 * S + (RST ? 2 : 0), HY = 2 * S + (RST ? 1 : 0);
    constexpr int LX = TX + 2 * HX, LY = TY + 2 * HY, NP = LX / 2, NPAIR = NP * LY;
    constexpr int NT = TileThreads<S, T, RST>::NT, NK = (NPAIR + NT - 1) / NT;
    // (a margin of one row + two cells either side: the straight-line pass below also evaluates the cells on the region's edge, whose
    // results are dropped, and reads one cell beyond them)
    constexpr int PAD = LX + 2;
    __shared__ double lds_raw[LY * LX + 2 * PAD];
    double *const lds = lds_raw + PAD;
    __shared__ double lq[RST ? TY * TX : 1];                  // RST: (rhs - L(phi)) / 4 of the tile's cells
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: every XCD takes a contiguous run of the tile list, so
    // that the tiles in flight on it are neighbours and the halo cells they share (2.25 x the tile's own at S = 4) are read from HBM
    // once.  order 2 lists the tiles in panels of 8 tile rows, column by column: the 64 tiles an XCD holds at a time form a square.
    int tl = blockIdx.x;
    if (g.order) {
        const int nt = g.ntx * g.nty, q8 = nt / 8, rem = nt % 8, xcd = tl % 8;
        tl = xcd * q8 + (xcd < rem ? xcd : rem) + tl / 8;
    }
    int tx = tl % g.ntx, ty = tl / g.ntx;
    if (g.order == 2) {
        const int pan = tl / (8 * g.ntx), r = tl - pan * 8 * g.ntx, hp = (g.nty - 8 * pan < 8) ? g.nty - 8 * pan : 8;
        tx = r / hp; ty = 8 * pan + r % hp;
    }
    const int tj0 = g.jbeg + ty * TY;                         // first row the tile writes
    const int gx0 = tx * TX - HX, gy0 = tj0 - HY;             // domain cell of LDS cell (0, 0)
    const int t = threadIdx.x;
    // thread -> column pairs: the pairs of a WAVE lie in rows of one parity (waves 0, 2: even rows of the region; 1, 3: odd rows), so
    // which cell of its pairs a colour pass advances is the same for the whole wave: a scalar branch picks one of two straight-line
    // bodies with the pair's coefficients addressed statically, instead of per-lane selects
    const int par = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 1;
    const int u0 = ((t >> 7) << 6) | (t & 63);                 // this thread among the NT / 2 of its parity
    auto pair_pos = [&](const int k, int &ly, int &lx) -> bool {
        const int u = u0 + (NT / 2) * k, r = u / NP;
        ly = 2 * r + par; lx = 2 * (u - r * NP);
        return ly < LY;
    };
    // (measured and dropped, profiles/r03_tile_probe.txt: the rows of a parity taken tile rows first, then the halo rows from the tile
    //  outwards, so that the later slots of a thread hold halo rows only and the straight-line passes skip a slot as a whole once none of
    //  its rows is advanced any more -- 34 slot-passes instead of 40, bitwise, and the launch takes the same 203 us)
    static_assert(LY % 2 == 0, "rows of the region come in pairs of parities");
    // a coarse-fine side takes precedence over the domain's periodicity (the patch does not wrap onto itself)
    const bool perx = v.per[0] && !v.cfx[0] && !v.cfx[1], pery = v.per[1] && !v.ext[0] && !v.ext[1];
    // sides of the LDS region that reach a physical boundary: nothing beyond them feeds the tile
    // (a coarse-fine side of an AMR patch: once the stored ghost column / row is in the region)
    const bool openW = !perx && gx0 <= (v.cfx[0] ? -1 : 0), openE = !perx && gx0 + LX - 1 >= v.nx - 1 + (v.cfx[1] ? 1 : 0);
    const bool openS = !pery && !v.rk[0] && gy0 <= (v.ext[0] ? -1 : 0), openN = !pery && !v.rk[1] && gy0 + LY - 1 >= v.ny - 1 + (v.ext[1] ? 1 : 0);
    auto wrap = [](int i, int n) { i %= n; return i < 0 ? i + n : i; };
    auto ld2 = [&](const double *__restrict__ p, int idx) { return *reinterpret_cast<const double2 *>(p + idx); };

    // coefficients of the thread's pairs as NAMED variables (an array ends up in scratch memory)
    PairCoef cf0, cf1, cf2, cf3, cf4;
    bool lv0, lv1, lv2, lv3, lv4;                             // the pair lies in the domain (or is a periodic image of one that does)
    static_assert(NK <= 5 && TX * TY / 4 <= NT, "tile too large for the workgroup");
#define TILE_EACH(F) { F(0, cf0, lv0); if constexpr (NK > 1) F(1, cf1, lv1); if constexpr (NK > 2) F(2, cf2, lv2); \
                       if constexpr (NK > 3) F(3, cf3, lv3); if constexpr (NK > 4) F(4, cf4, lv4); }
    bool special = false;                                     // a pair of this thread has a cell without ice or in a cut-off range of the gap height
    // most tiles of a large level: the whole region lies in the level's own cells (uniform) -- the index logic of the general case
    // (validity of every pair, periodic images, stored ghosts, halo rows) is 250 instructions per pair, 19 % of the launch at 2048^2
    const bool inner = gx0 >= 0 && gx0 + LX <= v.nx && gy0 >= 0 && gy0 + LY <= v.ny;
    // the coefficients of a pair in the domain (and the FAS correction added to its phi, PROLONGNL)
    auto ldcoef = [&](PairCoef &c, double2 &p2, const int idx, const int i, const int j) {
        if (g.pc) {
            const int ic = ((j >> 1) + g.gyc) * g.Pc + SUHMO_XOFF + (i >> 1);
            const double corr = 1.0 * g.pc[ic] + (-1.0) * g.pco[ic];      // axby(1, -1), then PROLONGNL
            p2.x = p2.x + corr; p2.y = p2.y + corr;
        }
        double2 d;
        d = ld2(g.frhs ? fp.f[SUHMO_F_RES] : fp.f[SUHMO_F_RHS], idx); c.rhs0 = d.x; c.rhs1 = d.y;
        d = ld2(fp.f[SUHMO_F_B], idx); c.B0 = d.x; c.B1 = d.y;
        d = ld2(fp.f[SUHMO_F_PI], idx); c.Pi0 = d.x; c.Pi1 = d.y;
        d = ld2(fp.f[SUHMO_F_ZB], idx); c.zb0 = d.x; c.zb1 = d.y;
        d = ld2(fp.f[SUHMO_F_MASK], idx); c.mneg = (d.x < 0.0 ? 1 : 0) | (d.y < 0.0 ? 2 : 0);
        if (HAS_ALPHA) { d = ld2(fp.f[SUHMO_F_ACOEF], idx); c.a0 = d.x; c.a1 = d.y; }
        d = ld2(fp.f[SUHMO_F_BY], idx); c.byS0 = d.x; c.byS1 = d.y;
        d = ld2(fp.f[SUHMO_F_BY], idx + v.P); c.byN0 = d.x; c.byN1 = d.y;
        d = ld2(fp.f[SUHMO_F_BX], idx); c.bx0 = d.x; c.bx1 = d.y; c.bx2 = fp.f[SUHMO_F_BX][idx + 2];
        special = special || c.mneg != 0 || ph.cutOffbr > c.B0 || ph.cutOffbr > c.B1 || ph.maxOffbr < c.B0 || ph.maxOffbr < c.B1;
    };
    auto load = [&](const int k, PairCoef &c, bool &live) {
        live = false;
        int ly, lx;
        if (pair_pos(k, ly, lx)) {
            int i = gx0 + lx, j = gy0 + ly;
            if (inner) {                                      // the region lies in the level's own cells: no test, no wrap, no ghost
                live = true;
                const int idx = cidx(v, i, j);
                double2 p2 = ld2(pin, idx);
                ldcoef(c, p2, idx, i, j);
                lds[ly * LX + lx] = p2.x; lds[ly * LX + lx + 1] = p2.y;
                return;
            }
            // (rank boundary of a strip: the rows beyond are the neighbour's cells, held in the canvas' halo rows with their
            //  coefficients; they are advanced redundantly like periodic images, without the wrap)
            // (only the halo rows the written rows depend on: the canvas ends at gy)
            const bool inx = perx || (i >= 0 && i < v.nx), iny = pery || (j >= 0 && j < v.ny) || (j < 0 && v.rk[0] && j >= g.jbeg - HY) || (j >= v.ny && v.rk[1]
                && j < g.jend + HY);
            // AMR patch: the ghost column / row beyond a coarse-fine side holds interpolated data: loaded, never advanced
            const bool exi = inx || (i == -2 && v.cfx[0]) || (i == v.nx && v.cfx[1]);
            const bool exj = iny || (j == -1 && v.ext[0] && !v.rk[0]) || (j == v.ny && v.ext[1] && !v.rk[1]);
            double2 p2 = make_double2(0.0, 0.0);
            if (exi && exj) {
                live = inx && iny;
                if (perx) i = wrap(i, v.nx);
                if (pery) j = wrap(j, v.ny);
                const int idx = cidx(v, i, j);
                p2 = ld2(pin, idx);
            }
            if (live) ldcoef(c, p2, cidx(v, i, j), i, j);
            lds[ly * LX + lx] = p2.x; lds[ly * LX + lx + 1] = p2.y;
        }
    };
    TILE_EACH(load);
    // Straight-line passes (below) when nothing in the region needs a case distinction: no physical-boundary, coarse-fine or
    // stored-ghost side in reach, every cell under ice and outside the cut-off ranges of COMPUTENONLINEARTERMS (workgroup-uniform)
    // (only in the instantiation that relaxes the large depths: with aCoef, the restriction or the chunk loop on top the second loop
    //  body costs the kernel its second wave per SIMD)
    constexpr bool PLAIN_OK = !HAS_ALPHA && !RST && !CHUNKED;
    const bool plain = PLAIN_OK && !__syncthreads_or(special ? 1 : 0) && ph.use_NL && !openW && !openE && !openS && !openN
                       && !v.cfx[0] && !v.cfx[1] && !(v.ext[0] && !v.rk[0]) && !(v.ext[1] && !v.rk[1]);
    if constexpr (!PLAIN_OK) __syncthreads();

    // The FAS right-hand side of a coarse depth, rhs = res + L(R phi) (applyOpMg + axby of the cycle, k_apply<., 2>), formed
    // from the restricted phi just loaded, for every cell the passes will advance; the tile's own cells also store it, L(phi)
    // and the copy of R phi the prolongation subtracts later.
    if (g.frhs) {
        const int ylo = openS ? 0 : 1, yhi = openN ? LY - 1 : LY - 2;
        auto mkrhs = [&](const int k, PairCoef &q_, const bool live) {
            int ly, lx;
            if (pair_pos(k, ly, lx) && live) {
                const int i = gx0 + lx, j = gy0 + ly;
                if (ly >= ylo && ly <= yhi) {
                    const double *row = lds + ly * LX;
                    double lo[2], cc[2];
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const int x = lx + a, ii = i + a;
                        double c = row[x];
                        double w = row[x > 0 ? x - 1 : 0], e = row[x < LX - 1 ? x + 1 : LX - 1];
                        double s_ = lds[(ly > 0 ? ly - 1 : 0) * LX + x], n = lds[(ly < LY - 1 ? ly + 1 : LY - 1) * LX + x];
                        if (!perx) {
                            if (ii == 0 && !v.cfx[0]) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                            if (ii == v.nx - 1 && !v.cfx[1]) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
                        }
                        if (!pery) {
                            if (j == 0 && !v.ext[0]) s_ = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                            if (j == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
                        }
                        double nl, dnl;
                        nl_terms(ph, c, a ? q_.B1 : q_.B0, a ? q_.Pi1 : q_.Pi0, a ? q_.zb1 : q_.zb0, (q_.mneg >> a) & 1 ? -1.0 : 1.0, nl, dnl);
                        const double bxW = a ? q_.bx1 : q_.bx0, bxE = a ? q_.bx2 : q_.bx1;
                        double aterm = HAS_ALPHA ? v.alpha * (a ? q_.a1 : q_.a0) : v.alpha;
                        lo[a] = lofphi_cell(v, aterm, c, e, w, n, s_, bxE, bxW, a ? q_.byN1 : q_.byN0, a ? q_.byS1 : q_.byS0, nl);
                        cc[a] = c;
                    }
                    // (a cell on the region's edge, which no pass advances, gets a value nobody reads)
                    q_.rhs0 = 1.0 * q_.rhs0 + 1.0 * lo[0]; q_.rhs1 = 1.0 * q_.rhs1 + 1.0 * lo[1];
                    if (lx >= HX && lx < HX + TX && ly >= HY && ly < HY + TY && i < v.nx && j < g.jend) {
                        const int idx = cidx(v, i, j);
                        *reinterpret_cast<double2 *>(fp.f[SUHMO_F_LPHI] + idx) = make_double2(lo[0], lo[1]);
                        *reinterpret_cast<double2 *>(fp.f[SUHMO_F_RHS] + idx) = make_double2(q_.rhs0, q_.rhs1);
                        *reinterpret_cast<double2 *>(fp.f[SUHMO_F_PHIOLD] + idx) = make_double2(cc[0], cc[1]);
                    }
                }
            }
        };
        TILE_EACH(mkrhs);
        // (the passes below only write phi in LDS, which mkrhs only read: but a pass may overwrite a cell a slower wave still has
        // to read for its L(phi))
        __syncthreads();
    }

    // CHUNKED (its own instantiation: the outer loop costs the large tiles their second workgroup per CU): a level that is one
    // tile can go on after 2S passes, what the halo lacks then are only the periodic images of its own cells
#pragma unroll 1
    for (int chunk = 0; chunk < (CHUNKED ? g.chunks : 1); chunk++) {
    if (CHUNKED && chunk > 0 && (perx || pery)) {
        auto refresh = [&](const int k, const PairCoef &, const bool live) {
            int ly, lx;
            if (pair_pos(k, ly, lx) && live) {
                const int i = gx0 + lx, j = gy0 + ly;
                if (i < 0 || i >= v.nx || j < 0 || j >= v.ny) {
                    const int src = ((pery ? wrap(j, v.ny) : j) - gy0) * LX + ((perx ? wrap(i, v.nx) : i) - gx0);
                    const double a0 = lds[src], a1 = lds[src + 1];      // cells of the domain: nobody writes them here
                    lds[ly * LX + lx] = a0; lds[ly * LX + lx + 1] = a1;
                }
            }
        };
        TILE_EACH(refresh);
        __syncthreads();
    }
    if constexpr (PLAIN_OK) { if (plain) {
        // ---- the passes as straight-line code: every pair of the thread is evaluated (edge cells and idle slots too: their results
        // land in the margin), all LDS reads of a pass come before its writes (a pass reads the other colour and its own
        // centres only), nothing branches per lane: five independent updates the scheduler can interleave.  The expressions are
        // those of the general pass with the case distinctions that cannot occur here taken out: the same bits.
#pragma unroll 1
        for (int p = 0; p < 2 * S; p++) {
            const int lo = p + 1, xhi = LX - 2 - p, yhi = LY - 2 - p;
            const int a_pass = (gy0 + par + v.j0 + p) & 1;    // scalar: the pairs of this wave share their rows' parity
            // (the coefficients are loop-invariant: keep the compiler from hoisting products of them out of the pass loop into
            //  registers that do not exist)
#define TOUCH(c) asm volatile("" : "+v"(c.rhs0), "+v"(c.rhs1), "+v"(c.B0), "+v"(c.B1), "+v"(c.Pi0), "+v"(c.Pi1), "+v"(c.zb0), "+v"(c.zb1), \
                              "+v"(c.byS0), "+v"(c.byS1), "+v"(c.byN0), "+v"(c.byN1), "+v"(c.bx0), "+v"(c.bx1), "+v"(c.bx2))
            TOUCH(cf0); if constexpr (NK > 1) TOUCH(cf1); if constexpr (NK > 2) TOUCH(cf2); if constexpr (NK > 3) TOUCH(cf3); if constexpr (NK > 4) TOUCH(cf4);
#undef TOUCH
            double nv0 = 0.0, nv1 = 0.0, nv2 = 0.0, nv3 = 0.0, nv4 = 0.0;
            int ad0 = -1, ad1 = -1, ad2 = -1, ad3 = -1, ad4 = -1;
            auto upd = [&](const int k, const PairCoef &q_, auto a_tag, double &nv, int &ad) {
                constexpr int A = decltype(a_tag)::value;
                int ly, lx;
                const bool ok = pair_pos(k, ly, lx);
                if (!ok) { ly = par; lx = 0; }                // idle slot: any valid pair for the reads, a cell of the margin for the write
                const int x = lx + A;
                const double *row = lds + ly * LX;
                double2 pr; double wl, er, s, n;
                pr = *reinterpret_cast<const double2 *>(row + lx); wl = row[lx - 1]; er = row[lx + 2]; s = row[x - LX]; n = row[x + LX];
                const double c = A ? pr.y : pr.x;
                const double w = A ? pr.x : wl, e = A ? er : pr.y;
                const double B = A ? q_.B1 : q_.B0;
                const double N = (A ? q_.Pi1 : q_.Pi0) - ph.rho_w_g * (c - (A ? q_.zb1 : q_.zb0));      // nl_terms, no case applies
                const double nl = -ph.A * B * N * N * N;
                const double dnl = 3.0 * ph.A * B * 1000.0 * ph.grav * N * N;
                const double bxW = A ? q_.bx1 : q_.bx0, bxE = A ? q_.bx2 : q_.bx1;
                const double byN = A ? q_.byN1 : q_.byN0, byS = A ? q_.byS1 : q_.byS0;
                const double aterm = HAS_ALPHA ? v.alpha * (A ? q_.a1 : q_.a0) : v.alpha;
                const double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, byN, byS, nl);
                const double lam = lambda_cell(v, aterm, bxE, bxW, byN, byS);
                const double denom = 1.0e-16 + lam + dnl;
                const double cnew = c + ((A ? q_.rhs1 : q_.rhs0) - lofphi) / denom;
                // (the result of a cell outside this pass's region goes to a cell of the margin: an unconditional store, so that the
                //  compiler cannot sink the evaluation into a branch per pair, which would serialise the five updates again)
                const bool in = ok && x >= lo && x <= xhi && ly >= lo && ly <= yhi;
                nv = cnew;
                ad = in ? ly * LX + x : -1;
            };
            if (a_pass) {
                using A1 = std::integral_constant<int, 1>;
                upd(0, cf0, A1(), nv0, ad0); if constexpr (NK > 1) upd(1, cf1, A1(), nv1, ad1); if constexpr (NK > 2) upd(2, cf2, A1(), nv2, ad2);
                if constexpr (NK > 3) upd(3, cf3, A1(), nv3, ad3); if constexpr (NK > 4) upd(4, cf4, A1(), nv4, ad4);
            } else {
                using A0 = std::integral_constant<int, 0>;
                upd(0, cf0, A0(), nv0, ad0); if constexpr (NK > 1) upd(1, cf1, A0(), nv1, ad1); if constexpr (NK > 2) upd(2, cf2, A0(), nv2, ad2);
                if constexpr (NK > 3) upd(3, cf3, A0(), nv3, ad3); if constexpr (NK > 4) upd(4, cf4, A0(), nv4, ad4);
            }
            lds[ad0] = nv0; if constexpr (NK > 1) lds[ad1] = nv1; if constexpr (NK > 2) lds[ad2] = nv2;
            if constexpr (NK > 3) lds[ad3] = nv3; if constexpr (NK > 4) lds[ad4] = nv4;
            __syncthreads();
        }
    } }
    if (!(PLAIN_OK && plain))
#pragma unroll 1
    for (int p = 0; p < 2 * S; p++) {
        const int xlo = openW ? 0 : p + 1, xhi = openE ? LX - 1 : LX - 2 - p;
        const int ylo = openS ? 0 : p + 1, yhi = openN ? LY - 1 : LY - 2 - p;
        auto relax = [&](const int k, const PairCoef &q_, const bool live) {
            int ly, lx;
            if (pair_pos(k, ly, lx) && live) {
                const int j = gy0 + ly;
                const int a = (j + v.j0 + p) & 1;             // which cell of the pair has this pass's colour
                const int x = lx + a, i = gx0 + x;
                if (x >= xlo && x <= xhi && ly >= ylo && ly <= yhi) {
                    const double *row = lds + ly * LX;
                    double c = row[x];
                    double w = row[x > 0 ? x - 1 : 0], e = row[x < LX - 1 ? x + 1 : LX - 1];
                    double s = lds[(ly > 0 ? ly - 1 : 0) * LX + x], n = lds[(ly < LY - 1 ? ly + 1 : LY - 1) * LX + x];
                    if (!perx) {                          // mixBCValues on the fly
                        if (i == 0 && !v.cfx[0]) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                        if (i == v.nx - 1 && !v.cfx[1]) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
                    }
                    if (!pery) {
                        if (j == 0 && !v.ext[0]) s = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                        if (j == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
                    }
                    double nl, dnl;
                    nl_terms(ph, c, a ? q_.B1 : q_.B0, a ? q_.Pi1 : q_.Pi0, a ? q_.zb1 : q_.zb0, (q_.mneg >> a) & 1 ? -1.0 : 1.0, nl, dnl);
                    const double bxW = a ? q_.bx1 : q_.bx0, bxE = a ? q_.bx2 : q_.bx1;
                    const double byN = a ? q_.byN1 : q_.byN0, byS = a ? q_.byS1 : q_.byS0;
                    double aterm = HAS_ALPHA ? v.alpha * (a ? q_.a1 : q_.a0) : v.alpha;
                    double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, byN, byS, nl);
                    double lam = lambda_cell(v, aterm, bxE, bxW, byN, byS);
                    double denom = 1.0e-16 + lam + dnl;
                    lds[ly * LX + x] = c + ((a ? q_.rhs1 : q_.rhs0) - lofphi) / denom;
                }
            }
        };
        TILE_EACH(relax);
        __syncthreads();
    }
    }

    const int oi1 = (tx * TX + TX < v.nx) ? tx * TX + TX : v.nx, oj1 = (tj0 + TY < g.jend) ? tj0 + TY : g.jend;
    const int wi0 = tx * TX - ((tx == 0 && v.cfx[0]) ? 1 : 0), wi1 = oi1 + ((oi1 == v.nx && v.cfx[1]) ? 1 : 0);
    const int wj0 = tj0 - ((ty == 0 && v.ext[0] && !v.rk[0]) ? 1 : 0), wj1 = oj1 + ((oj1 == v.ny && v.ext[1] && !v.rk[1]) ? 1 : 0);
    auto store = [&](const int k, const PairCoef &q_, const bool) {
        int ly, lx;
        if (pair_pos(k, ly, lx)) {
            const int i = gx0 + lx, j = gy0 + ly;
            if (inner && !RST) {                              // (the tile itself lies in the level too)
                if (lx >= HX && lx < HX + TX && ly >= HY && ly < HY + TY)
                    *reinterpret_cast<double2 *>(pout + cidx(v, i, j)) = make_double2(lds[ly * LX + lx], lds[ly * LX + lx + 1]);
                return;
            }
            // cells this tile writes: its own and, next to a coarse-fine side, the stored ghosts (the other canvas has to
            // carry them too)
            if (j >= wj0 && j < wj1) {
                const bool m0 = i >= wi0 && i < wi1, m1 = i + 1 >= wi0 && i + 1 < wi1;
                if (m0 && m1) *reinterpret_cast<double2 *>(pout + cidx(v, i, j)) = make_double2(lds[ly * LX + lx], lds[ly * LX + lx + 1]);
                else if (m0) pout[cidx(v, i, j)] = lds[ly * LX + lx];
                else if (m1) pout[cidx(v, i + 1, j)] = lds[ly * LX + lx + 1];
            }
            if (lx >= HX && lx < HX + TX && ly >= HY && ly < HY + TY && i < v.nx && j >= 0 && j < v.ny) {   // (halo rows: advanced, not restricted)
                if constexpr (RST) {
                    const double *row = lds + ly * LX;
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const int x = lx + a, ii = i + a;
                        double c = row[x], w = row[x - 1], e = row[x + 1];          // the tile never touches the edge of the LDS region
                        double s_ = lds[(ly - 1) * LX + x], n = lds[(ly + 1) * LX + x];
                        if (!perx) {
                            if (ii == 0 && !v.cfx[0]) w = (v.bct[0][0] == 0) ? v.two_v[0][0] - c : c + v.neu[0][0];
                            if (ii == v.nx - 1 && !v.cfx[1]) e = (v.bct[0][1] == 0) ? v.two_v[0][1] - c : c + v.neu[0][1];
                        }
                        if (!pery) {
                            if (j == 0 && !v.ext[0]) s_ = (v.bct[1][0] == 0) ? v.two_v[1][0] - c : c + v.neu[1][0];
                            if (j == v.ny - 1 && !v.ext[1]) n = (v.bct[1][1] == 0) ? v.two_v[1][1] - c : c + v.neu[1][1];
                        }
                        double nl, dnl;
                        nl_terms(ph, c, a ? q_.B1 : q_.B0, a ? q_.Pi1 : q_.Pi0, a ? q_.zb1 : q_.zb0, (q_.mneg >> a) & 1 ? -1.0 : 1.0, nl, dnl);
                        const double bxW = a ? q_.bx1 : q_.bx0, bxE = a ? q_.bx2 : q_.bx1;
                        double aterm = HAS_ALPHA ? v.alpha * (a ? q_.a1 : q_.a0) : v.alpha;
                        double lofphi = lofphi_cell(v, aterm, c, e, w, n, s_, bxE, bxW, a ? q_.byN1 : q_.byN0, a ? q_.byS1 : q_.byS0, nl);
                        lq[(ly - HY) * TX + (lx - HX) + a] = ((a ? q_.rhs1 : q_.rhs0) - lofphi) / 4.0;
                    }
                }
            }
        }
    };
    TILE_EACH(store);
#undef TILE_EACH
    if constexpr (RST) {
        __syncthreads();
        if (t < TX * TY / 4) {
            const int cx = t % (TX / 2), cy = t / (TX / 2);
            const int i = tx * TX + 2 * cx, j = tj0 + 2 * cy;
            if (i < v.nx && j >= 0 && j < v.ny) {
                const double *q0 = lq + (2 * cy) * TX + 2 * cx, *p0 = lds + (HY + 2 * cy) * LX + HX + 2 * cx;
                double acc = 0.0, accp = 0.0;
                acc = acc + q0[0]; accp = accp + p0[0] / 4.0;
                acc = acc + q0[1]; accp = accp + p0[1] / 4.0;
                acc = acc + q0[TX]; accp = accp + p0[LX] / 4.0;
                acc = acc + q0[TX + 1]; accp = accp + p0[LX + 1] / 4.0;
                const int ic = ((j >> 1) + g.rgy) * g.rP + SUHMO_XOFF + (i >> 1);
                g.rres[ic] = acc; g.rphi[ic] = accp;
            }
        }
    }
}

static bool tile_ok(const suhmo_level *L, const Depth &D)
{
    const DV &v = D.v;
    if (!L->gsrb_tile || (v.nx & 1)) return false;
    if ((v.rk[0] || v.rk[1]) && (!L->ex || !L->tile_strips || L->desc.nx_global > 0 || v.gy < 10 || v.ny < 10)) return false;   // rank strips: 2S + 1
                                                              // valid halo rows per launch (AMR patch strips: colour passes)
    if (v.per[1] && (v.ny & 1)) return false;                 // colour of a periodic image = colour of the cell
    if ((v.per[0] && v.cfx[0] != v.cfx[1]) || (v.per[1] && v.ext[0] != v.ext[1])) return false;   // patch on one side of a periodic domain
    return true;
}

template <int S, int T, bool RST = false>
static int launch_tile(suhmo_level *L, int depth, int chunks, int ext_rows, hipStream_t st)
{
    Depth &D = L->d[depth];
    const DV &v = D.v;
    if (!D.phi_alt) {
        HIPCHK(hipMalloc(&D.phi_alt, D.elems * sizeof(double)));
        HIPCHK(hipMemsetAsync(D.phi_alt, 0, D.elems * sizeof(double), st));
    }
    TileGeom g;
    constexpr int TX = TileShape<T, RST>::TX, TY = TileShape<T, RST>::TY;
    g.jbeg = v.rk[0] ? -ext_rows : 0; g.jend = v.ny + (v.rk[1] ? ext_rows : 0);
    g.ntx = (v.nx + TX - 1) / TX; g.nty = (g.jend - g.jbeg + TY - 1) / TY;
    g.pc = g.pco = nullptr; g.Pc = g.gyc = 0;
    g.rres = g.rphi = nullptr; g.rP = g.rgy = 0;
    if (RST) {
        Depth &C = L->d[depth + 1];
        g.rres = C.fp.f[SUHMO_F_RES]; g.rphi = C.fp.f[SUHMO_F_PHI]; g.rP = C.v.P; g.rgy = C.v.gy;
        C.phi_fresh = 0;
    }
    if (D.prolong_pending) {
        Depth &C = L->d[depth + 1];
        g.pc = C.fp.f[SUHMO_F_PHI]; g.pco = C.fp.f[SUHMO_F_PHIOLD]; g.Pc = C.v.P; g.gyc = C.v.gy;
        D.prolong_pending = 0;
    }
    g.chunks = chunks;
    g.order = L->tile_order;
    g.frhs = 0;
    if (D.rhs_pending) {
        if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_PHIOLD)) return -2;
        g.frhs = 1; D.rhs_pending = 0; L->frhs_tile++;
    }
    const double *pin = D.fp.f[SUHMO_F_PHI];
    if constexpr (S == 4) {
        if (chunks > 1) {
            if (v.alpha != 0.0)
                hipLaunchKernelGGL((k_gsrb_tile<S, T, true, RST, true>), dim3(g.ntx * g.nty), dim3(TileThreads<S, T, RST>::NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
            else
                hipLaunchKernelGGL((k_gsrb_tile<S, T, false, RST, true>), dim3(g.ntx * g.nty), dim3(TileThreads<S, T, RST>::NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
            { std::swap(D.fp.f[SUHMO_F_PHI], D.phi_alt); suhmo_fp_changed(); }
            return 0;
        }
    }
    if (chunks > 1) { suhmo_set_error("internal: chunked tile launch with S != 4"); return -4; }
    if (v.alpha != 0.0)
        hipLaunchKernelGGL((k_gsrb_tile<S, T, true, RST>), dim3(g.ntx * g.nty), dim3(TileThreads<S, T, RST>::NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
    else
        hipLaunchKernelGGL((k_gsrb_tile<S, T, false, RST>), dim3(g.ntx * g.nty), dim3(TileThreads<S, T, RST>::NT), 0, st, v, D.fp, pin, D.phi_alt, L->ph, g);
    { std::swap(D.fp.f[SUHMO_F_PHI], D.phi_alt); suhmo_fp_changed(); }
    return 0;
}
// tile edge: 16 below 600 k cells (enough workgroups for the chip -- 512^2 is 256 tiles of 32, one per CU: 6.4 us per sweep against
// 5.7 on 1024 tiles of 16; also the shorter launch on tiny levels), 32 above (1024^2: 15.0 against 17.2)
static int tile_edge(const suhmo_level *L, const DV &v)
{
    if (L->tile_t) return L->tile_t;
    return ((long)v.nx * v.ny >= 600000L) ? 32 : 16;
}
// a level that is ONE tile (16-wide, or 32-wide when there are sweeps enough to pay for the larger region) can take all its
// sweeps in one launch; returns the tile edge to use or 0
static int single_tile(const suhmo_level *L, const DV &v)
{
    if (L->tile_t != 32 && v.nx <= 16 && v.ny <= 16) return 16;
    if (L->tile_t != 16 && v.nx <= 32 && v.ny <= 28) return 32;   // (the restricting variant of the 32-wide tile is 28 rows high)
    return 0;
}
static int launch_tile_any(suhmo_level *L, int depth, int S, int chunks, bool rst, int ext_rows, hipStream_t st)
{
    const int T = chunks > 1 ? single_tile(L, L->d[depth].v) : tile_edge(L, L->d[depth].v);
    if (rst) {
        if (T == 32) return S == 4 ? launch_tile<4, 32, true>(L, depth, chunks, ext_rows, st) : S == 2 ? launch_tile<2, 32, true>(L, depth, chunks, ext_rows,
            st) : launch_tile<1, 32, true>(L, depth, chunks, ext_rows, st);
        return S == 4 ? launch_tile<4, 16, true>(L, depth, chunks, ext_rows, st) : S == 2 ? launch_tile<2, 16, true>(L, depth, chunks, ext_rows, st) : launch_tile<1,
            16, true>(L, depth, chunks, ext_rows, st);
    }
    if (T == 32) return S == 4 ? launch_tile<4, 32>(L, depth, chunks, ext_rows, st) : S == 2 ? launch_tile<2, 32>(L, depth, chunks, ext_rows, st) : launch_tile<1, 32>(L, depth, chunks, ext_rows, st);
    return S == 4 ? launch_tile<4, 16>(L, depth, chunks, ext_rows, st) : S == 2 ? launch_tile<2, 16>(L, depth, chunks, ext_rows, st) : launch_tile<1, 16>(L, depth, chunks, ext_rows, st);
}

static int pick_variant(const suhmo_level *L, const Depth &D)
{
    int variant = L->gsrb_variant;       // -1 auto, 0 simple, 1 fused K=1, 2 fused K=2
    if (variant < 0) {
        // the streaming kernel pays ~(Hc + 4K) serial row steps per workgroup: below ~2M cells
        // (cache-resident depths) two plain colour-pass launches are faster
        // (profiles/r01_c_vcycle_trace.txt)
        const long cells = (long)D.v.nx * D.v.ny;
        variant = (cells >= (long)L->fused_min_cells) ? 2 : 0;
        // whole levels of up to a few million cells: 4 sweeps per launch on LDS tiles beat 2 per streaming pass
        // (profiles/r01_s_tile_vs_streaming.txt: 2048^2 47.5 vs 56.8 us per sweep, 4096^2 175 vs 146)
        if (tile_ok(L, D) && cells < L->tile_max_cells) variant = 0;
    }
    return variant;
}
static int pick_K(const suhmo_level *L, const Depth &D, int variant, int remaining)
{
    if (variant >= 2 && remaining >= 2 && fused_ok(L, D, 2)) return 2;
    if (variant >= 1 && fused_ok(L, D, 1)) return 1;
    return 0;
}
// rank strips: halo rows that are valid on the fine depth AND (halved) on the coarse one stay valid when the
// correction is added while loading
static int prolong_halo_rows(const suhmo_level *L, int depth)
{
    const Depth &D = L->d[depth], &C = L->d[depth + 1];
    int R = D.phi_fresh < 2 * C.phi_fresh ? D.phi_fresh : 2 * C.phi_fresh;
    return R & ~1;
}
// the first relaxation of a coarse FAS depth can form its right-hand side itself
bool suhmo_gsrb_can_fuse_rhs(suhmo_level *L, int depth, int sweeps, bool rhs_local)
{
    Depth &D = L->d[depth];
    if (sweeps < 1 || !L->fas_rhs_in_relax) return false;
    const bool strip = D.v.rk[0] || D.v.rk[1];                // rank strips: k_apply<., 2> also copies the halo rows of R phi
    const int K = pick_K(L, D, pick_variant(L, D), sweeps);
    if (K <= 0) return !strip && (L->fas_rhs_in_relax & 1) && tile_ok(L, D);
    // streaming kernel: the two-sweep launch of one-wave workgroups on a whole level (k_gsrb_fused<2, false, 64, false, true>);
    // with only two sweeps to do that launch is the one that restricts
    if (!(L->fas_rhs_in_relax & 2)) return false;
    const int nt = L->fused_nt ? L->fused_nt : ((long)D.v.nx * D.v.ny >= 8000000L ? 256 : 64);
    if (!(K == 2 && nt == 64 && sweeps >= 4 && D.v.alpha == 0.0)) return false;
    if (!(D.v.ext[0] || D.v.ext[1])) return true;
    // rank strip: R phi and RES have just arrived in all halo rows (the caller's rhs_local exchange); the launch then loads all of them
    return strip && L->ex && rhs_local && L->desc.nx_global == 0 && D.phi_fresh >= D.v.gy && D.v.gy >= 2 * K + 1 && D.v.ny >= D.v.gy;
}
bool suhmo_gsrb_can_fuse_prolong(suhmo_level *L, int depth, int sweeps)
{
    Depth &D = L->d[depth];
    if (sweeps < 1 || depth + 1 >= L->ndepth) return false;
    int K = pick_K(L, D, pick_variant(L, D), sweeps);
    if (K <= 0) {
        if (!tile_ok(L, D)) return false;
        if (!(D.v.rk[0] || D.v.rk[1])) return true;
        const int TS = sweeps >= 4 && L->tile_s >= 4 ? 4 : sweeps >= 2 && L->tile_s >= 2 ? 2 : 1;
        return prolong_halo_rows(L, depth) >= 2 * TS;        // else: un-fused prolongation, then an exchange
    }
    const bool ext = L->ex && (D.v.ext[0] || D.v.ext[1]);
    if ((D.v.ext[0] || D.v.ext[1]) && !ext) return false;                 // stored ghost rows without a transport (AMR patch)
    return !ext || prolong_halo_rows(L, depth) >= 2 * K;                  // else: un-fused prolongation, then an exchange
}

// restricted != NULL: the caller restricts right after these sweeps (pre-smoothing of the FAS cycle); *restricted = 1 if
// the last launch did it (coarse RES and PHI written), 0 if the caller still has to
int suhmo_launch_gsrb(suhmo_level *L, int depth, int sweeps, int tail, hipStream_t st, int *restricted)
{
    Depth &D = L->d[depth];
    int variant = pick_variant(L, D);
    const bool ext = L->ex && (D.v.ext[0] || D.v.ext[1]);
    if (restricted) *restricted = 0;
    // rank strips: the rows next to the strip's ends come from the halo (one more valid row than the sweeps need, an even
    // number of redundantly advanced rows so that a coarse cell's two fine rows share a chunk); not on AMR patches
    const bool may_restrict = restricted && L->fused_restrict && depth + 1 < L->ndepth && !(D.v.ny & 1) && !(D.v.j0 & 1)
                              && (!(D.v.ext[0] || D.v.ext[1]) || (ext && L->desc.nx_global == 0 && D.v.gy >= 5))
                              && D.v.alpha == 0.0;          // with aCoef the extra coefficient row no longer fits 2 waves per SIMD
    // Strips: F = halo rows of phi that hold current neighbour values (Depth::phi_fresh).  A colour pass
    // needs 1, a K-sweep launch 2K; each launch also advances, redundantly, as many of the remaining halo
    // rows as the work still to come (rest of these sweeps + `tail` rows for the next reader) can use, so
    // one exchange of halo_rows rows feeds up to halo_rows colour passes.
    int F = ext ? D.phi_fresh : 0;
    if (ext && D.prolong_pending) F = prolong_halo_rows(L, depth);   // the first launch adds the correction while loading
    int it = 0;
    while (it < sweeps) {
        int K = pick_K(L, D, variant, sweeps - it);   // sweeps done by the next launch (0 = simple path, 1 sweep)
        int TS = (K == 0 && tile_ok(L, D)) ? (sweeps - it >= 4 && L->tile_s >= 4 ? 4 : sweeps - it >= 2 && L->tile_s >= 2 ? 2 : 1) : 0;   // sweeps of a tile launch
        int tE = 0;                                            // rank strip: halo rows the tile launch advances redundantly
        bool trst = TS && restricted && L->fused_restrict && (L->tile_restrict == 1 || (L->tile_restrict == 2 && ext)) && depth + 1 < L->ndepth && !(D.v.ny & 1) && !(D.v.j0 & 1);
        if (TS && ext) {
            // rank strip: the tile's halo rows beyond the strip are the neighbour's cells: 2S of them must be current (one more
            // when the launch also restricts); they are advanced redundantly and stale afterwards
            trst = trst && it + TS == sweeps;
            const int need = 2 * TS + (trst ? 1 : 0);
            if (F < need && !D.prolong_pending) {
                int rc = suhmo_ensure_phi_halo(L, depth, need, st); if (rc) return rc;
                F = D.phi_fresh;
            }
            if (F < need) {
                if (D.prolong_pending) { suhmo_set_error("internal: fused prolongation with a shallow halo"); return -4; }
                TS = 0;                                        // halo shallower than the tile needs: colour passes
            } else {
                // as many more halo rows as the work still to come can use are advanced too (one exchange feeds several launches)
                const int want = 2 * (sweeps - it - TS) + tail;
                tE = F - need < want ? F - need : want;
                if (tE > D.v.gy - need - 1) tE = D.v.gy - need - 1;
                tE = tE < 0 ? 0 : tE & ~1;
            }
        }
        if (K == 0 && !TS && D.prolong_pending) { suhmo_set_error("internal: prolong_pending without a fused relax"); return -4; }
        ProfEv pe{};
        bool prof = L->prof_on && depth == 0;
        if (prof) {
            HIPCHK(hipEventCreate(&pe.a)); HIPCHK(hipEventCreate(&pe.b));
            HIPCHK(hipEventRecord(pe.a, st));
        }
        int tchunks = 1;
        if (TS) {
            if (L->tile_chunks && !ext && TS == 4 && single_tile(L, D.v)) tchunks = (sweeps - it) / TS;       // e.g. the 16 bottom sweeps in one launch (1: as before)
            const bool rst = trst && it + TS * tchunks == sweeps;
            int rc = launch_tile_any(L, depth, TS, tchunks, rst, tE, st); if (rc) return rc;
            if (rst) *restricted = 1;
            if (ext) { F = tE; D.phi_fresh = tE; }
        } else if (K == 0) {
            for (int pass = 0; pass < 2; pass++) {
                if (ext && F < 1) {
                    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
                    F = D.phi_fresh;
                }
                int want = 2 * (sweeps - it) - (pass + 1) + tail;          // halo rows the work after this pass can use
                int E = ext ? (F - 1 < want ? F - 1 : want) : 0;
                if (L->desc.nx_global > 0) E = 0;                          // AMR patch strips: the coarse-fine ghost columns of halo
                                                                           // rows are not exchanged -> no redundant advance
                launch_simple(L, depth, pass, E, st);
                if (ext) { F = E; D.phi_fresh = F; }
            }
        } else {
            const int nt = L->fused_nt ? L->fused_nt : ((long)D.v.nx * D.v.ny >= 8000000L ? 256 : 64);   // 0 = by size
            bool rst = may_restrict && nt == 64 && K == 2 && it + K == sweeps;
            // the launch that ends the cycle (armed by the cycle for its last relax of depth 0) also leaves the residual of the final phi
            // behind (same needs as the restricting launch: one more final row and column around a chunk)
            bool rout = L->resout_armed && depth == 0 && !restricted && nt == 64 && K == 2 && it + K == sweeps && D.v.alpha == 0.0
                        && (!(D.v.ext[0] || D.v.ext[1]) || ext) && L->desc.nx_global == 0 && !D.rhs_pending;
            // rank strip: 2K current halo rows, one more when the launch also restricts / evaluates the residual (one exchange instead of
            // the one the separate pass would ask for)
            const int need = 2 * K + ((rst || rout) && ext ? 1 : 0);
            bool flying = false;                                           // the exchange is in flight on the second stream
            if (ext && F < need) {
                // only the native transport is stream-ordered (its pack / send / recv / unpack are enqueued on the stream the hook is
                // given); a host transport (torch.distributed, gloo, threads) orders against other streams or the whole device:
                // overlap_halo = 1 overlaps with the native transport only, 2 = the caller vouches for its hook
                if ((L->overlap_halo == 2 || (L->overlap_halo && (L->rccl || L->ipc))) && !D.prolong_pending && D.v.ny >= 6 * need) {
                    if (!L->xstream) {
                        HIPCHK(hipStreamCreateWithFlags(&L->xstream, hipStreamNonBlocking));
                        HIPCHK(hipEventCreateWithFlags(&L->xev[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&L->xev[1], hipEventDisableTiming));
                    }
                    HIPCHK(hipEventRecord(L->xev[0], st));                 // the rows that travel are final
                    HIPCHK(hipStreamWaitEvent(L->xstream, L->xev[0], 0));
                    int rc2 = suhmo_ensure_phi_halo(L, depth, need, L->xstream); if (rc2) return rc2;
                    HIPCHK(hipEventRecord(L->xev[1], L->xstream));
                    flying = true;
                } else { int rc2 = suhmo_ensure_phi_halo(L, depth, need, st); if (rc2) return rc2; }
                F = D.phi_fresh;
                if (F < 2 * K) { suhmo_set_error("internal: halo shallower than 2K"); return -4; }
            }
            int want = 2 * (sweeps - it - K) + tail;
            if (D.rhs_pending && ext) want = F;                            // the launch that forms the right-hand side loads (and keeps) all halo rows
            int E = ext ? (F - 2 * K < want ? F - 2 * K : want) : 0;
            int rc = 0;
            if ((rst || rout) && ext) {
                if (F < 2 * K + 1) rst = rout = false;
                else { E = F - 2 * K - 1 < want ? F - 2 * K - 1 : want; if (rst) E &= ~1; }
            }
            auto launch = [&](int part) {
                if (rout) return launch_fused<2, 64, 2>(L, depth, E, st, part);
                if (rst) return launch_fused<2, 64, 1>(L, depth, E, st, part);
                if (nt == 64) return (K == 2) ? launch_fused<2, 64>(L, depth, E, st, part) : launch_fused<1, 64>(L, depth, E, st, part);
                return (K == 2) ? launch_fused<2, 256>(L, depth, E, st, part) : launch_fused<1, 256>(L, depth, E, st, part);
            };
            if (flying) {
                rc = launch(1);                                            // the chunks that read no halo row: now
                if (rc < 0) return rc;
                HIPCHK(hipStreamWaitEvent(st, L->xev[1], 0));              // the halo rows have arrived
                if (rc == 1) rc = launch(0); else { rc = launch(2); L->overlapped++; }
            } else rc = launch(0);
            if (rc) return rc;
            if (rst) *restricted = 1;
            if (rout) { L->resout_done = 1; L->resout_count++; }
            if (ext) { F = E; D.phi_fresh = F; }
        }
        int done = TS ? TS * tchunks : K == 0 ? 1 : K;
        if (prof) {
            HIPCHK(hipEventRecord(pe.b, st));
            pe.cells = (long)D.v.nx * D.v.ny * done;
            pe.restricts = (restricted && *restricted && it + done == sweeps) ? 1 : 0;
            L->prof.push_back(pe);
        }
        it += done;
    }
    HIPCHK(hipGetLastError());
    return 0;
}
