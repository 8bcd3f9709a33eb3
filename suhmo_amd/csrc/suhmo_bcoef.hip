// suhmo_bcoef.hip -- UpdateOperator (WFlx_level: gradient, Re, bCoef on faces; the fused single-kernel form), AverageOperator and MGnewOp's
// coefficient coarsening.  (Split off suhmo_level.hip in round 4; reference citations: file:line in the SUHMO checkout.)
#include "suhmo_hier.h"
#include "suhmo_level_int.h"
#include <algorithm>
#include <cmath>
#include <initializer_list>
// ------------------------------------------------------------------ bCoef update (WFlx_level)
// step 1: cell-centred gradient = EdgeToCell(NEWMACGRAD) (util/Gradient.cpp:96-127, :623;
// util/GradientF.ChF:57-70)
__device__ __forceinline__ void d_gradcc_val(const DV &v, const FP &fp, int hasMask, int i, int j, double &gx, double &gy)
{
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    double c = phi[idx];
    double e = phiE(v, phi, idx, i, c, false), w = phiW(v, phi, idx, i, c, false);
    double n = phiN(v, phi, idx, j, c, false), s = phiS(v, phi, idx, j, c, false);
    double gW = v.fdx * (c - w), gE = v.fdx * (e - c), gS = v.fdy * (c - s), gN = v.fdy * (n - c);
    if (hasMask) {
        const double *__restrict__ m = fp.f[SUHMO_F_MASK];
        bool mc = m[idx] < 1e-6;
        if (mc || m[idx - 1] < 1e-6) gW = 0.0;
        if (mc || m[idx + 1] < 1e-6) gE = 0.0;
        if (mc || m[idx - v.P] < 1e-6) gS = 0.0;
        if (mc || m[idx + v.P] < 1e-6) gN = 0.0;
    }
    gx = 0.5 * (gW + gE);
    gy = 0.5 * (gS + gN);
}
__device__ __forceinline__ void d_gradcc_at(const DV &v, const FP &fp, int hasMask, int i, int j)
{
    if (i >= v.nx || j >= v.ny) return;
    double gx, gy;
    d_gradcc_val(v, fp, hasMask, i, j, gx, gy);
    const int idx = cidx(v, i, j);
    fp.f[SUHMO_F_GRADX][idx] = gx;
    fp.f[SUHMO_F_GRADY][idx] = gy;
}
__device__ __forceinline__ void d_gradcc(const DV &v, const FP &fp, int hasMask)
{
    d_gradcc_at(v, fp, hasMask, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y);
}
__global__ __launch_bounds__(256) void k_gradcc(DV v, FP fp, int hasMask)
{
    d_gradcc(v, fp, hasMask);
}
// the same at a list of cells (x = i, y = j): the coarse cells a finer level's coarse-fine interpolation of the gradient reads
__global__ __launch_bounds__(256) void k_gradcc_list(DV v, FP fp, int hasMask, const int2 *__restrict__ cells, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d_gradcc_at(v, fp, hasMask, cells[t].x, cells[t].y);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_gradcc_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int hasMask)
{
    d_gradcc(vt[blockIdx.z], ft[blockIdx.z], hasMask);
}
// step 2: ghosts of the gradient: exchange (periodic wrap) + ExtrapGhostCells
// (src/AmrHydro.cpp:1490-1491, util/ExtrapGhostCells.cpp:94-180, util/ExtrapBCF.ChF:21-29)
__device__ __forceinline__ void d_grad_ghosts(const DV &v, double *__restrict__ gx, double *__restrict__ gy)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    double *g2[2] = {gx, gy};
    if (t < 2 * v.ny) {
        int side = t / v.ny, j = t % v.ny;
        if (v.cfx[side]) return;
        for (int c = 0; c < 2; c++) {
            double *g = g2[c];
            if (side == 0) { int idx = cidx(v, 0, j); g[idx - 1] = v.per[0] ? g[idx + v.nx - 1] : 2.0 * g[idx] - g[idx + 1]; }
            else { int idx = cidx(v, v.nx - 1, j); g[idx + 1] = v.per[0] ? g[idx - (v.nx - 1)] : 2.0 * g[idx] - g[idx - 1]; }
        }
        return;
    }
    t -= 2 * v.ny;
    if (t < 2 * v.nx) {
        int side = t / v.nx, i = t % v.nx;
        if (v.ext[side]) return;
        for (int c = 0; c < 2; c++) {
            double *g = g2[c];
            if (side == 0) { int idx = cidx(v, i, 0); g[idx - v.P] = v.per[1] ? g[idx + (v.ny - 1) * v.P] : 2.0 * g[idx] - g[idx + v.P]; }
            else { int idx = cidx(v, i, v.ny - 1); g[idx + v.P] = v.per[1] ? g[idx - (v.ny - 1) * v.P] : 2.0 * g[idx] - g[idx - v.P]; }
        }
    }
}
__global__ void k_grad_ghosts(DV v, double *__restrict__ gx, double *__restrict__ gy)
{
    d_grad_ghosts(v, gx, gy);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_grad_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    d_grad_ghosts(vt[blockIdx.z], ft[blockIdx.z].f[SUHMO_F_GRADX], ft[blockIdx.z].f[SUHMO_F_GRADY]);
}
// steps 1 + 2 of every box of a level in ONE launch: the thread of a cell on a physical side of its box also writes the ghost cell beyond it,
// from the gradient of the neighbour the extrapolation (or the periodic wrap) reads, evaluated a second time: d_grad_ghosts' expressions on the
// same values (boxes of AMR levels are at least two cells wide: block_factor 2)
__device__ __forceinline__ void d_gradcc_ghosts(const DV &v, const FP &fp, int hasMask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    double gx, gy, ox, oy;
    d_gradcc_val(v, fp, hasMask, i, j, gx, gy);
    const int idx = cidx(v, i, j);
    double *__restrict__ GX = fp.f[SUHMO_F_GRADX], *__restrict__ GY = fp.f[SUHMO_F_GRADY];
    GX[idx] = gx; GY[idx] = gy;
    if (i == 0 && !v.cfx[0]) {
        d_gradcc_val(v, fp, hasMask, v.per[0] ? v.nx - 1 : 1, j, ox, oy);
        GX[idx - 1] = v.per[0] ? ox : 2.0 * gx - ox; GY[idx - 1] = v.per[0] ? oy : 2.0 * gy - oy;
    }
    if (i == v.nx - 1 && !v.cfx[1]) {
        d_gradcc_val(v, fp, hasMask, v.per[0] ? 0 : v.nx - 2, j, ox, oy);
        GX[idx + 1] = v.per[0] ? ox : 2.0 * gx - ox; GY[idx + 1] = v.per[0] ? oy : 2.0 * gy - oy;
    }
    if (j == 0 && !v.ext[0]) {
        d_gradcc_val(v, fp, hasMask, i, v.per[1] ? v.ny - 1 : 1, ox, oy);
        GX[idx - v.P] = v.per[1] ? ox : 2.0 * gx - ox; GY[idx - v.P] = v.per[1] ? oy : 2.0 * gy - oy;
    }
    if (j == v.ny - 1 && !v.ext[1]) {
        d_gradcc_val(v, fp, hasMask, i, v.per[1] ? 0 : v.ny - 2, ox, oy);
        GX[idx + v.P] = v.per[1] ? ox : 2.0 * gx - ox; GY[idx + v.P] = v.per[1] ? oy : 2.0 * gy - oy;
    }
}
__global__ __launch_bounds__(256) void k_gradcc_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int hasMask)
{
    d_gradcc_ghosts(vt[blockIdx.z], ft[blockIdx.z], hasMask);
}
// ... of SEVERAL levels (UpdateOperator of an AMR level evaluates the gradient of its own and of the coarser level)
__global__ __launch_bounds__(256) void k_gradcc_ghosts_lv(suhmo_lvboxes lv, int hasMask)
{
    int z = blockIdx.z, q = -1;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t < lv.n && q < 0) { if (z < lv.nbox[t]) q = t; else z -= lv.nbox[t]; }
    if (q < 0) return;
    const DV *dv = nullptr; const FP *fp = nullptr;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++) if (t == q) { dv = lv.dv[t]; fp = lv.fp[t]; }
    d_gradcc_ghosts(dv[z], fp[z], hasMask);
}
// step 3: COMPUTERE on the ghosted box (src/AmrHydro.cpp:1495-1505, src/AmrHydroF.ChF:92-109)
__device__ __forceinline__ double d_re_val(const FP &fp, const suhmo_phys_t &ph, int idx)
{
    double gx = fp.f[SUHMO_F_GRADX][idx], gy = fp.f[SUHMO_F_GRADY][idx], B = fp.f[SUHMO_F_B][idx];
    double sg = sqrt(gx * gx + gy * gy);
    double discr = 1.0 + 4.0 * ph.omega * (B * B * B * ph.grav * sg) / (12.0 * ph.nu * ph.nu);
    return (-1.0 + sqrt(discr)) / (2.0 * ph.omega);
}
__device__ __forceinline__ void d_re(const DV &v, const FP &fp, suhmo_phys_t ph)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    bool xo = (i < 0 || i >= v.nx), yo = (j < 0 || j >= v.ny);
    if (xo && yo) return;                                      // corner ghosts are never read
    int idx = cidx(v, i, j);
    fp.f[SUHMO_F_RE][idx] = d_re_val(fp, ph, idx);
}
__global__ __launch_bounds__(256) void k_re(DV v, FP fp, suhmo_phys_t ph)
{
    d_re(v, fp, ph);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_re_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph)
{
    d_re(vt[blockIdx.z], ft[blockIdx.z], ph);
}
// step 4: CellToEdge(Re), CellToEdge(B), setup_iceMask_EC, COMPUTEBCOEFF
// (src/AmrHydro.cpp:1512-1537, src/HydroIBC.cpp:139-184, src/AmrHydroF.ChF:212-228)
__device__ __forceinline__ double bcoef_face(const suhmo_phys_t &ph, double Rc, double Rm, double Bc, double Bm,
                                             double mc, double mm, bool dom_edge)
{
    double Ref = 0.5 * (Rc + Rm), Bf = 0.5 * (Bc + Bm);
    double mec;
    if (fabs(mc - mm) < 1e-10) mec = (mc > 0.0) ? 1.0 : -1.0; else mec = 0.0;
    if (dom_edge) mec = 0.0;
    double num_q = -(Bf * Bf * Bf * ph.grav);
    double denom_q = 12.0 * ph.nu * (1.0 + ph.omega * Ref);
    if (mec < 0.0 && ph.cutOffB > 0) return 0.0;
    return num_q / denom_q;
}
__device__ __forceinline__ void d_bcoef_faces(const DV &v, const FP &fp, suhmo_phys_t ph)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    const double *__restrict__ Re = fp.f[SUHMO_F_RE], *__restrict__ B = fp.f[SUHMO_F_B], *__restrict__ m = fp.f[SUHMO_F_MASK];
    if (j < v.ny)   // x-face (i,j) between cells (i-1,j) and (i,j)
        fp.f[SUHMO_F_BX][idx] = bcoef_face(ph, Re[idx], Re[idx - 1], B[idx], B[idx - 1], m[idx], m[idx - 1], i + v.i0 == 0 || i + v.i0 == v.nxg);
    if (i < v.nx) { // y-face (i,j) between cells (i,j-1) and (i,j)
        int jg = j + v.j0;
        fp.f[SUHMO_F_BY][idx] = bcoef_face(ph, Re[idx], Re[idx - v.P], B[idx], B[idx - v.P], m[idx], m[idx - v.P], jg == 0 || jg == v.nyg);
    }
}
__global__ __launch_bounds__(256) void k_bcoef_faces(DV v, FP fp, suhmo_phys_t ph)
{
    d_bcoef_faces(v, fp, ph);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ __launch_bounds__(256) void k_bcoef_faces_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph)
{
    d_bcoef_faces(vt[blockIdx.z], ft[blockIdx.z], ph);
}
// steps 3 + 4 of every box of a level in ONE launch: the thread of a position of the ghosted box stores Re there (d_re) and the two faces on
// its low sides (d_bcoef_faces), with the Re of the two cells across those faces evaluated a second time from the same gradient and gap height
__global__ __launch_bounds__(256) void k_re_bcoef_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph)
{
    const DV &v = vt[blockIdx.z];
    const FP &fp = ft[blockIdx.z];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    const bool xo = (i < 0 || i >= v.nx), yo = (j < 0 || j >= v.ny);
    if (xo && yo) return;                                      // corner ghosts are never read
    const int idx = cidx(v, i, j);
    const double Rc = d_re_val(fp, ph, idx);
    fp.f[SUHMO_F_RE][idx] = Rc;
    if (i < 0 || j < 0) return;                                // (faces (i, j), 0 <= i <= nx, 0 <= j <= ny)
    const double *__restrict__ B = fp.f[SUHMO_F_B], *__restrict__ m = fp.f[SUHMO_F_MASK];
    if (j < v.ny)
        fp.f[SUHMO_F_BX][idx] = bcoef_face(ph, Rc, d_re_val(fp, ph, idx - 1), B[idx], B[idx - 1], m[idx], m[idx - 1], i + v.i0 == 0 || i + v.i0 == v.nxg);
    if (i < v.nx) {
        const int jg = j + v.j0;
        fp.f[SUHMO_F_BY][idx] = bcoef_face(ph, Rc, d_re_val(fp, ph, idx - v.P), B[idx], B[idx - v.P], m[idx], m[idx - v.P], jg == 0 || jg == v.nyg);
    }
}

// ---- fused WFlx_level: one kernel = steps 1-4 above on a tile staged in LDS.
// A block of 64 x 4 threads owns BT_X x BT_Y = 62 x 14 cells (26 KB of LDS, 6 blocks per CU: 0.26 ms at 4096^2 vs
// 0.29 ms with 30 rows).  phi tile (halo 2) and
// the B / mask tiles (halo 1) are loaded up front (one exposure to HBM latency); then
// cell-centred gradient on the Re range (halo 1: one lane per column, NK rows per
// thread, in registers) -> ghost gradients by linear extrapolation (periodic images and
// exchanged halo rows are ordinary cells) -> Re (LDS, aliasing the dead phi tile) -> the
// tile's W and S faces (+ the domain's E / N faces in the last tile column / row).
// Every value comes from the same expressions as the four-kernel path (bitwise equal); halo
// cells are recomputed instead of stored, so HBM sees phi, B, mask once and bx, by once.
// Tile shapes: 62 x 14 on 64 x 4 threads, or 126 x 14 on 128 x 2 threads (BT_X + 2 lanes per row; level option bcoef_tile_x): the wider tile's
// rows are 1040 instead of 528 bytes, so the 128-byte lines its unaligned ends drag in weigh half as much.
// INTERIOR: the tile and its two-cell halo lie inside the level -- no boundary condition, no wrap, no missing cell: the
// same expressions without the case distinctions (most tiles; uniform per workgroup)
template <bool INTERIOR, int BT_X, int BT_Y>
__device__ __forceinline__ void bcoef_tile(const DV &v, const FP &fp, const suhmo_phys_t &ph, int hasMask, double *sphi, double *sB, double *sM,
                                           unsigned *negflag, unsigned epoch)
{
    constexpr int PW = BT_X + 4, PH = BT_Y + 4;     // phi tile: cells [i0-2, i0+BT_X+1] x [j0-2, j0+BT_Y+1]
    constexpr int RW = BT_X + 2, RH = BT_Y + 2;     // Re tile:  cells [i0-1, i0+BT_X]   x [j0-1, j0+BT_Y]
    constexpr int TX = BT_X + 2, TY = 256 / TX;     // threads of the workgroup: TX lanes along a row, TY rows at a time
    constexpr int NK = RH / TY;
    static_assert(TX * TY == 256 && RH % TY == 0, "tile shape");
    double *sre = sphi;                              // phi is dead once the gradients exist
    const int i0 = blockIdx.x * BT_X, j0 = blockIdx.y * BT_Y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI], *__restrict__ Bf = fp.f[SUHMO_F_B], *__restrict__ mk = fp.f[SUHMO_F_MASK];
    double *__restrict__ bxo = fp.f[SUHMO_F_BX], *__restrict__ byo = fp.f[SUHMO_F_BY];
    const bool halo_lo = v.ext[0], halo_hi = v.ext[1], selfper_y = v.per[1] && !halo_lo && !halo_hi;
    // a cell "exists" (has its own phi) inside the domain, as a periodic image, or in an exchanged halo row
    auto xin = [&](int i) { return INTERIOR || (i >= 0 && i < v.nx) || v.per[0]; };
    auto yin = [&](int j) { return INTERIOR || (j >= 0 && j < v.ny) || selfper_y || (j < 0 && halo_lo && j >= -v.gy) || (j >= v.ny && halo_hi && j < v.ny + v.gy); };
    auto wrapx = [&](int i) { return (!INTERIOR && v.per[0]) ? (i < 0 ? i + v.nx : (i >= v.nx ? i - v.nx : i)) : i; };
    auto wrapy = [&](int j) { return (!INTERIOR && selfper_y) ? (j < 0 ? j + v.ny : (j >= v.ny ? j - v.ny : j)) : j; };

    // ---- phi tile.  Cells that do not exist get the physical-BC ghost of their interior
    // neighbour (only the first ghost layer is used: face gradient of the boundary cell).
    for (int lj = ty; lj < PH; lj += TY) {
        const int j = j0 - 2 + lj;
        const bool yi = yin(j);
        for (int li = tx; li < PW; li += TX) {
            const int i = i0 - 2 + li;
            const bool xi = xin(i);
            double val = 0.0;
            if (xi && yi) val = phi[cidx(v, wrapx(i), wrapy(j))];
            else if (yi && (i == -1 || i == v.nx)) {
                int ic = i < 0 ? 0 : v.nx - 1, idx = cidx(v, ic, wrapy(j));
                double c = phi[idx];
                val = i < 0 ? phiW(v, phi, idx, ic, c, false) : phiE(v, phi, idx, ic, c, false);
            } else if (xi && (j == -1 || j == v.ny)) {
                int jc = j < 0 ? 0 : v.ny - 1, idx = cidx(v, wrapx(i), jc);
                double c = phi[idx];
                val = j < 0 ? phiS(v, phi, idx, jc, c, false) : phiN(v, phi, idx, jc, c, false);
            }
            sphi[lj * PW + li] = val;
        }
    }
    // ---- B and mask on the Re range (stored ghosts included: caller data, src/AmrHydro.cpp:686-701)
    const int i = i0 - 1 + tx;
    const bool xi = xin(i);
    double Br[NK];
    bool hasB[NK];
    bool neg = false;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int lj = ty + TY * k, j = j0 - 1 + lj;
        hasB[k] = INTERIOR || (i >= -1 && i <= v.nx && j >= -v.gy && j <= v.ny + v.gy - 1 && !((i < 0 || i >= v.nx) && (j < 0 || j >= v.ny)));
        double b = 0.0, m = 0.0;
        if (hasB[k]) { int idx = cidx(v, i, j); b = Bf[idx]; m = mk[idx]; }
        neg = neg || (m < 0.0 && i >= 0 && i < v.nx && j >= 0 && j < v.ny);      // (no branch here: the loads of the unrolled rows stay batched)
        Br[k] = b;
        sB[lj * RW + tx] = b; sM[lj * RW + tx] = m;
    }
    // this pass sees the ice mask of every cell of the level anyway: it leaves word whether any is negative, so that the relaxation
    // launches of the same V-cycle may skip reading the array (suhmo_gsrb.hip; COMPUTENONLINEARTERMS only asks mask < 0)
    if (negflag && neg) *negflag = epoch;
    __syncthreads();
    // cell-centred gradient of the cell at phi-tile position p (k_gradcc); (gi, gj) = its indices
    auto gradcc = [&](int p, int gi, int gj, double &gx, double &gy) {
        double c = sphi[p], w = sphi[p - 1], e = sphi[p + 1], s = sphi[p - PW], n = sphi[p + PW];
        double gW = v.fdx * (c - w), gE = v.fdx * (e - c), gS = v.fdy * (c - s), gN = v.fdy * (n - c);
        if (hasMask) {
            int idx = cidx(v, wrapx(gi), wrapy(gj));
            bool mc = mk[idx] < 1e-6;
            if (mc || mk[idx - 1] < 1e-6) gW = 0.0;
            if (mc || mk[idx + 1] < 1e-6) gE = 0.0;
            if (mc || mk[idx - v.P] < 1e-6) gS = 0.0;
            if (mc || mk[idx + v.P] < 1e-6) gN = 0.0;
        }
        gx = 0.5 * (gW + gE); gy = 0.5 * (gS + gN);
    };
    double rer[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int lj = ty + TY * k, j = j0 - 1 + lj;
        const bool yi = yin(j);
        const int p = (lj + 1) * PW + (tx + 1);
        double gx = 0.0, gy = 0.0;
        if (xi && yi) {
            gradcc(p, i, j, gx, gy);
        } else if (xi != yi) {
            // first ghost layer on a non-periodic domain side: linear extrapolation of the two
            // interior neighbours' gradients (k_grad_ghosts, util/ExtrapBCF.ChF:21-29)
            int d = 0, di = 0, dj = 0;
            if (!xi && (i == -1 || i == v.nx)) { di = i < 0 ? 1 : -1; d = di; }
            else if (!yi && (j == -1 || j == v.ny)) { dj = j < 0 ? 1 : -1; d = dj * PW; }
            if (d != 0) {
                double g1x, g1y, g2x, g2y;
                gradcc(p + d, i + di, j + dj, g1x, g1y);
                gradcc(p + 2 * d, i + 2 * di, j + 2 * dj, g2x, g2y);
                gx = 2.0 * g1x - g2x; gy = 2.0 * g1y - g2y;
            }
        }
        // Re on the (ghosted) range (k_re)
        double re = 0.0;
        if (hasB[k]) {
            double B = Br[k];
            double sg = sqrt(gx * gx + gy * gy);
            double discr = 1.0 + 4.0 * ph.omega * (B * B * B * ph.grav * sg) / (12.0 * ph.nu * ph.nu);
            re = (-1.0 + sqrt(discr)) / (2.0 * ph.omega);
        }
        rer[k] = re;
    }
    __syncthreads();                                 // every lane is done reading the phi tile
#pragma unroll
    for (int k = 0; k < NK; k++) sre[(ty + TY * k) * RW + tx] = rer[k];
    __syncthreads();
    // ---- faces (k_bcoef_faces): lane tx >= 1 owns cell column i (its W and S faces); the last tile
    // column / row also owns the domain's E / N faces
    const int nxt = (!INTERIOR && i0 + BT_X >= v.nx) ? v.nx - i0 + 1 : BT_X, nyt = (!INTERIOR && j0 + BT_Y >= v.ny) ? v.ny - j0 + 1 : BT_Y;
    const int fx = tx - 1;                           // face column index inside the tile
    if (fx >= 0 && fx < nxt) {
        for (int fy = ty; fy < nyt; fy += TY) {
            const int j = j0 + fy, idx = cidx(v, i, j), r = (fy + 1) * RW + tx;
            if (INTERIOR || j < v.ny)
                bxo[idx] = bcoef_face(ph, sre[r], sre[r - 1], sB[r], sB[r - 1], sM[r], sM[r - 1], !INTERIOR && (i == 0 || i == v.nx));
            if (INTERIOR || i < v.nx) {
                int jg = j + v.j0;
                byo[idx] = bcoef_face(ph, sre[r], sre[r - RW], sB[r], sB[r - RW], sM[r], sM[r - RW], !INTERIOR && (jg == 0 || jg == v.nyg));
            }
        }
    }
}
// rank strip: the relaxation also reads the ice mask of its halo rows (the neighbours' cells, on every depth that streams); k_bcoef_fused
// reports on the strip's own cells, this one on the stored halo rows of the depths [0, nd)
struct MaskHalo { const double *m[SUHMO_MAXDEPTH]; int nx[SUHMO_MAXDEPTH], ny[SUHMO_MAXDEPTH], P[SUHMO_MAXDEPTH], gy[SUHMO_MAXDEPTH]; int nd, lo, hi; };
__global__ __launch_bounds__(256) void k_mask_halo_report(MaskHalo h, unsigned *negflag, unsigned epoch)
{
    const int d = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= h.nd || i >= h.nx[d]) return;
    const int g = h.gy[d];
    bool neg = false;
    for (int r = blockIdx.y; r < 2 * g; r += gridDim.y) {
        const bool top = r >= g;
        if (top ? !h.hi : !h.lo) continue;
        const int j = top ? h.ny[d] + (r - g) : -1 - r;
        neg = neg || h.m[d][(long)(j + g) * h.P[d] + SUHMO_XOFF + i] < 0.0;
    }
    if (neg) *negflag = epoch;
}
template <int BT_X, int BT_Y>
__global__ __launch_bounds__(256) void k_bcoef_fused(DV v, FP fp, suhmo_phys_t ph, int hasMask, unsigned *negflag, unsigned epoch)
{
    __shared__ double sphi[(BT_X + 4) * (BT_Y + 4)], sB[(BT_X + 2) * (BT_Y + 2)], sM[(BT_X + 2) * (BT_Y + 2)];
    const int i0 = blockIdx.x * BT_X, j0 = blockIdx.y * BT_Y;
    const bool interior = i0 - 2 >= 0 && i0 + BT_X + 1 <= v.nx - 1 && j0 - 2 >= 0 && j0 + BT_Y + 1 <= v.ny - 1;
    if (interior) bcoef_tile<true, BT_X, BT_Y>(v, fp, ph, hasMask, sphi, sB, sM, negflag, epoch);
    else bcoef_tile<false, BT_X, BT_Y>(v, fp, ph, hasMask, sphi, sB, sM, negflag, epoch);
}

extern "C" int suhmo_level_update_operator(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::UpdateOperator");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    // fused single-kernel path: needs >= 3 cells per direction (extrapolation sources inside every
    // edge tile) and, on rank boundaries, 2 exchanged phi rows for the halo-row gradient
    bool fused = L->bcoef_fused && D.v.nx >= 4 && D.v.ny >= 4 && (!(D.v.ext[0] || D.v.ext[1]) || (D.v.gy >= 2 && D.v.ny >= 2))
                 && L->desc.nx_global == 0;      // AMR patches: un-fused kernels (coarse-fine ghosts are stored data)
    int rc = suhmo_ensure_phi_halo(L, depth, fused ? 2 : 1, st); if (rc) return rc;
    if (fused) {
        const bool wide = L->bcoef_tile_x == 126 && D.v.nx >= 256;          // tiles of 126 x 14 cells on 128 x 2 threads (else 62 x 14 on 64 x 4)
        const int BX = wide ? 126 : 62, BY = 14;
        dim3 grd((D.v.nx + BX - 1) / BX, (D.v.ny + BY - 1) / BY);   // the last tile column / row also owns the E / N faces
        // depth 0 of a whole level: the kernel also reports (device word = this call's number) whether the ice mask has a negative cell
        // (the V-cycle that called takes the report up, suhmo_fas.hip: it holds until that cycle ends, not across calls of this entry point)
        unsigned *flag = nullptr;
        if (depth == 0) { L->maskflag_epoch = 0; L->mask_reported = 0; }
        if (depth == 0 && L->skip_mask) {
            flag = (unsigned *)(L->scratch + L->scratch_elems - 1);
            if (++L->mask_epoch == 0) L->mask_epoch = 1;
            L->mask_reported = 1;
        }
        if (wide) hipLaunchKernelGGL((k_bcoef_fused<126, 14>), grd, dim3(128, 2), 0, st, D.v, D.fp, L->ph, L->ph.use_mask_gradients, flag, L->mask_epoch);
        else hipLaunchKernelGGL((k_bcoef_fused<62, 14>), grd, dim3(64, 4), 0, st, D.v, D.fp, L->ph, L->ph.use_mask_gradients, flag, L->mask_epoch);
        if (flag && (D.v.ext[0] || D.v.ext[1])) {
            MaskHalo h;
            h.nd = 0; h.lo = D.v.ext[0]; h.hi = D.v.ext[1];
            const int last = L->coarse_mask_ok ? (L->agg ? L->agg_depth : L->ndepth) : 1;      // (agglomerated depths keep no halo rows)
            int gmax = 1;
            for (int k = 0; k < last && k < SUHMO_MAXDEPTH; k++) {
                const Depth &Dk = L->d[k];
                h.m[k] = Dk.fp.f[SUHMO_F_MASK]; h.nx[k] = Dk.v.nx; h.ny[k] = Dk.v.ny; h.P[k] = Dk.v.P; h.gy[k] = Dk.v.gy;
                if (Dk.v.gy > gmax) gmax = Dk.v.gy;
                h.nd = k + 1;
            }
            hipLaunchKernelGGL(k_mask_halo_report, dim3((D.v.nx + 255) / 256, 2 * gmax, h.nd), dim3(256), 0, st, h, flag, L->mask_epoch);
        }
    } else {
        if (depth == 0) { L->maskflag_epoch = 0; L->mask_reported = 0; }
        if (!suhmo_field(L, depth, SUHMO_F_GRADX) || !suhmo_field(L, depth, SUHMO_F_GRADY) || !suhmo_field(L, depth, SUHMO_F_RE)) return -2;
        hipLaunchKernelGGL(k_gradcc, grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph.use_mask_gradients);
        rc = suhmo_exchange_fields(L, depth, {SUHMO_F_GRADX, SUHMO_F_GRADY}, st); if (rc) return rc;
        int n = 2 * D.v.ny + 2 * D.v.nx;
        hipLaunchKernelGGL(k_grad_ghosts, dim3((n + 255) / 256), dim3(256), 0, st, D.v, D.fp.f[SUHMO_F_GRADX], D.fp.f[SUHMO_F_GRADY]);
        hipLaunchKernelGGL(k_re, grid2d(D.v.nx + 2, D.v.ny + 2), BLK2D, 0, st, D.v, D.fp, L->ph);
        hipLaunchKernelGGL(k_bcoef_faces, grid2d(D.v.nx + 1, D.v.ny + 1), BLK2D, 0, st, D.v, D.fp, L->ph);
    }
    HIPCHK(hipGetLastError());
    // strips: the fused relaxation recomputes halo rows, so it needs the coefficients there too (faces_deferred: the V-cycle sends them
    // with the coarse depths' faces, one message for all depths: suhmo_average_operator_all)
    if (depth == 0 && L->faces_deferred) return 0;
    rc = suhmo_exchange_fields(L, depth, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc;
    return 0;
}

// pieces of the un-fused WFlx_level for the AMR fine level (suhmo_amr.hip): cell-centred gradient with its
// domain-side ghosts; then (after the coarse-fine ghosts were interpolated) Re and bCoef
int suhmo_grad_cc(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_GRADX) || !suhmo_field(L, depth, SUHMO_F_GRADY) || !suhmo_field(L, depth, SUHMO_F_RE)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    hipLaunchKernelGGL(k_gradcc, grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph.use_mask_gradients);
    rc = suhmo_exchange_fields(L, depth, {SUHMO_F_GRADX, SUHMO_F_GRADY}, st); if (rc) return rc;    // lvlgradH.exchange() :1490
    int n = 2 * D.v.ny + 2 * D.v.nx;
    hipLaunchKernelGGL(k_grad_ghosts, dim3((n + 255) / 256), dim3(256), 0, st, D.v, D.fp.f[SUHMO_F_GRADX], D.fp.f[SUHMO_F_GRADY]);
    HIPCHK(hipGetLastError());
    return 0;
}
// GRADX / GRADY at the listed cells only (device list of (i, j)); nothing else of the two fields is touched
int suhmo_grad_cc_list(suhmo_level *L, int depth, const int2 *d_cells, int n, hipStream_t st)
{
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_GRADX) || !suhmo_field(L, depth, SUHMO_F_GRADY) || !suhmo_field(L, depth, SUHMO_F_RE)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (n) hipLaunchKernelGGL(k_gradcc_list, dim3((n + 255) / 256), dim3(256), 0, st, D.v, D.fp, L->ph.use_mask_gradients, d_cells, n);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_re_bcoef_unfused(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth];
    hipLaunchKernelGGL(k_re, grid2d(D.v.nx + 2, D.v.ny + 2), BLK2D, 0, st, D.v, D.fp, L->ph);
    hipLaunchKernelGGL(k_bcoef_faces, grid2d(D.v.nx + 1, D.v.ny + 1), BLK2D, 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}

// aCoeff_bCoeff (src/AmrHydro.cpp:1781-1817, called at :3087-3102): the bCoef the solver's operators are defined with, from the
// lagged Re and gap height of the time step (RE, B with their ghosts) -- the first residual of a solve sees it
int suhmo_bcoef_faces(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth];
    hipLaunchKernelGGL(k_bcoef_faces, grid2d(D.v.nx + 1, D.v.ny + 1), BLK2D, 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_re_cells(suhmo_level *L, int depth, hipStream_t st)      // COMPUTERE on the ghosted box (time step on a hierarchy)
{
    Depth &D = L->d[depth];
    hipLaunchKernelGGL(k_re, grid2d(D.v.nx + 2, D.v.ny + 2), BLK2D, 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
// grad h (cell centred, extrapolated ghosts) and Re on the ghosted level, for the time step
// (suhmo_step.hip): the un-fused steps 1-3 above
int suhmo_grad_re(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth];
    int rc = suhmo_grad_cc(L, depth, st); if (rc) return rc;       // rank strips: phi halo row + exchange of the gradient
    hipLaunchKernelGGL(k_re, grid2d(D.v.nx + 2, D.v.ny + 2), BLK2D, 0, st, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}

// AverageOperator: CoarseAverageFace(bCoef[0] -> bCoef[depth], ratio r = 2^depth), sequential
// sum of the r collinear fine faces divided by r  (src/VCAMRNonLinearPoissonOp.cpp:66-95)
__global__ void k_average_faces(DV vf, const double *__restrict__ bxf, const double *__restrict__ byf,
                                DV vc, double *__restrict__ bxc, double *__restrict__ byc, int r)
{
    int ic = blockIdx.x * blockDim.x + threadIdx.x, jc = blockIdx.y * blockDim.y + threadIdx.y;
    if (ic > vc.nx || jc > vc.ny) return;
    if (jc < vc.ny) {
        double sm = 0.0;
        int base = cidx(vf, ic * r, jc * r);
        for (int k = 0; k < r; k++) sm = sm + bxf[base + k * vf.P];
        bxc[cidx(vc, ic, jc)] = sm / (double)r;
    }
    if (ic < vc.nx) {
        double sm = 0.0;
        int base = cidx(vf, ic * r, jc * r);
        for (int k = 0; k < r; k++) sm = sm + byf[base + k];
        byc[cidx(vc, ic, jc)] = sm / (double)r;
    }
}
// All depths of AverageOperator in ONE pass over the depth-0 faces (the V-cycle refreshes every
// depth right after UpdateOperator).  The reference's arithmetic is a sequential sum of the
// r = 2^d collinear fine faces divided by r; the running sum of the first r/2 faces of a group IS
// the (unscaled) depth d-1 sum, so one walk over 2^(nd-1) faces yields every depth bit for bit.
struct AvgOut { double *bx[SUHMO_MAXDEPTH], *by[SUHMO_MAXDEPTH]; int P[SUHMO_MAXDEPTH]; int gy[SUHMO_MAXDEPTH]; };
// x-faces: thread = (even fine column i, block of R = 2^(nd-1) rows); walks the rows
__device__ __forceinline__ void d_average_faces_x_all(const DV &vf, const double *__restrict__ bxf, const AvgOut &o, int nd, int bix, int biy, int tx, int ty)
{
    const int R = 1 << (nd - 1);
    int ih = bix * 64 + tx;                              // i = 2 * ih
    int jb = biy * 4 + ty;
    int i = 2 * ih;
    if (i > vf.nx || jb * R >= vf.ny) return;
    double sum[SUHMO_MAXDEPTH];
    const int base = cidx(vf, i, jb * R);
    auto take = [&](int k, double f) {
#pragma unroll
        for (int d = 1; d < SUHMO_MAXDEPTH; d++) {
            if (d >= nd) break;
            const int r = 1 << d;
            if ((i & (r - 1)) != 0) break;               // column not on depth d's face grid (nor deeper)
            sum[d] = ((k & (r - 1)) == 0) ? 0.0 + f : sum[d] + f;
            if ((k & (r - 1)) == r - 1)
                o.bx[d][((jb * R + k) / r + o.gy[d]) * o.P[d] + SUHMO_XOFF + i / r] = sum[d] / (double)r;
        }
    };
    if (R >= 8) {
        for (int k0 = 0; k0 < R; k0 += 8) {              // 8 independent loads in flight, then the (sequential) sums
            double f8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) f8[u] = bxf[base + (k0 + u) * vf.P];
#pragma unroll
            for (int u = 0; u < 8; u++) take(k0 + u, f8[u]);
        }
    } else {
        for (int k = 0; k < R; k++) take(k, bxf[base + k * vf.P]);
    }
}
// y-faces: one wave walks 64 consecutive columns of YR even fine rows (their loads in flight together); lane l = column
#define AVG_YR 8
__device__ __forceinline__ void d_average_faces_y_all(const DV &vf, const double *__restrict__ byf, const AvgOut &o, int nd, int bix, int biy, int tid)
{
    const int lane = tid & 63;
    const int i = bix * 64 + lane;
    const int jb = 2 * AVG_YR * (biy * 4 + (tid >> 6));
    if (jb > vf.ny) return;                               // whole wave leaves together
    double fr[AVG_YR];
#pragma unroll
    for (int q = 0; q < AVG_YR; q++) {
        const int j = jb + 2 * q;
        fr[q] = (i < vf.nx && j <= vf.ny) ? byf[cidx(vf, i, j)] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < AVG_YR; q++) {
        const int j = jb + 2 * q;
        if (j > vf.ny) break;                             // uniform
        const double f = fr[q];
        double run = 0.0 + f;                             // depth-0 "sum" of a single face
        for (int d = 1; d < nd; d++) {
            const int r = 1 << d;
            if ((j & (r - 1)) != 0) break;                // row not on depth d's face grid
            // sequential continuation: (((run + f[l + r/2]) + f[l + r/2 + 1]) + ... + f[l + r - 1])
            double acc = run;
            for (int k = r / 2; k < r; k++) acc = acc + __shfl(f, (lane + k) & 63);
            run = acc;                                    // valid on lanes with (lane % r) == 0
            if ((lane & (r - 1)) == 0 && i < vf.nx)
                o.by[d][(j / r + o.gy[d]) * o.P[d] + SUHMO_XOFF + i / r] = run / (double)r;
        }
    }
}

// both directions in ONE launch of 256-thread workgroups: the first gxx * gxy of them are the x-face walkers (64 x 4 threads), the rest the
// y-face walkers (four waves): nothing of one reads what the other writes
__global__ __launch_bounds__(256) void k_average_faces_all(DV vf, const double *__restrict__ bxf, const double *__restrict__ byf, AvgOut o, int nd, int gxx, int gxy, int gyx)
{
    const int b = blockIdx.x, nbx = gxx * gxy;
    if (b < nbx) d_average_faces_x_all(vf, bxf, o, nd, b % gxx, b / gxx, threadIdx.x & 63, threadIdx.x >> 6);
    else d_average_faces_y_all(vf, byf, o, nd, (b - nbx) % gyx, (b - nbx) / gyx, threadIdx.x);
}

extern "C" int suhmo_level_average_operator(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::AverageOperator");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    if (depth == 0) return 0;
    HIPCHK(hipSetDevice(L->device));
    Depth &F = L->d[0], &C = L->d[depth];
    hipLaunchKernelGGL(k_average_faces, grid2d(C.v.nx + 1, C.v.ny + 1), BLK2D, 0, (hipStream_t)s, F.v, F.fp.f[SUHMO_F_BX], F.fp.f[SUHMO_F_BY],
                       C.v, C.fp.f[SUHMO_F_BX], C.fp.f[SUHMO_F_BY], 1 << depth);
    HIPCHK(hipGetLastError());
    int rc = suhmo_exchange_fields(L, depth, {SUHMO_F_BX, SUHMO_F_BY}, (hipStream_t)s); if (rc) return rc;
    return 0;
}

int suhmo_average_operator_all(suhmo_level *L, int nd, hipStream_t st)
{
    Depth &F = L->d[0];
    const bool d0 = L->faces_deferred != 0;          // the halo rows of the depth-0 faces are still to travel
    L->faces_deferred = 0;
    if (nd < 2) return d0 ? suhmo_exchange_fields(L, 0, {SUHMO_F_BX, SUHMO_F_BY}, st) : 0;
    if (nd > 7 || F.v.nx % (1 << (nd - 1)) || F.v.ny % (1 << (nd - 1))) {     // generic fallback
        if (d0) { int rc = suhmo_exchange_fields(L, 0, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc; }
        for (int k = 1; k < nd; k++) { int rc = suhmo_level_average_operator(L, k, (suhmo_stream_t)st); if (rc) return rc; }
        return suhmo_agg_gather_faces(L, nd, st);
    }
    AvgOut o;
    for (int d = 0; d < nd; d++) { o.bx[d] = L->d[d].fp.f[SUHMO_F_BX]; o.by[d] = L->d[d].fp.f[SUHMO_F_BY]; o.P[d] = L->d[d].v.P; o.gy[d] = L->d[d].v.gy; }
    const int R = 1 << (nd - 1);
    dim3 gx((F.v.nx / 2 + 1 + 63) / 64, (F.v.ny / R + 3) / 4);
    dim3 gy((F.v.nx + 63) / 64, ((F.v.ny / 2 + 1 + AVG_YR - 1) / AVG_YR + 3) / 4);
    hipLaunchKernelGGL(k_average_faces_all, dim3(gx.x * gx.y + gy.x * gy.y), dim3(256), 0, st, F.v, F.fp.f[SUHMO_F_BX], F.fp.f[SUHMO_F_BY], o, nd, (int)gx.x, (int)gx.y, (int)gy.x);
    HIPCHK(hipGetLastError());
    // strips: the coarse face coefficients of all depths travel as one message group when the transport can batch
    if (L->ipc) { int rc = suhmo_ipc_batch(L, 1, st); if (rc) return rc; }
    else if (L->ex_begin && L->ex) { int rc = L->ex_begin(L->user); if (rc) return rc; }
    for (int k = d0 ? 0 : 1; k < nd && !(L->agg && k >= L->agg_depth); k++) {
        int rc = suhmo_exchange_fields(L, k, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc;
    }
    if (L->ipc) { int rc = suhmo_ipc_batch(L, 0, st); if (rc) return rc; }
    else if (L->ex_end && L->ex) { int rc = L->ex_end(L->user, L, (suhmo_stream_t)st); if (rc) return rc; }
    return suhmo_agg_gather_faces(L, nd, st);       // agglomerated depths: every rank's rows of the coarse faces -> the whole-level copy
}

// MGnewOp coefficient coarsening: CoarseAverage (arithmetic) of aCoef, B, Pi, zb, iceMask from
// depth 0 with ratio r: sequential sum (ii fastest) * 1/r^2 (src/VCAMRNonLinearPoissonOp.cpp:1116-1138)
// The five coefficient fields of every coarse depth in ONE launch (blockIdx.z = (depth - 1) * 5 + field): each depth averages
// depth 0 directly, so they are independent.  The sum of a coarse cell is one sequential chain of r*r additions whatever the
// kernel does, so at the deep depths (few coarse cells, r = 16, 32) the time is the chain plus the latency of its loads: a row
// of the block is fetched as r/2 independent 16-byte loads, then added in order.
struct AvgDepth { int nx, ny, P, gy; int boff, nbx; double *c[5]; };   // boff: first workgroup of the depth, nbx: its workgroups per row of tiles
struct AvgAll { const double *f[5]; AvgDepth d[SUHMO_MAXDEPTH - 1]; };
template <int R>
__device__ __forceinline__ double average_block(const double *__restrict__ f, int base, int P, int r_)
{
    const int r = R ? R : r_;
    double sm = 0.0;
    if constexpr (R >= 2) {
        for (int jj = 0; jj < R; jj++) {
            double2 row[R / 2];
#pragma unroll
            for (int k = 0; k < R / 2; k++) row[k] = *reinterpret_cast<const double2 *>(f + base + jj * P + 2 * k);   // ic * r is even
#pragma unroll
            for (int k = 0; k < R / 2; k++) { sm = sm + row[k].x; sm = sm + row[k].y; }
        }
    } else {
        for (int jj = 0; jj < r; jj++)
            for (int ii = 0; ii < r; ii++) sm = sm + f[base + jj * P + ii];
    }
    return sm * (1.0 / (double)(r * r));
}
__global__ __launch_bounds__(256) void k_average_cells_all(DV vf, AvgAll a, int nd)
{
    // workgroups are numbered depth by depth (a grid sized for the largest depth would dispatch mostly empty ones)
    int dep = 1;
    while (dep + 1 < nd && (int)blockIdx.x >= a.d[dep].boff) dep++;
    const AvgDepth &C = a.d[dep - 1];
    const int local = blockIdx.x - C.boff, q = local % 5, tile = local / 5;
    const int ic = (tile % C.nbx) * blockDim.x + threadIdx.x, jc = (tile / C.nbx) * blockDim.y + threadIdx.y;
    if (ic >= C.nx || jc >= C.ny) return;
    const double *__restrict__ f = a.f[q];
    const int r = 1 << dep, base = cidx(vf, ic * r, jc * r);
    double m;
    switch (r) {
    case 2: m = average_block<2>(f, base, vf.P, r); break;
    case 4: m = average_block<4>(f, base, vf.P, r); break;
    case 8: m = average_block<8>(f, base, vf.P, r); break;
    case 16: m = average_block<16>(f, base, vf.P, r); break;
    case 32: m = average_block<32>(f, base, vf.P, r); break;
    default: m = average_block<0>(f, base, vf.P, r); break;
    }
    C.c[q][(jc + C.gy) * C.P + SUHMO_XOFF + ic] = m;
}
// ghosts of coarse B / Pi / zb / mask: periodic wrap or Neumann copy (NeumBCForB :1309-1341)
__device__ __forceinline__ void d_coef_ghosts(const DV &v, double *__restrict__ p)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * v.ny) {
        int side = t / v.ny, j = t % v.ny;
        if (v.cfx[side]) return;
        if (side == 0) { int idx = cidx(v, 0, j); p[idx - 1] = v.per[0] ? p[idx + v.nx - 1] : p[idx]; }
        else { int idx = cidx(v, v.nx - 1, j); p[idx + 1] = v.per[0] ? p[idx - (v.nx - 1)] : p[idx]; }
        return;
    }
    t -= 2 * v.ny;
    if (t < 2 * v.nx) {
        int side = t / v.nx, i = t % v.nx;
        if (v.ext[side]) return;
        if (side == 0) { int idx = cidx(v, i, 0); p[idx - v.P] = v.per[1] ? p[idx + (v.ny - 1) * v.P] : p[idx]; }
        else { int idx = cidx(v, i, v.ny - 1); p[idx + v.P] = v.per[1] ? p[idx - (v.ny - 1) * v.P] : p[idx]; }
    }
}
__global__ void k_coef_ghosts(DV v, double *__restrict__ p)
{
    d_coef_ghosts(v, p);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_coef_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field)
{
    d_coef_ghosts(vt[blockIdx.z], ft[blockIdx.z].f[field]);
}
__global__ void k_coef_ghosts_all(DV v0, AvgAll a)        // B, Pi, zb, mask of every coarse depth (blockIdx.y = (depth - 1) * 4 + field - 1)
{
    const AvgDepth &C = a.d[blockIdx.y / 4];
    double *__restrict__ p = C.c[1 + blockIdx.y % 4];
    const int nx = C.nx, ny = C.ny, P = C.P;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * ny) {
        int side = t / ny, j = t % ny;
        if (v0.cfx[side]) return;
        if (side == 0) { int idx = (j + C.gy) * P + SUHMO_XOFF; p[idx - 1] = v0.per[0] ? p[idx + nx - 1] : p[idx]; }
        else { int idx = (j + C.gy) * P + SUHMO_XOFF + nx - 1; p[idx + 1] = v0.per[0] ? p[idx - (nx - 1)] : p[idx]; }
        return;
    }
    t -= 2 * ny;
    if (t < 2 * nx) {
        int side = t / nx, i = t % nx;
        if (v0.ext[side]) return;
        if (side == 0) { int idx = C.gy * P + SUHMO_XOFF + i; p[idx - P] = v0.per[1] ? p[idx + (ny - 1) * P] : p[idx]; }
        else { int idx = (ny - 1 + C.gy) * P + SUHMO_XOFF + i; p[idx + P] = v0.per[1] ? p[idx - (ny - 1) * P] : p[idx]; }
    }
}
// exchange + CopyGhostCells of a cell field (util/ExtrapGhostCells.cpp:182-269)
int suhmo_copy_ghosts(suhmo_level *L, int depth, int field, hipStream_t st)
{
    Depth &D = L->d[depth];
    double *p = suhmo_field(L, depth, field);
    if (!p) return -2;
    int n = 2 * D.v.ny + 2 * D.v.nx;
    hipLaunchKernelGGL(k_coef_ghosts, dim3((n + 255) / 256), dim3(256), 0, st, D.v, p);
    HIPCHK(hipGetLastError());
    return 0;
}
// with_faces = false: the caller's cycle re-averages bCoef itself (bcoeff_otf: UpdateOperator + AverageOperator every V-cycle)
int suhmo_build_mg_coefficients(suhmo_level *L, bool with_faces, hipStream_t st)
{
    static const int fields[5] = {SUHMO_F_ACOEF, SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK};
    Depth &F = L->d[0];
    const int nd = L->ndepth;
    if (nd > 1) {
        AvgAll a;
        int nblocks = 0;
        for (int q = 0; q < 5; q++) a.f[q] = F.fp.f[fields[q]];
        for (int dep = 1; dep < nd; dep++) {
            const Depth &C = L->d[dep];
            AvgDepth &o = a.d[dep - 1];
            o.nx = C.v.nx; o.ny = C.v.ny; o.P = C.v.P; o.gy = C.v.gy;
            for (int q = 0; q < 5; q++) o.c[q] = C.fp.f[fields[q]];
            o.boff = nblocks; o.nbx = (C.v.nx + 63) / 64;
            nblocks += 5 * o.nbx * ((C.v.ny + 3) / 4);
        }
        const Depth &C1 = L->d[1];
        hipLaunchKernelGGL(k_average_cells_all, dim3(nblocks), dim3(64, 4), 0, st, F.v, a, nd);
        const int n = 2 * C1.v.ny + 2 * C1.v.nx;
        hipLaunchKernelGGL(k_coef_ghosts_all, dim3((n + 255) / 256, 4 * (nd - 1)), dim3(256), 0, st, F.v, a);
        HIPCHK(hipGetLastError());
    }
    for (int dep = 1; dep < nd; dep++) {
        int rc;
        if (with_faces && (rc = suhmo_level_average_operator(L, dep, (suhmo_stream_t)st))) return rc;
        if (L->agg && dep >= L->agg_depth) continue;                  // agglomerated depths: no halo rows, the whole rows travel below
        rc = suhmo_exchange_fields(L, dep, {SUHMO_F_ACOEF, SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK}, st); if (rc) return rc;
    }
    L->coarse_mask_ok = 1;
    return suhmo_agg_gather_static(L, with_faces, st);
}
extern "C" int suhmo_level_build_mg_coefficients(suhmo_level_t *L, suhmo_stream_t s)
{
    ARG(L);
    HIPCHK(hipSetDevice(L->device));
    return suhmo_build_mg_coefficients(L, true, (hipStream_t)s);
}

// ---- every box of a multi-box AMR level in one launch
int suhmo_multi_grad_cc(const suhmo_multi &m, int hasMask, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    if (m.merged) {
        hipLaunchKernelGGL(k_gradcc_ghosts_m, grid_m(m), BLK2D, 0, st, m.dv, m.fp, hasMask);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k_gradcc_m, grid_m(m), BLK2D, 0, st, m.dv, m.fp, hasMask);
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_grad_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_levels_grad_cc(const suhmo_lvboxes &lv, int hasMask, hipStream_t st)
{
    int nz = 0;
    for (int q = 0; q < lv.n; q++) nz += lv.nbox[q];
    if (nz <= 0) return 0;
    hipLaunchKernelGGL(k_gradcc_ghosts_lv, dim3((lv.maxnx + 63) / 64, (lv.maxny + 3) / 4, nz), BLK2D, 0, st, lv, hasMask);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_multi_re(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_re_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp, ph);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_multi_bcoef_faces(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_bcoef_faces_m, grid_m(m, 1, 1), BLK2D, 0, st, m.dv, m.fp, ph);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_multi_re_bcoef(const suhmo_multi &m, const suhmo_phys_t &ph, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    if (!m.merged) { int rc = suhmo_multi_re(m, ph, st); return rc ? rc : suhmo_multi_bcoef_faces(m, ph, st); }
    hipLaunchKernelGGL(k_re_bcoef_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp, ph);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_multi_coef_ghosts(const suhmo_multi &m, int field, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_coef_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp, field);
    HIPCHK(hipGetLastError());
    return 0;
}

