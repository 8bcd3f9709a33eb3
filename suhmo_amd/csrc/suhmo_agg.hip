// suhmo_agg.hip -- agglomeration of the coarse multigrid depths of a level cut into rank strips ("agglomerate MG
// depths whose per-GPU cell count < ~64 K").
//
// The reference keeps the fine box -> rank map on every multigrid depth (coarsen_dbl, src/VCAMRNonLinearPoissonOp.cpp:1059-1060):
// at 8 x 4096^2 strips the deepest depth is 16 k cells per rank and each of its relaxations is a latency-bound message pair.
// Here, from the first depth d_a whose strip holds fewer than `agg_min_cells` cells, the rest of the V-cycle runs REDUNDANTLY on
// every rank, on a whole-level handle A (A's depth k = the level's depth d_a + k):
//   once per coefficient build   all-gather of B, Pi, zb, iceMask (aCoef) of the depths >= d_a        (static during a solve)
//   once per V-cycle             all-gather of the face coefficients of the depths >= d_a             (AverageOperator averages
//                                from depth 0, not in a cascade: every rank averages its own rows, then they travel)
//   at depth d_a - 1             all-gather of R phi and RES of depth d_a -> A forms rhs = res + L(R phi), runs depths d_a ...
//                                bottom without a message, and the rank copies its rows AND its halo rows of phi and R phi back
// i.e. two all-gathers per V-cycle instead of about two message groups per agglomerated depth.  GSRB is colour-Jacobi and every
// other kernel cell-local or a fixed 2 x 2 aggregation, so the bits do not depend on who computes a row (tests/test_gpu_strips.py).
#include "suhmo_common.h"
#include <algorithm>

int suhmo_copy_ghosts(suhmo_level *L, int depth, int field, hipStream_t st);      // suhmo_bcoef.hip

namespace {
constexpr int MAXE = 40;
// one field of one depth: `rows` rows of the strip (from row 0; nx + 2 columns: x-ghost column, cells / faces 0 .. nx) -> rows rk * roff ... of A
struct Seg { const double *src; double *dst; int sP, sgy, dP, dgy, nx, rows, roff; long off; };
struct SegList { Seg e[MAXE]; int n; };
__global__ void k_agg_pack(SegList sl, double *__restrict__ buf)
{
    const Seg &q = sl.e[blockIdx.z];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, r = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > q.nx || r >= q.rows) return;
    buf[q.off + (long)r * (q.nx + 2) + (i + 1)] = q.src[(size_t)(r + q.sgy) * q.sP + SUHMO_XOFF + i];
}
__global__ void k_agg_unpack(SegList sl, const double *__restrict__ buf, long stride)
{
    const int e = blockIdx.z % sl.n, rk = blockIdx.z / sl.n;
    const Seg &q = sl.e[e];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, r = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > q.nx || r >= q.rows) return;
    if (r >= q.roff && rk != (int)(gridDim.z / sl.n) - 1) return;        // the row a strip shares with the next one: the next one writes it
    q.dst[(size_t)(rk * q.roff + r + q.dgy) * q.dP + SUHMO_XOFF + i] = buf[(long)rk * stride + q.off + (long)r * (q.nx + 2) + (i + 1)];
}
// rows [jlo, jhi) of the strip's depth d_a (halo rows included) <- the rows of A that hold them; two fields per launch
__global__ void k_agg_scatter(DV sv, DV av, int joff, int jlo, int jhi, int wrap, double *__restrict__ d0, const double *__restrict__ s0,
                              double *__restrict__ d1, const double *__restrict__ s1)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i > sv.nx || j >= jhi) return;
    int ja = joff + j;
    if (wrap) { if (ja < 0) ja += av.ny; else if (ja >= av.ny) ja -= av.ny; }
    const size_t si = (size_t)(ja + av.gy) * av.P + SUHMO_XOFF + i, di = (size_t)(j + sv.gy) * sv.P + SUHMO_XOFF + i;
    d0[di] = s0[si];
    d1[di] = s1[si];
}

int gather(suhmo_level *L, const std::vector<Seg> &segs_in, hipStream_t st)
{
    const int world = L->agg_world;
    for (size_t first = 0; first < segs_in.size(); first += MAXE) {
        SegList sl;
        sl.n = (int)std::min<size_t>(MAXE, segs_in.size() - first);
        long count = 0;
        int maxnx = 0, maxrows = 0;
        for (int k = 0; k < sl.n; k++) {
            sl.e[k] = segs_in[first + k];
            sl.e[k].off = count;
            count += (long)sl.e[k].rows * (sl.e[k].nx + 2);
            maxnx = std::max(maxnx, sl.e[k].nx); maxrows = std::max(maxrows, sl.e[k].rows);
        }
        if ((size_t)count > L->agg_cap) {
            if (L->agg_send) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(L->agg_send); (void)hipFree(L->agg_recv); L->agg_send = L->agg_recv = nullptr; }
            L->agg_cap = (size_t)count;
            HIPCHK(hipMalloc(&L->agg_send, L->agg_cap * sizeof(double)));
            HIPCHK(hipMalloc(&L->agg_recv, L->agg_cap * world * sizeof(double)));
        }
        dim3 blk(64, 4), grd((maxnx + 2 + 63) / 64, (maxrows + 3) / 4, sl.n);
        hipLaunchKernelGGL(k_agg_pack, grd, blk, 0, st, sl, L->agg_send);
        HIPCHK(hipGetLastError());
        int rc = L->ag(L->ag_user, L->agg_send, count, L->agg_recv, (suhmo_stream_t)st);
        if (rc) return rc;
        L->agg_gathers++;
        grd.z = sl.n * world;
        hipLaunchKernelGGL(k_agg_unpack, grd, blk, 0, st, sl, L->agg_recv, count);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// field `f` of the level's depth d -> the same field of A's depth d - d_a
int seg_of(suhmo_level *L, int d, int f, Seg &q)
{
    suhmo_level *A = L->agg;
    const int k = d - L->agg_depth;
    double *src = suhmo_field(L, d, f), *dst = suhmo_field(A, k, f);
    if (!src || !dst) { suhmo_set_error("field allocation failed"); return -2; }
    const DV &sv = L->d[d].v, &av = A->d[k].v;
    q.src = src; q.dst = dst; q.sP = sv.P; q.sgy = sv.gy; q.dP = av.P; q.dgy = av.gy; q.nx = sv.nx;
    q.rows = sv.ny + (f == SUHMO_F_BY ? 1 : 0);           // y-faces: face row ny of a strip is face row 0 of the next: every rank sends it (one layout for
                                                          // all ranks), only the LAST rank's is unpacked (it closes A): one writer per row of A
    q.roff = sv.ny; q.off = 0;
    return 0;
}
}  // namespace

void suhmo_agg_release(suhmo_level *L)
{
    if (L->agg) { suhmo_level_destroy(L->agg); L->agg = nullptr; }
    if (L->agg_send) { (void)hipFree(L->agg_send); (void)hipFree(L->agg_recv); L->agg_send = L->agg_recv = nullptr; }
    L->agg_cap = 0; L->agg_depth = 0; L->agg_static_stale = 0;
}

// decide d_a and create A; called when the all-gather transport is attached (and when agg_min_cells changes)
int suhmo_agg_setup(suhmo_level *L)
{
    suhmo_agg_release(L);
    const DV &v0 = L->d[0].v;
    if (!L->ag || L->agg_min_cells <= 0 || !(v0.rk[0] || v0.rk[1]) || L->desc.nx_global > 0) return 0;
    if (v0.ny <= 0 || v0.nyg % v0.ny) return 0;                              // equal strips only
    int da = 0;
    for (int d = 1; d < L->ndepth; d++) if ((long)L->d[d].v.nx * L->d[d].v.ny < L->agg_min_cells) { da = d; break; }
    if (!da) return 0;
    suhmo_level_desc_t desc = L->desc;
    const DV &vd = L->d[da].v;
    desc.nx = vd.nx; desc.ny = v0.nyg >> da; desc.j0 = 0; desc.ny_global = desc.ny; desc.dx = vd.dx; desc.dy = vd.dy;
    desc.nbox = 0; desc.boxes = nullptr; desc.max_box = 1 << (L->ndepth - da);   // boxes that allow exactly the remaining depths (MGnewOp's rule)
    desc.halo_rows = 1; desc.i0 = 0; desc.nx_global = 0; desc.patch_j0 = 0; desc.patch_ny = 0;
    suhmo_level *A = nullptr;
    int rc = suhmo_level_create(&A, &desc);
    if (rc) return rc;
    if (A->ndepth != L->ndepth - da || (desc.ny << da) != v0.nyg) {          // (a level whose size does not halve cleanly: stay on the strips)
        suhmo_level_destroy(A);
        return 0;
    }
    A->graph_max_cells = 0;                                                  // driven depth by depth from the strip's cycle
    L->agg = A; L->agg_depth = da; L->agg_world = v0.nyg / v0.ny; L->agg_rank = v0.j0 / v0.ny;
    L->agg_static_stale = 1;                                                 // (nothing of the level's coefficients is in A yet: suhmo_agg_gather_static)
    return 0;
}

// B, Pi, zb, iceMask (aCoef) of the depths >= d_a, and their face coefficients when the build made them
int suhmo_agg_gather_static(suhmo_level *L, bool with_faces, hipStream_t st)
{
    if (!L->agg) return 0;
    SUHMO_TIME("agglomeration: all-gather of the coarse coefficients");
    std::vector<Seg> segs;
    int rc;
    for (int d = L->agg_depth; d < L->ndepth; d++) {
        for (int f : {SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK, SUHMO_F_ACOEF, SUHMO_F_BX, SUHMO_F_BY}) {
            if (f == SUHMO_F_ACOEF && L->d[0].v.alpha == 0.0) continue;
            if ((f == SUHMO_F_BX || f == SUHMO_F_BY) && !with_faces) continue;
            Seg q;
            if ((rc = seg_of(L, d, f, q))) return rc;
            segs.push_back(q);
        }
    }
    if ((rc = gather(L, segs, st))) return rc;
    L->agg_static_stale = 0;
    for (int k = 0; k < L->agg->ndepth; k++)
        for (int f : {SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK}) if ((rc = suhmo_copy_ghosts(L->agg, k, f, st))) return rc;
    return 0;
}
// face coefficients of the depths d_a ... nd - 1 after this cycle's AverageOperator
int suhmo_agg_gather_faces(suhmo_level *L, int nd, hipStream_t st)
{
    if (!L->agg || nd <= L->agg_depth) return 0;
    SUHMO_TIME("agglomeration: all-gather of the coarse face coefficients");
    std::vector<Seg> segs;
    for (int d = L->agg_depth; d < nd; d++)
        for (int f : {SUHMO_F_BX, SUHMO_F_BY}) { Seg q; int rc = seg_of(L, d, f, q); if (rc) return rc; segs.push_back(q); }
    return gather(L, segs, st);
}
// R phi and RES of depth d_a (just restricted on the strip's rows) -> A's depth 0
int suhmo_agg_gather_state(suhmo_level *L, hipStream_t st)
{
    SUHMO_TIME("agglomeration: all-gather of R phi and RES");
    std::vector<Seg> segs;
    for (int f : {SUHMO_F_PHI, SUHMO_F_RES}) { Seg q; int rc = seg_of(L, L->agg_depth, f, q); if (rc) return rc; segs.push_back(q); }
    int rc = gather(L, segs, st);
    L->agg->d[0].phi_fresh = 0;
    return rc;
}
// phi and R phi (PHIOLD) of depth d_a back onto the strip, halo rows included: the prolongation that follows needs no exchange
int suhmo_agg_scatter(suhmo_level *L, hipStream_t st)
{
    suhmo_level *A = L->agg;
    Depth &C = L->d[L->agg_depth], &AD = A->d[0];
    if (!suhmo_field(L, L->agg_depth, SUHMO_F_PHIOLD) || !suhmo_field(A, 0, SUHMO_F_PHIOLD)) { suhmo_set_error("field allocation failed"); return -2; }
    const int h = C.v.gy < C.v.ny ? C.v.gy : C.v.ny;                      // as deep as an exchange would fill them
    const int jlo = C.v.rk[0] ? -h : 0, jhi = C.v.ny + (C.v.rk[1] ? h : 0);
    const int wrap = AD.v.per[1] ? 1 : 0;
    dim3 blk(64, 4), grd((C.v.nx + 2 + 63) / 64, (jhi - jlo + 3) / 4);
    hipLaunchKernelGGL(k_agg_scatter, grd, blk, 0, st, C.v, AD.v, L->agg_rank * C.v.ny, jlo, jhi, wrap, C.fp.f[SUHMO_F_PHI], AD.fp.f[SUHMO_F_PHI],
                       C.fp.f[SUHMO_F_PHIOLD], AD.fp.f[SUHMO_F_PHIOLD]);
    HIPCHK(hipGetLastError());
    C.phi_fresh = h;
    return 0;
}
