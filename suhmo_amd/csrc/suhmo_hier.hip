// suhmo_hier.hip -- AMR hierarchies whose levels are UNIONS OF BOXES, the way the reference grids them
// (BRMeshRefine with fill_ratio < 1 and block_factor 2: several abutting and disjoint boxes per level,
// src/AmrHydro.cpp:4176-4604, exec/AMR_multiMoulins/run_C_3lev/input.hydro:37,64-83).
//
// Level 0 is one level handle (the domain, with its multigrid depths: every fast kernel of suhmo_gsrb.hip runs there).
// A level l >= 1 is a list of rectangles, each an ordinary level handle created as a patch of the refined domain
// (desc.i0 / nx_global / j0 / ny_global): a rectangle keeps its own ghost ring in its canvas, exactly as a Chombo box
// keeps its own ghost cells -- at a re-entrant corner of the union the same index is the x-ghost of one box and the
// y-ghost of another, with different interpolated values.  What ties the rectangles together is compiled ONCE, when
// the hierarchy is created, into index plans that live in HBM; every inter-box / inter-level step is then one kernel
// launch over a plan, whatever the number of boxes:
//   ff      ghost cell <- the cell of the box of the same level that holds it (Copier::exchange,
//           src/VCAMRNonLinearPoissonOp.cpp:912-913; sides, and corners for the fields exchanged with the default copier)
//   cf      coarse-fine ghost cell <- QuadCFInterp from level l-1, the tangential stencil chosen on the host from the
//           coverage of the coarse cells ([Chombo] QuadCFStencil; oracle/amrm.c:cf_interp states the same rule)
//   pwl     ghost cell (corners included) <- PiecewiseLinearFillPatch from level l-1
//   avg     rectangles (fine box x coarse box) for FORT_AVERAGE / zeroing covered cells
//   win     per fine box the coarse correction over coarsen(box) grown by one cell, gathered from the boxes of level
//           l-1 (the copyTo of AMRProlongS_2, src/AMRNonLinearPoissonOp.cpp:1156), then PROLONG_2_NL
//   reflux  per coarse cell next to coarse-fine faces: its faces in the order (fine box, direction, side)
// Field pointers of levels >= 1 are held in a device table per level (the boxes relax with in-place colour passes, so
// the pointers never move); the base level's pointers travel as a kernel argument (its phi canvases ping-pong).
//
// Cycle = suhmo_amr.hip's, arithmetic = oracle/amrm.c, bit for bit.  [Chombo] pieces are
// restated from upstream Chombo 3.2 (fork not vendored): unpinned against the reference.
#include "suhmo_hier.h"
#include <algorithm>
#include <map>
#include <string>

int suhmo_grad_cc(suhmo_level *L, int depth, hipStream_t st);          // suhmo_bcoef.hip
int suhmo_re_bcoef_unfused(suhmo_level *L, int depth, hipStream_t st);
int suhmo_re_cells(suhmo_level *L, int depth, hipStream_t st);
int suhmo_copy_ghosts(suhmo_level *L, int depth, int field, hipStream_t st);
int suhmo_gsrb_colour_pass(suhmo_level *L, int depth, int pass, hipStream_t st);   // suhmo_gsrb.hip
int suhmo_apply_and_residual_rects(suhmo_level *L, int depth, const int4 *d_rects, int n, int maxw, int maxh, hipStream_t st);   // suhmo_ops.hip
int suhmo_grad_cc_list(suhmo_level *L, int depth, const int2 *d_cells, int n, hipStream_t st);

namespace {
struct Ref { int b, off; };                          // cell of a level: box index, canvas offset
struct CopyEnt { Ref d, s; };
struct CfEnt { Ref f; int step; int kind; int xsign; Ref c[3]; };
// kind: 0 centred (c = cm, c0, cp)   1 forward 2nd order (c0, cp, cpp)   2 forward 1st order (c0, cp)
//       3 backward 2nd order (c0, cm, cmm)   4 backward 1st order (c0, cm)   5 no tangential derivative (c0)
struct PwlEnt { Ref f; Ref c[9]; int par; int sx, sy; };   // c[4] = the coarse cell; b = -1: outside the domain; par: bit0 gi&1, bit1 gj&1
                                                           // sx, sy: slope stencil 0 central, 1 one-sided hi (no lo neighbour), 2 one-sided lo
struct RectEnt { int fb, cb, foff, coff, w, h; };          // average: w x h coarse cells
struct WinEnt { int cb, coff, woff, w, h; };               // window gather: w x h coarse cells into the window buffer
struct Face { int dir, side; int fb, foff; Ref hi, lo, bq; };
struct Target { Ref t; int first, count; };
struct Win { int i0, j0, nx, ny; size_t base; };           // coarse window of a fine box: origin (level l-1 indices), size, offset in the level's buffer

template <class T> struct DevVec {
    T *d = nullptr; size_t n = 0;
    int upload(const std::vector<T> &h)
    {
        n = h.size();
        if (!n) return 0;
        if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return -2;
        if (hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return -2;
        return 0;
    }
    void release() { if (d) (void)hipFree(d); d = nullptr; n = 0; }
};

// spatial index of a level's boxes (bucket grid), for the point and rectangle queries of the plan builder
struct BoxIndex {
    int nxd = 0, nyd = 0, bs = 32, nbx = 0, nby = 0;
    std::vector<int> start, items;
    const std::vector<int> *b4 = nullptr;
    void build(const std::vector<int> &boxes, int nx, int ny)
    {
        b4 = &boxes; nxd = nx; nyd = ny; nbx = (nx + bs - 1) / bs; nby = (ny + bs - 1) / bs;
        std::vector<int> cnt((size_t)nbx * nby + 1, 0);
        const int nb = (int)boxes.size() / 4;
        for (int pass = 0; pass < 2; pass++) {
            for (int k = 0; k < nb; k++) {
                const int *b = &boxes[4 * k];
                for (int by = b[1] / bs; by <= b[3] / bs; by++)
                    for (int bx = b[0] / bs; bx <= b[2] / bs; bx++) {
                        size_t q = (size_t)by * nbx + bx;
                        if (pass == 0) cnt[q + 1]++; else items[start[q] + cnt[q]++] = k;
                    }
            }
            if (pass == 0) {
                start.assign(cnt.size(), 0);
                for (size_t q = 1; q < cnt.size(); q++) start[q] = start[q - 1] + cnt[q];
                items.resize(start.back());
                std::fill(cnt.begin(), cnt.end(), 0);
            }
        }
    }
    int find(int i, int j) const            // box holding cell (i,j) (inside the domain), -1 = none
    {
        size_t q = (size_t)(j / bs) * nbx + i / bs;
        for (int p = start[q]; p < start[q + 1]; p++) {
            const int *b = &(*b4)[4 * items[p]];
            if (i >= b[0] && i <= b[2] && j >= b[1] && j <= b[3]) return items[p];
        }
        return -1;
    }
};

// ---- levels dealt to the ranks (owner computes).  A cell that a plan executed on one rank reads from a box another rank owns travels
// as ONE packed value: the owner packs the cells somebody needs (send, in a fixed order), one all-gather moves every rank's segment,
// the reader scatters what it needs into its MIRROR of the owner's box (recv: mirror cell, owner rank, position in that segment).
// Mirrors exist only for boxes some plan of this rank reads; every other foreign box is a stub without storage.
struct SyncRecv { Ref d; int rank, pos; };
struct Sync {
    DevVec<Ref> send; DevVec<SyncRecv> recv; long stride = 0;         // stride: longest segment over the ranks (0: nothing travels, no collective)
    void release() { send.release(); recv.release(); stride = 0; }
};
struct Xf { int owner, b, off, reader; };                             // plan building: cell (b, off) of `owner` is read on `reader`
struct PutEnt { int cb, coff, rank, pos, w, h; };                     // w x h averaged cells arriving in `rank`'s segment at pos -> box cb at coff
struct HLev {
    int l = 0, nxd = 0, nyd = 0;
    std::vector<suhmo_level *> box;
    std::vector<int> b4;
    BoxIndex index;
    // plans (device)
    DevVec<CopyEnt> ff_side, ff_all;                 // ff_all = sides, then corners (the default copier's exchange in one launch)
    DevVec<int2> push; DevVec<int> pbase;            // ff_side seen from the source cell (the colour passes push, suhmo_gsrb.hip)
    DevVec<int2> halo; DevVec<int> hbase; bool halo_ok = false;   // per box the cells of its 4-cell surroundings (box, canvas offset; -1: no cell of the level): two sweeps per launch
    DevVec<CfEnt> cf;
    DevVec<PwlEnt> pwl;
    DevVec<RectEnt> avg; int avg_w = 0, avg_h = 0;
    DevVec<WinEnt> wing; int wing_w = 0, wing_h = 0; DevVec<int> wstart; int win_max = 0;   // wstart[k] .. wstart[k + 1]: the pieces of box k's window
    DevVec<Target> targets; DevVec<Face> faces;
    // level 1 only, in the cells of this rank's part of level 0: the rectangles whose L(phi) / residual change when level 1's head is
    // averaged down (coarsen(box) grown by one cell, periodic images included), and the cells its gradient interpolation reads
    DevVec<int4> dirty0; int dirty_w = 0, dirty_h = 0; DevVec<int2> gcells;
    std::vector<Win> win; double *winbuf = nullptr, *winold = nullptr; size_t winelems = 0; Win *d_win = nullptr; int *d_wing_box = nullptr;
    // field pointer / view tables of the boxes
    std::vector<FP> h_fp; FP *d_fp = nullptr; DV *d_dv = nullptr;
    // the same table with the two canvases of the head trading places: a relaxation of an odd number of launches inside a V-cycle leaves its
    // result on the second canvas and makes THAT the head (swap_head) instead of copying it back; the post-smoothing undoes it
    std::vector<FP> h_fp_alt; FP *d_fp_alt = nullptr; bool swapped = false;
    unsigned long tab_epoch = 0;                                  // suhmo_fp_epoch() the tables were last compared at
    unsigned long long ensured = 0;                               // fields every box is known to have
    int self_wrap = -1;                                            // some box of the level is its own periodic neighbour (-1: not looked at yet)
    double *d_red = nullptr; int maxnx = 0, maxny = 0;            // reduction scratch (64 nbox + 16 doubles), largest box
    // ---- owner computes (rank strips, creation option partition_min_cells): boxes own[r] .. own[r+1] belong to rank r (LoadBalance,
    // src/AmrHydro.cpp:4283, 4929).  EVERY pass over the level runs on the owner's boxes only; plans are executed by the owner of the cell
    // they write; what they read of other ranks' boxes travels as packed cells (Sync), what they write into them (averages) as packed
    // rectangles (avg_put / avg_get).  The ghost exchange before a colour pass (Copier::exchange, src/VCAMRNonLinearPoissonOp.cpp:692,
    // 912-913) moves the side cells of ONE colour of the boxes that have a neighbour on another rank and nothing else.
    bool part = false;
    std::vector<int> own, owner;                     // owner[k]
    std::vector<char> held;                          // this rank keeps storage for box k (its own, or a mirror some plan here reads)
    int b0 = 0, nown = 0;                            // this rank's boxes: b0 .. b0 + nown
    Sync sy_side[2], sy_sides, sy_all;               // cells of THIS level: sources of fine-fine side ghosts by colour / both colours / sides + corners
    Sync sy_cread, sy_win;                           // cells of level l-1 (>= 1) the stencils / the correction windows of this level's plans read here
    Sync sy_fface;                                   // cells of THIS level the reflux into level l-1 reads on the owners of the coarse cells
    DevVec<RectEnt> avg_cov; int cov_w = 0, cov_h = 0;       // covered rectangles by the owner of the COARSE cells (zeroing / marking them)
    DevVec<RectEnt> avg_put; DevVec<PutEnt> avg_get; long put_stride = 0, put_mine = 0; int put_w = 0, put_h = 0, get_w = 0, get_h = 0;
    long owned_cells = 0, held_boxes = 0;
};
}  // namespace

struct suhmo_hier {
    int nlev = 0, device = 0;
    HLev lev[8];
    suhmo_bc_t bc;
    suhmo_level_desc_t base_desc;
    suhmo_hier *gap = nullptr; double gap_dt = 0.0;        // implicit gap-height operator of the time step, owned
    // ---- level 0 cut into rank strips (one process per GPU): a rank holds its own rows of level 0 and ALL boxes of the finer
    // levels.  What level 1 reads of level 0 (coarse-fine stencils, linear fill, correction windows, reflux) comes from a
    // SHADOW: canvases with the geometry of the whole level 0, kept current only at the cells the plans read (`need`, sorted
    // by row, so the cells a rank owns are one segment); one all-gather refreshes a field (or several) before a plan runs.
    // What level 1 writes into level 0 (averages, reflux) is clipped to the rank's own rows when the plans are built.
    int rank = 0, world = 1;
    bool push_ghosts = true;                               // option push_ghosts = 0: an exchange launch before every colour pass instead
    bool shadowed = false;                                 // world > 1, or creation option shadow = 1 (tests: the whole path on one rank)
    std::string options;                                   // as given to suhmo_hier_create_opts (the gap hierarchy is created with the same)
    DV vglob;                                              // level 0 as one canvas (= the base view when it is not cut)
    FP shadow{};                                           // COMPACT: only the rows of level 0 that hold a cell some plan reads (shadow_rows of them, pitch
    size_t shadow_elems = 0; int shadow_rows = 0;          // vglob.P; a plan's offset = compact row * P + column); fields allocated on first use
    double *cover_whole = nullptr;                         // SUHMO_F_COVER of the WHOLE level 0 (geometry only; the moulin integrals run over all of it)
    DevVec<int> need, need_c; DevVec<int2> need_rl;        // offsets in vglob (what the owner packs) / in the compact shadow; (owner rank, position in its segment)
    std::vector<int> seg;                                  // need[seg[r] .. seg[r+1]) are rows of rank r
    long cnt_max = 0;                                      // longest segment: every rank contributes cnt_max doubles per field
    double *xs = nullptr, *xr = nullptr; size_t xcap = 0;  // staging of the all-gather
    suhmo_hier_allgather_fn ag = nullptr; void *ag_user = nullptr;
    long gathers = 0;
    // coarse-fine ghosts of the head of level l are current while neither level l's nor level l-1's head has been written since they
    // were interpolated: phi_ver[l] counts the writes, cf_seen[l] = the two versions the ghosts were made from
    unsigned long phi_ver[8] = {1, 1, 1, 1, 1, 1, 1, 1}, cf_seen[8][2] = {}, ff_seen[8] = {};    // ff_seen: likewise the fine-fine side ghosts
    bool phi_shadow_fresh = false;                         // the shadow's head is current: nothing has written level 0's head since its refresh
    // LPHI and RES = rhs - LPHI of level 0 were evaluated over the whole level (the composite residual of the solve loop) and, since
    // then, level 0's head has changed only where level 1 was averaged down: the next composite residual re-evaluates only the
    // rectangles lev[1].dirty0.  base_full_ver counts every other write to level 0 (head, right-hand side, coefficients: all of them
    // pass through the level-0 V-cycle or an entry point of the C-ABI)
    unsigned long base_full_ver = 1, base_res_seen = 0;
    // ... or were left behind by the launch that ended level 0's own V-cycle (suhmo_gsrb.hip, residual output): the solve loop's residual
    // evaluation then needs no pass over level 0 at all
    unsigned long base_fused_ver = 0;
    bool fused_relax = true;                               // option fused_relax: two sweeps per launch on levels of boxes (0: a launch per colour pass)
    long n_fused_relax = 0;                                // launches of that kind (read-only option fused_relax_launches)
    int box_sweeps = 4;                                    // option box_sweeps: sweeps per launch of k_gsrb_box_m (4 or 2)
    bool merged_launches = true;                           // option merged_launches: both kinds of ghost cell in one launch, one norm read-back per hierarchy, ... (0: a launch each)
    bool fused_prolong = true;                             // option fused_prolong: AMRProlongS_2 of a box in one workgroup (0: gather, BC, prolongation as three launches)
    bool incremental = true;                               // option incremental_residual
    long part_min_cells = 350000;                          // creation option partition_min_cells: when the largest level >= 1 holds at least this many cells
                                                           // PER RANK, the levels >= 1 are dealt to the ranks (below it a pass is shorter than the messages it needs)
    bool part = false;                                     // ... they are
    long part_gathers = 0, part_bytes = 0;                 // collectives of the partition; bytes THIS rank contributed to them
    long side_bytes[8] = {};                               // bytes this rank contributes to ONE colour-pass ghost exchange of level l (the larger colour)
    double *ps = nullptr, *pr = nullptr; size_t pcap = 0;  // staging of those collectives (pcap doubles per rank)
    // read-only counters (suhmo_hier_get_option): composite residuals of level 0 evaluated on the dirty rectangles only / not at all (left
    // behind by the launch that ended level 0's V-cycle), coarse gradients evaluated on the cell list only
    long n_incr_residual = 0, n_fused_residual = 0, n_sparse_grad = 0;
    double *red_all = nullptr;                             // partial maxima of a norm over all levels of boxes (64 per box + 16)
    DevVec<RectEnt> cover_full;                            // coarsen(boxes of level 1) in the shadow: COVER of the whole level 0
};

namespace {
inline bool wrap_cell(const suhmo_hier *H, const HLev &V, int &i, int &j)
{
    if (H->bc.periodic[0]) { if (i < 0) i += V.nxd; else if (i >= V.nxd) i -= V.nxd; }
    if (H->bc.periodic[1]) { if (j < 0) j += V.nyd; else if (j >= V.nyd) j -= V.nyd; }
    return i >= 0 && i < V.nxd && j >= 0 && j < V.nyd;
}
inline int owner_of(const suhmo_hier *H, const HLev &V, int i, int j)
{
    if (!wrap_cell(H, V, i, j)) return -1;
    if (V.l == 0) return 0;
    return V.index.find(i, j);
}
// canvas reference of the cell (i,j) (level indices, wrapped into the domain) in the box that holds it; b = -1 if none
inline Ref cell_ref(const suhmo_hier *H, const HLev &V, int i, int j)
{
    Ref r{-1, 0};
    if (!wrap_cell(H, V, i, j)) return r;
    int o = V.l == 0 ? 0 : V.index.find(i, j);
    if (o < 0) return r;
    const DV &v = V.l == 0 ? H->vglob : V.box[o]->d[0].v;
    r.b = o; r.off = cidx(v, i - v.i0, j - v.j0);
    return r;
}
inline bool dist_base(const suhmo_hier *H) { return H->shadowed; }
inline suhmo_level *base_of(suhmo_hier *H) { return H->lev[0].box[0]; }
inline Ref local_ref(const HLev &V, int k, int il, int jl) { return Ref{k, cidx(V.box[k]->d[0].v, il, jl)}; }

// ------------------------------------------------------------------ kernels over the plans
__device__ __forceinline__ double *fptr(const FP *tab, const FP &base, int use_base, int b, int field)
{
    return use_base ? base.f[field] : tab[b].f[field];
}
__device__ __forceinline__ void d_ff(const CopyEnt &c, const FP *__restrict__ tab, int f0, int f1)
{
    tab[c.d.b].f[f0][c.d.off] = tab[c.s.b].f[f0][c.s.off];
    if (f1 >= 0) tab[c.d.b].f[f1][c.d.off] = tab[c.s.b].f[f1][c.s.off];
}
__global__ void k_ff(const CopyEnt *__restrict__ e, int n, const FP *__restrict__ tab, int f0, int f1)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d_ff(e[t], tab, f0, f1);
}
// [Chombo] QuadCFInterp (oracle/amrm.c:cf_interp)
__device__ __forceinline__ void d_cf(const CfEnt &q, const FP *__restrict__ ftab, int ff0, const FP *__restrict__ ctab, const FP &cbase,
                                     int use_base, int fc0, int ff1, int fc1)
{
    const double c_s = 8.0 / 15.0, c_b = 2.0 / 3.0, c_a = -0.2;
    const double xt = q.xsign ? 0.25 : -0.25;
  for (int pass = 0; pass < (ff1 >= 0 ? 2 : 1); pass++) {          // one or two fields over the same stencils (the two gradient components)
    const int ff = pass ? ff1 : ff0, fc = pass ? fc1 : fc0;
#define CVAL(m) fptr(ctab, cbase, use_base, q.c[m].b, fc)[q.c[m].off]
    double c0, d1 = 0.0, d2 = 0.0;
    if (q.kind == 0) { double cm = CVAL(0), cp = CVAL(2); c0 = CVAL(1); d1 = 0.5 * (cp - cm); d2 = cp - 2.0 * c0 + cm; }
    else if (q.kind == 1) { c0 = CVAL(0); double cp = CVAL(1), cpp = CVAL(2); d1 = 0.5 * (-3.0 * c0 + 4.0 * cp - cpp); d2 = c0 - 2.0 * cp + cpp; }
    else if (q.kind == 2) { c0 = CVAL(0); double cp = CVAL(1); d1 = cp - c0; }
    else if (q.kind == 3) { c0 = CVAL(0); double cm = CVAL(1), cmm = CVAL(2); d1 = 0.5 * (3.0 * c0 - 4.0 * cm + cmm); d2 = c0 - 2.0 * cm + cmm; }
    else if (q.kind == 4) { c0 = CVAL(0); double cm = CVAL(1); d1 = c0 - cm; }
    else c0 = CVAL(0);
#undef CVAL
    double phistar = c0 + xt * d1 + (0.5 * xt * xt) * d2;
    double *f = ftab[q.f.b].f[ff];
    f[q.f.off] = c_s * phistar + c_b * f[q.f.off + q.step] + c_a * f[q.f.off + 2 * q.step];
  }
}
__global__ void k_cf(const CfEnt *__restrict__ e, int n, const FP *__restrict__ ftab, int ff0, const FP *__restrict__ ctab, FP cbase,
                     int use_base, int fc0, int ff1, int fc1)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d_cf(e[t], ftab, ff0, ctab, cbase, use_base, fc0, ff1, fc1);
}
// both kinds of ghost cell of one or two fields of a level in ONE launch: the coarse-fine ghosts are interpolated from the level below and two
// VALID cells of their own box, the fine-fine ghosts are copies of VALID cells of the neighbouring boxes -- neither reads what the other
// writes (workgroups [0, nbcf): the cf plan, the rest: the ff plan)
__global__ void k_cf_ff(const CfEnt *__restrict__ ce, int ncf, int nbcf, const CopyEnt *__restrict__ fe, int nff, const FP *__restrict__ ftab, int ff0,
                        const FP *__restrict__ ctab, FP cbase, int use_base, int fc0, int ff1, int fc1)
{
    if ((int)blockIdx.x < nbcf) {
        int t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < ncf) d_cf(ce[t], ftab, ff0, ctab, cbase, use_base, fc0, ff1, fc1);
    } else {
        int t = (blockIdx.x - nbcf) * blockDim.x + threadIdx.x;
        if (t < nff) d_ff(fe[t], ftab, ff0, ff1);
    }
}
// ... of the head of SEVERAL levels (their ghosts depend on valid cells only, of the level itself and of the one below: no order among them).
// By value, indexed with constants only (unrolled search), so that the tables stay in scalar registers.
struct LvGhosts { const CfEnt *cf[SUHMO_LVMAX]; const CopyEnt *ff[SUHMO_LVMAX]; const FP *ftab[SUHMO_LVMAX], *ctab[SUHMO_LVMAX];
                  int ncf[SUHMO_LVMAX], nbcf[SUHMO_LVMAX], nff[SUHMO_LVMAX], nb[SUHMO_LVMAX], use_base[SUHMO_LVMAX]; int n; };
__global__ void k_cf_ff_lv(LvGhosts lv, FP cbase, int field)
{
    int b = blockIdx.x, q = -1;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t < lv.n && q < 0) { if (b < lv.nb[t]) q = t; else b -= lv.nb[t]; }
    if (q < 0) return;
    const CfEnt *ce = nullptr; const CopyEnt *fe = nullptr; const FP *ftab = nullptr, *ctab = nullptr; int ncf = 0, nbcf = 0, nff = 0, use_base = 0;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t == q) { ce = lv.cf[t]; fe = lv.ff[t]; ftab = lv.ftab[t]; ctab = lv.ctab[t]; ncf = lv.ncf[t]; nbcf = lv.nbcf[t]; nff = lv.nff[t]; use_base = lv.use_base[t]; }
    if (b < nbcf) {
        int t = b * blockDim.x + threadIdx.x;
        if (t < ncf) d_cf(ce[t], ftab, field, ctab, cbase, use_base, field, -1, -1);
    } else {
        int t = (b - nbcf) * blockDim.x + threadIdx.x;
        if (t < nff) d_ff(fe[t], ftab, field, -1);
    }
}
// [Chombo] PiecewiseLinearFillPatch (oracle/amr_step.c:or_pwl_fill)
__global__ void k_pwl(const PwlEnt *__restrict__ e, int n, const FP *__restrict__ ftab, int ff, const FP *__restrict__ ctab, FP cbase,
                      int use_base, int fc)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    PwlEnt q = e[t];
#define CVAL(m) fptr(ctab, cbase, use_base, q.c[m].b, fc)[q.c[m].off]
    const double c0 = CVAL(4);
    double s0, s1;
    if (q.sx == 0) s0 = 0.5 * (CVAL(5) - CVAL(3)); else if (q.sx == 1) s0 = CVAL(5) - c0; else s0 = c0 - CVAL(3);
    if (q.sy == 0) s1 = 0.5 * (CVAL(7) - CVAL(1)); else if (q.sy == 1) s1 = CVAL(7) - c0; else s1 = c0 - CVAL(1);
    double smax = c0, smin = c0;
    for (int m = 0; m < 9; m++) {
        if (q.c[m].b < 0) continue;
        double v = CVAL(m);
        smax = fmax(smax, v); smin = fmin(smin, v);
    }
#undef CVAL
    const double deltasum = 0.5 * (fabs(s0) + fabs(s1));
    if (deltasum > 0.0) {
        double etamax = (smax - c0) / deltasum, etamin = (c0 - smin) / deltasum;
        double eta = fmax(fmin(fmin(etamin, etamax), 1.0), 0.0);
        s0 = eta * s0; s1 = eta * s1;
    }
    double v = c0;
    v = v + s0 * ((q.par & 1) ? 0.25 : -0.25);
    v = v + s1 * ((q.par & 2) ? 0.25 : -0.25);
    ftab[q.f.b].f[ff][q.f.off] = v;
}
// [Chombo] FORT_AVERAGE (mode 0) / covered cells <- val (mode 1)
__device__ __forceinline__ void d_avg(const RectEnt &q, int I, int J, const FP *__restrict__ ftab, const DV *__restrict__ fdv, int ff,
                                      const FP *__restrict__ ctab, const DV *__restrict__ cdv, const FP &cbase, const DV &cbdv, int use_base, int fc, int mode, double val)
{
    if (I >= q.w || J >= q.h) return;
    const int Pc = use_base ? cbdv.P : cdv[q.cb].P;
    double *c = fptr(ctab, cbase, use_base, q.cb, fc);
    if (mode == 1) { c[q.coff + J * Pc + I] = val; return; }
    const int Pf = fdv[q.fb].P;
    const double *f = ftab[q.fb].f[ff];
    int b = q.foff + 2 * J * Pf + 2 * I;
    double s = 0.0;
    s = s + f[b]; s = s + f[b + 1]; s = s + f[b + Pf]; s = s + f[b + Pf + 1];
    c[q.coff + J * Pc + I] = s * 0.25;
}
__global__ void k_avg(const RectEnt *__restrict__ e, const FP *__restrict__ ftab, const DV *__restrict__ fdv, int ff,
                      const FP *__restrict__ ctab, const DV *__restrict__ cdv, FP cbase, DV cbdv, int use_base, int fc, int mode, double val)
{
    d_avg(e[blockIdx.z], blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y, ftab, fdv, ff, ctab, cdv, cbase, cbdv, use_base, fc, mode, val);
}
// owner computes: the same averages into this rank's segment of an all-gather (q.coff = position, pitch = q.w) ...
__global__ void k_avg_put(const RectEnt *__restrict__ e, const FP *__restrict__ ftab, const DV *__restrict__ fdv, int ff, double *__restrict__ buf)
{
    RectEnt q = e[blockIdx.z];
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= q.w || J >= q.h) return;
    const int Pf = fdv[q.fb].P;
    const double *f = ftab[q.fb].f[ff];
    int b = q.foff + 2 * J * Pf + 2 * I;
    double s = 0.0;
    s = s + f[b]; s = s + f[b + 1]; s = s + f[b + Pf]; s = s + f[b + Pf + 1];
    buf[q.coff + (long)J * q.w + I] = s * 0.25;
}
// ... and the holder of the coarse cells takes its rectangles out of the writers' segments
__global__ void k_put_unpack(const PutEnt *__restrict__ e, const FP *__restrict__ ctab, const DV *__restrict__ cdv, FP cbase, DV cbdv, int use_base, int fc,
                             const double *__restrict__ buf, long stride)
{
    PutEnt q = e[blockIdx.z];
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= q.w || J >= q.h) return;
    const int Pc = use_base ? cbdv.P : cdv[q.cb].P;
    double *c = fptr(ctab, cbase, use_base, q.cb, fc);
    c[q.coff + J * Pc + I] = buf[(long)q.rank * stride + q.pos + (long)J * q.w + I];
}
// old != NULL: the window gets c - old (the correction phi - phi_saved, as axby(phi, saved, 1, -1) states it), old being an earlier
// gather of the same cells
__global__ void k_win_gather(const WinEnt *__restrict__ e, double *__restrict__ wbuf, const FP *__restrict__ ctab, const DV *__restrict__ cdv,
                             FP cbase, DV cbdv, int use_base, int fc, const Win *__restrict__ wins, const int *__restrict__ went_box,
                             const double *__restrict__ old = nullptr)
{
    WinEnt q = e[blockIdx.z];
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= q.w || J >= q.h) return;
    const int Pc = use_base ? cbdv.P : cdv[q.cb].P;
    const double *c = fptr(ctab, cbase, use_base, q.cb, fc);
    const Win w = wins[went_box[blockIdx.z]];
    const size_t o = w.base + q.woff + (size_t)J * w.nx + I;
    const double cv = c[q.coff + J * Pc + I];
    wbuf[o] = old ? 1.0 * cv + -1.0 * old[o] : cv;
}
// physical BC of the coarse level on the window of every fine box (m_bc on a_temp, AMRProlongS_2 :1160-1166), along the
// coarsened box's own extent only: the corner cells beyond it are never written (value 0)
__global__ void k_win_bc(const Win *__restrict__ wins, int nwin, double *__restrict__ wbuf, DV cv /* a view of level l-1: BC data, domain size */)
{
    int k = blockIdx.y;
    if (k >= nwin) return;
    const Win w = wins[k];
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int inx = w.nx - 2, iny = w.ny - 2;             // the coarsened box itself
    double *p = wbuf + w.base;
    int dir, side, tt;
    if (t < 2 * iny) { dir = 0; side = t / iny; tt = t % iny; }
    else { t -= 2 * iny; if (t >= 2 * inx) return; dir = 1; side = t / inx; tt = t % inx; }
    if (cv.per[dir]) return;
    const int ndom = dir == 0 ? cv.nxg : cv.nyg;
    const int g = dir == 0 ? (side ? w.i0 + w.nx - 1 : w.i0) : (side ? w.j0 + w.ny - 1 : w.j0);   // global index of the ghost layer
    if (g >= 0 && g <= ndom - 1) return;
    const int il = dir == 0 ? (side ? w.nx - 1 : 0) : tt + 1, jl = dir == 0 ? tt + 1 : (side ? w.ny - 1 : 0);
    const int in_ = dir == 0 ? (side ? w.nx - 2 : 1) : il, jn_ = dir == 0 ? jl : (side ? w.ny - 2 : 1);
    const double nearv = p[(size_t)jn_ * w.nx + in_];
    double gv;
    if (cv.bct[dir][side] == 0) gv = cv.two_v[dir][side] - nearv; else gv = nearv + cv.neu[dir][side];
    p[(size_t)jl * w.nx + il] = gv;
}
// PROLONG_2_NL (src/AMRNonLinearPoissonOpF.ChF:660-705) of every box of a level from its window
__global__ void k_prolong2_win(const Win *__restrict__ wins, const double *__restrict__ wbuf, const FP *__restrict__ ftab, const DV *__restrict__ fdv)
{
    const int k = blockIdx.z;
    const DV v = fdv[k];
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    const Win w = wins[k];
    const double *c = wbuf + w.base;
    const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
    int gi = i + v.i0, gj = j + v.j0;
    int ic = gi / 2, jc = gj / 2, o1 = 2 * (gi % 2) - 1, o2 = 2 * (gj % 2) - 1;
    int cc = (jc - w.j0) * w.nx + (ic - w.i0);
    double *phi = ftab[k].f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    double p = phi[idx];
    p = p + fx2 * c[cc] + f0 * c[cc + o1 + o2 * w.nx];
    p = p + fx1 * (c[cc + o1] + c[cc + o2 * w.nx]);
    phi[idx] = p;
}
// AMRProlongS_2 of one box per workgroup: the three steps above (gather of the coarse correction into the box's window, physical BC on the
// window, PROLONG_2_NL) with the window in LDS instead of three launches over a buffer in HBM; the same expressions on the same operands.
// wstart[k] .. wstart[k + 1]: the gather pieces of box k.  old != NULL: the window gets c - old (see k_win_gather)
// fc_minus >= 0: the coarse field is 1 fc + (-1) fc_minus formed on the fly (the correction PHI - PHIOLD of a level of boxes leaving its FAS problem),
// and the workgroups from nk on ARE that leaving (k_fas_leave_m's RHS <- RHS0, CORR <- PHI - PHIOLD on the coarse level's boxes: they write
// neither PHI nor PHIOLD): one launch instead of two
__global__ __launch_bounds__(256) void k_prolong2_fused(const WinEnt *__restrict__ e, const int *__restrict__ wstart, const Win *__restrict__ wins, int k0,
                                                        const FP *__restrict__ ctab, const DV *__restrict__ cdv, FP cbase, DV cbdv, int use_base, int fc,
                                                        const double *__restrict__ old, const FP *__restrict__ ftab, const DV *__restrict__ fdv,
                                                        int fc_minus, int nk, int cgx, int cgy)
{
    extern __shared__ double win[];
    if (fc_minus >= 0 && (int)blockIdx.x >= nk) {
        const int b = blockIdx.x - nk, bx = b % cgx, by = (b / cgx) % cgy, bz = b / (cgx * cgy);
        const DV &v = cdv[bz];
        const FP &f = ctab[bz];
        const int i = bx * 64 + (int)(threadIdx.x & 63) - 1, j = by * 4 + (int)(threadIdx.x >> 6) - 1;
        if (i > v.nx || j > v.ny) return;
        const int idx = cidx(v, i, j);
        f.f[SUHMO_F_RHS][idx] = f.f[SUHMO_F_RHS0][idx];
        if (i >= 0 && i < v.nx && j >= 0 && j < v.ny) f.f[SUHMO_F_CORR][idx] = 1.0 * f.f[SUHMO_F_PHI][idx] + -1.0 * f.f[SUHMO_F_PHIOLD][idx];
        return;
    }
    const int k = k0 + blockIdx.x, tid = threadIdx.x;
    const Win w = wins[k];
    const int nw = w.nx * w.ny;
    for (int t = tid; t < nw; t += 256) win[t] = 0.0;                       // (cells no piece and no BC writes: the corners, value 0)
    __syncthreads();
    for (int p = wstart[k]; p < wstart[k + 1]; p++) {
        const WinEnt q = e[p];
        const int Pc = use_base ? cbdv.P : cdv[q.cb].P;
        const double *c = fptr(ctab, cbase, use_base, q.cb, fc);
        const double *cm = fc_minus >= 0 ? fptr(ctab, cbase, use_base, q.cb, fc_minus) : nullptr;
        for (int t = tid; t < q.w * q.h; t += 256) {
            const int J = t / q.w, I = t - J * q.w;
            const int o = q.woff + J * w.nx + I;
            double cv = c[q.coff + J * Pc + I];
            if (cm) cv = 1.0 * cv + -1.0 * cm[q.coff + J * Pc + I];
            win[o] = old ? 1.0 * cv + -1.0 * old[w.base + o] : cv;
        }
    }
    __syncthreads();
    {   // k_win_bc
        const int inx = w.nx - 2, iny = w.ny - 2;
        for (int t0 = tid; t0 < 2 * iny + 2 * inx; t0 += 256) {
            int t = t0, dir, side, tt;
            if (t < 2 * iny) { dir = 0; side = t / iny; tt = t % iny; }
            else { t -= 2 * iny; dir = 1; side = t / inx; tt = t % inx; }
            if (cbdv.per[dir]) continue;
            const int ndom = dir == 0 ? cbdv.nxg : cbdv.nyg;
            const int g = dir == 0 ? (side ? w.i0 + w.nx - 1 : w.i0) : (side ? w.j0 + w.ny - 1 : w.j0);
            if (g >= 0 && g <= ndom - 1) continue;
            const int il = dir == 0 ? (side ? w.nx - 1 : 0) : tt + 1, jl = dir == 0 ? tt + 1 : (side ? w.ny - 1 : 0);
            const int in_ = dir == 0 ? (side ? w.nx - 2 : 1) : il, jn_ = dir == 0 ? jl : (side ? w.ny - 2 : 1);
            const double nearv = win[jn_ * w.nx + in_];
            double gv;
            if (cbdv.bct[dir][side] == 0) gv = cbdv.two_v[dir][side] - nearv; else gv = nearv + cbdv.neu[dir][side];
            win[jl * w.nx + il] = gv;
        }
    }
    __syncthreads();
    {   // k_prolong2_win
        const DV v = fdv[k];
        const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
        double *phi = ftab[k].f[SUHMO_F_PHI];
        for (int t = tid; t < v.nx * v.ny; t += 256) {
            const int j = t / v.nx, i = t - j * v.nx;
            const int gi = i + v.i0, gj = j + v.j0;
            const int ic = gi / 2, jc = gj / 2, o1 = 2 * (gi % 2) - 1, o2 = 2 * (gj % 2) - 1;
            const int cc = (jc - w.j0) * w.nx + (ic - w.i0);
            const int idx = cidx(v, i, j);
            double p = phi[idx];
            p = p + fx2 * win[cc] + f0 * win[cc + o1 + o2 * w.nx];
            p = p + fx1 * (win[cc + o1] + win[cc + o2 * w.nx]);
            phi[idx] = p;
        }
    }
}
// [Chombo] LevelFluxRegister (oracle/amrm.c:reflux): one thread per coarse cell next to coarse-fine faces
__device__ __forceinline__ void d_reflux(const Target &T, const Face *__restrict__ faces, const FP *__restrict__ ftab, const DV *__restrict__ fdv,
                                         const FP *__restrict__ ctab, const FP &cbase, const FP &cdst, int use_base, int field_c, double dxc, double dyc, double beta, int residual)
{
    // cdst: the level itself; cbase: where its cells are read (the shadow of a cut level 0).  residual: the register is added to
    // LPHI's value and field_c <- rhs - that (the axby of the composite residual, for the cells the reflux reaches)
    double *lof = fptr(ctab, cdst, use_base, T.t.b, field_c);
    const double rscale = 1.0 / (dxc * dyc);
    double acc = residual ? fptr(ctab, cdst, use_base, T.t.b, SUHMO_F_LPHI)[T.t.off] : lof[T.t.off];
    for (int m = 0; m < T.count; m++) {
        Face f = faces[T.first + m];
        const double dxd = f.dir == 0 ? dxc : dyc, tsize = f.dir == 0 ? dyc : dxc;
        const double cs = beta * 1 / dxd, fs = beta * 2 / dxd;
        const double sign = f.side == 0 ? 1.0 : -1.0;
        double phihi = fptr(ctab, cbase, use_base, f.hi.b, SUHMO_F_PHI)[f.hi.off], philo = fptr(ctab, cbase, use_base, f.lo.b, SUHMO_F_PHI)[f.lo.off];
        double bc_ = fptr(ctab, cbase, use_base, f.bq.b, f.dir == 0 ? SUHMO_F_BX : SUHMO_F_BY)[f.bq.off];
        double Fc = -bc_ * ((phihi - philo) * cs);
        double reg = -(tsize * Fc);
        const double *phif = ftab[f.fb].f[SUHMO_F_PHI], *bf = ftab[f.fb].f[f.dir == 0 ? SUHMO_F_BX : SUHMO_F_BY];
        const int Pf = fdv[f.fb].P;
        for (int k = 0; k < 2; k++) {
            int idx = f.foff + (f.dir == 0 ? k * Pf : k);
            double ph_hi = phif[idx], ph_lo = f.dir == 0 ? phif[idx - 1] : phif[idx - Pf];
            double Ff = -bf[idx] * ((ph_hi - ph_lo) * fs);
            reg = reg + (tsize * Ff) * 0.5;
        }
        acc = acc + sign * rscale * reg;
    }
    lof[T.t.off] = residual ? -1.0 * acc + 1.0 * fptr(ctab, cdst, use_base, T.t.b, SUHMO_F_RHS)[T.t.off] : acc;
}
__global__ void k_reflux(const Target *__restrict__ tg, int n, const Face *__restrict__ faces, const FP *__restrict__ ftab, const DV *__restrict__ fdv,
                         const FP *__restrict__ ctab, FP cbase, FP cdst, int use_base, int field_c, double dxc, double dyc, double beta, int residual)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    d_reflux(tg[t], faces, ftab, fdv, ctab, cbase, cdst, use_base, field_c, dxc, dyc, beta, residual);
}
// the refluxes of SEVERAL levels (level l's adds to cells of level l-1 from fluxes of levels l and l-1: no order among them)
struct LvReflux { const Target *tg[SUHMO_LVMAX]; const Face *faces[SUHMO_LVMAX]; const FP *ftab[SUHMO_LVMAX], *ctab[SUHMO_LVMAX]; const DV *fdv[SUHMO_LVMAX];
                  int ntg[SUHMO_LVMAX], nb[SUHMO_LVMAX], use_base[SUHMO_LVMAX]; double dxc[SUHMO_LVMAX], dyc[SUHMO_LVMAX], beta[SUHMO_LVMAX]; int n; };
// ... and, in the workgroups from nb0 on, ONE level's FORT_AVERAGE of the same field onto the cells it covers (the step that follows the reflux in a
// V-cycle's down-leg: it reads the fine residual and writes COVERED coarse cells, the reflux writes uncovered ones next to the coarse-fine faces)
struct AvgPart { const RectEnt *e; const FP *ftab, *ctab; const DV *fdv, *cdv; int use_base, n, gx, gy, nb0; };
__global__ void k_reflux_lv(LvReflux lv, FP cbase, FP cdst, int field_c, int residual, AvgPart av, DV cbdv)
{
    if (av.n > 0 && (int)blockIdx.x >= av.nb0) {
        const int b = blockIdx.x - av.nb0, bx = b % av.gx, by = (b / av.gx) % av.gy, bz = b / (av.gx * av.gy);
        d_avg(av.e[bz], bx * 64 + (int)(threadIdx.x & 63), by * 4 + (int)(threadIdx.x >> 6), av.ftab, av.fdv, field_c, av.ctab, av.cdv, cdst, cbdv, av.use_base, field_c, 0, 0.0);
        return;
    }
    int b = blockIdx.x, q = -1;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t < lv.n && q < 0) { if (b < lv.nb[t]) q = t; else b -= lv.nb[t]; }
    if (q < 0) return;
    const Target *tg = nullptr; const Face *faces = nullptr; const FP *ftab = nullptr, *ctab = nullptr; const DV *fdv = nullptr;
    int ntg = 0, use_base = 0; double dxc = 0.0, dyc = 0.0, beta = 0.0;
#pragma unroll
    for (int t = 0; t < SUHMO_LVMAX; t++)
        if (t == q) { tg = lv.tg[t]; faces = lv.faces[t]; ftab = lv.ftab[t]; ctab = lv.ctab[t]; fdv = lv.fdv[t]; ntg = lv.ntg[t]; use_base = lv.use_base[t];
                      dxc = lv.dxc[t]; dyc = lv.dyc[t]; beta = lv.beta[t]; }
    const int t = b * blockDim.x + threadIdx.x;
    if (t < ntg) d_reflux(tg[t], faces, ftab, fdv, ctab, cbase, cdst, use_base, field_c, dxc, dyc, beta, residual);
}

// ------------------------------------------------------------------ plan building (host)
// transfers (owner, cell, reader) of one kind -> this rank's part of the exchange: the cells it packs (what anybody reads of its boxes, sorted:
// position = rank in that order), the cells it unpacks into its mirrors of V's boxes (marked as held), the longest segment
int make_sync(suhmo_hier *H, HLev &V, std::vector<Xf> &x, Sync &S)
{
    const int me = H->rank;
    auto key = [](const Xf &a, const Xf &b) { return a.owner != b.owner ? a.owner < b.owner : a.b != b.b ? a.b < b.b : a.off != b.off ? a.off < b.off : a.reader < b.reader; };
    std::sort(x.begin(), x.end(), key);
    x.erase(std::unique(x.begin(), x.end(), [](const Xf &a, const Xf &b) { return a.owner == b.owner && a.b == b.b && a.off == b.off && a.reader == b.reader; }), x.end());
    std::vector<Ref> send; std::vector<SyncRecv> recv;
    std::vector<long> cnt(H->world, 0);
    for (size_t t = 0; t < x.size();) {                     // one cell of one owner, its readers
        size_t u = t;
        bool mine = false;
        while (u < x.size() && x[u].owner == x[t].owner && x[u].b == x[t].b && x[u].off == x[t].off) { mine = mine || x[u].reader == me; u++; }
        const int pos = (int)cnt[x[t].owner]++;
        if (x[t].owner == me) send.push_back(Ref{x[t].b, x[t].off});
        else if (mine) { recv.push_back(SyncRecv{Ref{x[t].b, x[t].off}, x[t].owner, pos}); V.held[x[t].b] = 1; }
        t = u;
    }
    S.stride = *std::max_element(cnt.begin(), cnt.end());
    return S.send.upload(send) | S.recv.upload(recv);
}
int build_plans(suhmo_hier *H, int l)
{
    HLev &F = H->lev[l], &C = H->lev[l - 1];
    const int nb = (int)F.box.size();
    std::vector<CopyEnt> ffs, ffc;
    std::vector<CfEnt> cf;
    std::vector<PwlEnt> pwl;
    std::vector<RectEnt> avg;
    std::vector<WinEnt> wing; std::vector<int> wing_box;
    F.win.resize(nb);
    size_t wtot = 0;
    // level 0 cut into rank strips: its cells are read through the shadow (offsets in H->vglob, collected in `needv`) and
    // written in the rank's own rows only
    const bool cut = C.l == 0 && dist_base(H);
    const DV sv = C.l == 0 ? base_of(H)->d[0].v : DV{};
    std::vector<int> needv;
    std::vector<int4> dirty0; std::vector<std::pair<int, int>> gcell;        // (level l == 1)
    std::vector<RectEnt> cover_full;
    auto note = [&](const Ref &r) { if (cut && r.b >= 0) needv.push_back(r.off); };
    // owner computes: who executes what.  ownF / ownC: the rank that holds box k of this level / box o of level l-1 (a replicated level:
    // every rank, i.e. "me"); a cell of a cut level 0 belongs to the strip its row lies in, and is READ through the shadow
    const bool P = F.part;
    const int me = H->rank;
    auto ownF = [&](int k) { return P ? F.owner[k] : me; };
    auto ownC = [&](int o) { return (P && C.l >= 1) ? C.owner[o] : me; };
    std::vector<Xf> x_side[2], x_all, x_cread, x_win, x_fface;
    std::vector<RectEnt> avg_cov, avg_put; std::vector<PutEnt> avg_get;
    std::vector<long> putpos(H->world, 0);
    auto good_cell = [&](int I, int J) -> bool {          // coarse cell (I,J) of level l-1 good for tangential stencils?
        if (!wrap_cell(H, C, I, J)) return false;
        return F.index.find(2 * I, 2 * J) < 0;
    };
    for (int k = 0; k < nb; k++) {
        const int *b = &F.b4[4 * k];
        const DV &v = F.box[k]->d[0].v;
        // ---- ghost ring: fine-fine copies, coarse-fine interpolation entries, linear fill entries
        for (int j = b[1] - 1; j <= b[3] + 1; j++)
            for (int i = b[0] - 1; i <= b[2] + 1; i++) {
                const bool gx = i < b[0] || i > b[2], gy = j < b[1] || j > b[3];
                if (!gx && !gy) continue;
                int iw = i, jw = j;
                if (!wrap_cell(H, F, iw, jw)) continue;                       // domain ghost
                const Ref mine = local_ref(F, k, i - b[0], j - b[1]);
                const int o = F.index.find(iw, jw);
                if (o >= 0) {
                    const DV &vo = F.box[o]->d[0].v;
                    CopyEnt e{mine, Ref{o, cidx(vo, iw - vo.i0, jw - vo.j0)}};
                    if (P && F.owner[o] != F.owner[k]) {                      // the source cell travels to the owner of the ghost
                        const Xf x{F.owner[o], e.s.b, e.s.off, F.owner[k]};
                        x_all.push_back(x);
                        if (!(gx && gy)) x_side[(iw + jw) & 1].push_back(x);
                    }
                    if (ownF(k) == me) (gx && gy ? ffc : ffs).push_back(e);
                    continue;
                }
                // coarse-fine cell
                {
                    PwlEnt p;
                    p.f = mine; p.par = (iw & 1) | ((jw & 1) << 1);
                    const int I = iw >> 1, J = jw >> 1;
                    for (int jj = -1; jj <= 1; jj++)
                        for (int ii = -1; ii <= 1; ii++) {
                            int In = I + ii, Jn = J + jj;
                            Ref r{-1, 0};
                            if (In >= 0 && In <= C.nxd - 1 && Jn >= 0 && Jn <= C.nyd - 1) {     // as or_pwl_fill: no periodic images
                                r = cell_ref(H, C, In, Jn);
                                if (r.b < 0) { suhmo_set_error("hier: level %d is not properly nested in level %d (linear fill stencil)", l, l - 1); return -1; }
                            }
                            p.c[(jj + 1) * 3 + (ii + 1)] = r;
                            note(r);
                            if (P && C.l >= 1 && r.b >= 0 && C.owner[r.b] != F.owner[k]) x_cread.push_back(Xf{C.owner[r.b], r.b, r.off, F.owner[k]});
                        }
                    p.sx = (I - 1 >= 0 && I + 1 <= C.nxd - 1) ? 0 : (I - 1 < 0 ? 1 : 2);
                    p.sy = (J - 1 >= 0 && J + 1 <= C.nyd - 1) ? 0 : (J - 1 < 0 ? 1 : 2);
                    if (ownF(k) == me) pwl.push_back(p);
                }
                if (gx && gy) continue;                                       // QuadCFInterp: sides only
                const int dir = gx ? 0 : 1, side = gx ? (i < b[0] ? 0 : 1) : (j < b[1] ? 0 : 1);
                const int g = dir == 0 ? i : j, t = dir == 0 ? j : i;
                CfEnt e;
                e.f = mine; e.step = (side == 0 ? 1 : -1) * (dir == 0 ? 1 : v.P);
                e.xsign = t & 1;
                const int icn = g >> 1, ict = t >> 1;
                auto good = [&](int o_) { return dir == 0 ? good_cell(icn, ict + o_) : good_cell(ict + o_, icn); };
                auto cref = [&](int o_) { return dir == 0 ? cell_ref(H, C, icn, ict + o_) : cell_ref(H, C, ict + o_, icn); };
                const bool lo = good(-1), hi = good(1);
                int need[3] = {0, 0, 0}, nneed = 1;
                if (lo && hi) { e.kind = 0; need[0] = -1; need[1] = 0; need[2] = 1; nneed = 3; }
                else if (hi) { if (good(2)) { e.kind = 1; need[1] = 1; need[2] = 2; nneed = 3; } else { e.kind = 2; need[1] = 1; nneed = 2; } }
                else if (lo) { if (good(-2)) { e.kind = 3; need[1] = -1; need[2] = -2; nneed = 3; } else { e.kind = 4; need[1] = -1; nneed = 2; } }
                else e.kind = 5;
                for (int m = 0; m < 3; m++) {
                    e.c[m] = m < nneed ? cref(need[m]) : Ref{0, 0};
                    if (m < nneed && e.c[m].b < 0) { suhmo_set_error("hier: level %d is not properly nested in level %d (coarse-fine stencil)", l, l - 1); return -1; }
                    if (m < nneed) note(e.c[m]);
                    if (m < nneed && P && C.l >= 1 && C.owner[e.c[m].b] != F.owner[k]) x_cread.push_back(Xf{C.owner[e.c[m].b], e.c[m].b, e.c[m].off, F.owner[k]});
                    if (m < nneed && C.l == 0) {                              // the coarse cell, wrapped into the domain (as cell_ref did)
                        int I = dir == 0 ? icn : ict + need[m], J = dir == 0 ? ict + need[m] : icn;
                        (void)wrap_cell(H, C, I, J);
                        gcell.push_back(std::make_pair(J, I));
                    }
                }
                if (ownF(k) == me) cf.push_back(e);
            }
        // ---- average / covered rectangles: coarsen(box) split over the boxes of level l-1
        const int ci0 = b[0] / 2, cj0 = b[1] / 2, ci1 = b[2] / 2, cj1 = b[3] / 2;
        auto split = [&](int I0, int J0, int I1, int J1, auto &&emit) -> int {     // region inside the domain
            if (C.l == 0) { emit(0, I0, J0, I1, J1); return 0; }
            long cells = 0;
            for (int by = J0 / C.index.bs; by <= J1 / C.index.bs; by++)
                for (int bx = I0 / C.index.bs; bx <= I1 / C.index.bs; bx++) {
                    size_t q = (size_t)by * C.index.nbx + bx;
                    for (int p = C.index.start[q]; p < C.index.start[q + 1]; p++) {
                        const int o = C.index.items[p];
                        const int *cb = &C.b4[4 * o];
                        // each box once: only from the bucket that holds the corner of the intersection
                        int a0 = std::max(I0, cb[0]), a1 = std::min(I1, cb[2]), c0 = std::max(J0, cb[1]), c1 = std::min(J1, cb[3]);
                        if (a0 > a1 || c0 > c1) continue;
                        if (a0 / C.index.bs != bx || c0 / C.index.bs != by) continue;
                        emit(o, a0, c0, a1, c1);
                        cells += (long)(a1 - a0 + 1) * (c1 - c0 + 1);
                    }
                }
            if (cells != (long)(I1 - I0 + 1) * (J1 - J0 + 1)) { suhmo_set_error("hier: level %d is not nested in level %d", l, l - 1); return -1; }
            return 0;
        };
        int rc = split(ci0, cj0, ci1, cj1, [&](int o, int a0, int c0, int a1, int c1) {
            if (cut) cover_full.push_back(RectEnt{k, 0, 0, cidx(H->vglob, a0, c0), a1 - a0 + 1, c1 - c0 + 1});
            const DV &vc = C.box[o]->d[0].v;
            // the piece by the rank that holds its coarse cells: a strip of a cut level 0 (its rows), the owner of coarse box o, or everybody
            const int r0 = (C.l == 0 && cut) ? c0 / sv.ny : 0, r1 = (C.l == 0 && cut) ? c1 / sv.ny : 0;
            for (int r = r0; r <= r1; r++) {
                int d0 = c0, d1 = c1, dest = ownC(o);
                if (C.l == 0 && cut) { d0 = std::max(c0, r * sv.ny); d1 = std::min(c1, r * sv.ny + sv.ny - 1); dest = r; }
                if (d0 > d1) continue;
                const int w = a1 - a0 + 1, h = d1 - d0 + 1, writer = ownF(k);
                const int foff = cidx(v, 2 * a0 - b[0], 2 * d0 - b[1]), coff = cidx(vc, a0 - vc.i0, d0 - vc.j0);
                if (dest == me) avg_cov.push_back(RectEnt{k, o, foff, coff, w, h});
                if (!P) { if (dest == me) avg.push_back(RectEnt{k, o, foff, coff, w, h}); continue; }      // a replicated level: every rank averages into what it holds
                if (writer == dest) { if (writer == me) avg.push_back(RectEnt{k, o, foff, coff, w, h}); continue; }
                // the owner of the fine box averages into its segment of an all-gather, the holder of the coarse cells takes them from there
                if (writer == me) avg_put.push_back(RectEnt{k, 0, foff, (int)putpos[writer], w, h});
                if (dest == me) avg_get.push_back(PutEnt{o, coff, writer, (int)putpos[writer], w, h});
                putpos[writer] += (long)w * h;
            }
        });
        if (rc) return rc;
        // ---- window: coarsen(box) grown by one cell, gathered from level l-1 (periodic images included)
        Win &w = F.win[k];
        w.i0 = ci0 - 1; w.j0 = cj0 - 1; w.nx = ci1 - ci0 + 3; w.ny = cj1 - cj0 + 3; w.base = wtot;
        if (ownF(k) == me) wtot += (size_t)w.nx * w.ny;                      // (only the windows of this rank's boxes exist)
        for (int sy = -1; sy <= 1; sy++)
            for (int sx = -1; sx <= 1; sx++) {
                if ((sx && !H->bc.periodic[0]) || (sy && !H->bc.periodic[1])) continue;
                // window cells [w.i0 .. ] that are images (shifted by sx nxd, sy nyd) of domain cells
                int I0 = std::max(w.i0, sx * C.nxd), I1 = std::min(w.i0 + w.nx - 1, sx * C.nxd + C.nxd - 1);
                int J0 = std::max(w.j0, sy * C.nyd), J1 = std::min(w.j0 + w.ny - 1, sy * C.nyd + C.nyd - 1);
                if (I0 > I1 || J0 > J1) continue;
                rc = split(I0 - sx * C.nxd, J0 - sy * C.nyd, I1 - sx * C.nxd, J1 - sy * C.nyd, [&](int o, int a0, int c0, int a1, int c1) {
                    const DV &vc = C.l == 0 ? H->vglob : C.box[o]->d[0].v;
                    if (C.l == 0) {                                           // the part of this window piece in this rank's rows
                        const int d0 = std::max(c0, sv.j0), d1 = std::min(c1, sv.j0 + sv.ny - 1);
                        if (d0 <= d1) dirty0.push_back(int4{a0, d0 - sv.j0, a1 - a0 + 1, d1 - d0 + 1});
                    }
                    if (cut) for (int J = c0; J <= c1; J++) for (int I = a0; I <= a1; I++) needv.push_back(cidx(vc, I, J));
                    if (P && C.l >= 1 && C.owner[o] != F.owner[k])
                        for (int J = c0; J <= c1; J++) for (int I = a0; I <= a1; I++) x_win.push_back(Xf{C.owner[o], o, cidx(vc, I - vc.i0, J - vc.j0), F.owner[k]});
                    if (ownF(k) != me) return;
                    wing.push_back(WinEnt{o, cidx(vc, a0 - vc.i0, c0 - vc.j0), (c0 + sy * C.nyd - w.j0) * w.nx + (a0 + sx * C.nxd - w.i0), a1 - a0 + 1, c1 - c0 + 1});
                    wing_box.push_back(k);
                });
                if (rc) return rc;
            }
    }
    // ---- reflux: faces grouped by the coarse cell they feed, in the order (fine box, direction, side)
    std::map<std::pair<int, int>, std::vector<Face>> by_target;
    std::vector<std::pair<int, int>> order;
    for (int k = 0; k < nb; k++) {
        const int *b = &F.b4[4 * k];
        const DV &v = F.box[k]->d[0].v;
        const int ci0 = b[0] / 2, cj0 = b[1] / 2, ci1 = b[2] / 2, cj1 = b[3] / 2;
        for (int dir = 0; dir < 2; dir++) {
            const int ndomc = dir == 0 ? C.nxd : C.nyd;
            for (int side = 0; side < 2; side++) {
                const int Fc = dir == 0 ? (side == 0 ? ci0 : ci1 + 1) : (side == 0 ? cj0 : cj1 + 1);
                const int outside = side == 0 ? Fc - 1 : Fc;
                if ((outside < 0 || outside > ndomc - 1) && !H->bc.periodic[dir]) continue;
                const int tlo = dir == 0 ? cj0 : ci0, thi = dir == 0 ? cj1 : ci1;
                for (int T = tlo; T <= thi; T++) {
                    const int oi = dir == 0 ? outside : T, oj = dir == 0 ? T : outside;
                    if (owner_of(H, F, 2 * oi, 2 * oj) >= 0) continue;                 // fine-fine side
                    Face f;
                    f.dir = dir; f.side = side; f.fb = k;
                    f.foff = dir == 0 ? cidx(v, 2 * Fc - b[0], 2 * T - b[1]) : cidx(v, 2 * T - b[0], 2 * Fc - b[1]);
                    f.hi = dir == 0 ? cell_ref(H, C, Fc, T) : cell_ref(H, C, T, Fc);
                    f.lo = dir == 0 ? cell_ref(H, C, Fc - 1, T) : cell_ref(H, C, T, Fc - 1);
                    if (f.hi.b < 0 || f.lo.b < 0) { suhmo_set_error("hier: level %d is not properly nested in level %d (reflux)", l, l - 1); return -1; }
                    note(f.hi); note(f.lo);
                    f.bq = f.hi;                                                       // the face is the low face of its high-side cell
                    Ref t = side == 0 ? f.lo : f.hi;
                    auto key = std::make_pair(t.b, t.off);
                    if (!by_target.count(key)) order.push_back(key);
                    by_target[key].push_back(f);
                }
            }
        }
    }
    std::vector<Target> targets; std::vector<Face> faces;
    for (auto &key : order) {
        auto &fv = by_target[key];
        Ref t{key.first, key.second};
        // the register of a coarse cell is added up by the rank that holds the cell: a strip of a cut level 0 (the cell of the shadow -> the
        // same cell of that strip), the owner of its box, or everybody
        int exec = me;
        if (cut) {
            const int J = t.off / H->vglob.P - H->vglob.gy, I = t.off % H->vglob.P - SUHMO_XOFF;
            exec = J / sv.ny;
            if (exec == me) t.off = cidx(sv, I, J - sv.j0);
        } else if (C.l >= 1) exec = ownC(t.b);
        if (P)
            for (const Face &f : fv) {
                if (F.owner[f.fb] != exec) {                            // the fine cells and faces the register reads (k_reflux)
                    const int Pf = F.box[f.fb]->d[0].v.P;
                    for (int kk = 0; kk < 2; kk++) {
                        const int idx = f.foff + (f.dir == 0 ? kk * Pf : kk);
                        x_fface.push_back(Xf{F.owner[f.fb], f.fb, idx, exec});
                        x_fface.push_back(Xf{F.owner[f.fb], f.fb, f.dir == 0 ? idx - 1 : idx - Pf, exec});
                    }
                }
                if (C.l >= 1) for (const Ref &r : {f.hi, f.lo}) if (C.owner[r.b] != exec) x_cread.push_back(Xf{C.owner[r.b], r.b, r.off, exec});
            }
        if (exec != me) continue;
        targets.push_back(Target{t, (int)faces.size(), (int)fv.size()});
        faces.insert(faces.end(), fv.begin(), fv.end());
    }
    if (cut) {
        std::sort(needv.begin(), needv.end());
        needv.erase(std::unique(needv.begin(), needv.end()), needv.end());
        const int N = (int)needv.size();
        // the shadow keeps only the rows that hold a needed cell, in ascending order (rows next to each other stay next to each other: the
        // window rectangles, whose every cell is needed, remain rectangles): every offset into level 0 the plans carry is mapped over
        const int Pg = H->vglob.P, gyg = H->vglob.gy;
        std::vector<int> rowc(H->vglob.nyg, -1);
        for (int t = 0; t < N; t++) rowc[needv[t] / Pg - gyg] = 0;
        int nrow = 0;
        for (int J = 0; J < H->vglob.nyg; J++) if (rowc[J] == 0) rowc[J] = nrow++;
        auto remap = [&](int off) { return rowc[off / Pg - gyg] * Pg + off % Pg; };
        for (CfEnt &e : cf) { const int nn = e.kind == 0 || e.kind == 1 || e.kind == 3 ? 3 : (e.kind == 5 ? 1 : 2); for (int m = 0; m < nn; m++) e.c[m].off = remap(e.c[m].off); }
        for (PwlEnt &q : pwl) for (int m = 0; m < 9; m++) if (q.c[m].b >= 0) q.c[m].off = remap(q.c[m].off);
        for (WinEnt &w : wing) w.coff = remap(w.coff);
        for (Face &f : faces) { f.hi.off = remap(f.hi.off); f.lo.off = remap(f.lo.off); f.bq.off = remap(f.bq.off); }
        std::vector<int> needc(N);
        for (int t = 0; t < N; t++) needc[t] = remap(needv[t]);
        H->shadow_rows = nrow; H->shadow_elems = (size_t)Pg * (size_t)(nrow + 1);
        if (H->need_c.upload(needc)) { suhmo_set_error("hier: plan upload failed"); return -2; }
        std::vector<int2> rl(N);
        H->seg.assign(H->world + 1, 0);
        for (int t = 0; t < N; t++) {
            const int J = needv[t] / H->vglob.P - H->vglob.gy;
            const int r = J / sv.ny;
            H->seg[r + 1]++;
            rl[t].x = r;
        }
        for (int r = 0; r < H->world; r++) { H->cnt_max = std::max<long>(H->cnt_max, H->seg[r + 1]); H->seg[r + 1] += H->seg[r]; }
        for (int t = 0; t < N; t++) rl[t].y = t - H->seg[rl[t].x];
        if (H->need.upload(needv) || H->need_rl.upload(rl) || H->cover_full.upload(cover_full)) { suhmo_set_error("hier: plan upload failed"); return -2; }
    }
    int rc = 0;
    if (!P) {   // the side copies by source cell: per box W, E (ny entries each), S, N (nx each)
        std::vector<int> pbase(nb);
        size_t tot = 0;
        for (int k = 0; k < nb; k++) { const DV &v = F.box[k]->d[0].v; pbase[k] = (int)tot; tot += 2 * (size_t)(v.nx + v.ny); }
        std::vector<int2> push(tot, int2{-1, 0});
        for (const CopyEnt &e : ffs) {
            const DV &vo = F.box[e.s.b]->d[0].v, &vk = F.box[e.d.b]->d[0].v;
            const int js = e.s.off / vo.P - vo.gy, is = e.s.off % vo.P - SUHMO_XOFF;      // the source cell in its box
            const int jd = e.d.off / vk.P - vk.gy, id = e.d.off % vk.P - SUHMO_XOFF;      // the ghost cell in its box
            int slot;
            if (id < 0) slot = vo.ny + js;                  // a W ghost is fed by a cell on the E side of its box
            else if (id >= vk.nx) slot = js;
            else if (jd < 0) slot = 2 * vo.ny + vo.nx + is; // an S ghost by a cell on the N side
            else slot = 2 * vo.ny + is;
            const bool ok = (id < 0 ? is == vo.nx - 1 : id >= vk.nx ? is == 0 : jd < 0 ? js == vo.ny - 1 : js == 0);
            int2 &q = push[pbase[e.s.b] + slot];
            if (!ok || q.x >= 0) { suhmo_set_error("hier: internal: fine-fine copy without a unique source side cell"); return -4; }
            q = int2{e.d.b, e.d.off};
        }
        rc |= F.push.upload(push); rc |= F.pbase.upload(pbase);
    }
    if (!P) {   // several sweeps per launch (suhmo_gsrb.hip:k_gsrb_box_m): for every position of a box grown by 8 cells the box that holds the cell
        constexpr int G = SUHMO_BOX_HALO;
        size_t tot = 0;
        std::vector<int> hb(nb);
        bool fits = true;
        for (int k = 0; k < nb; k++) { const DV &v = F.box[k]->d[0].v; hb[k] = (int)tot; tot += (size_t)(v.nx + 2 * G) * (v.ny + 2 * G);
                                       fits = fits && v.nx >= 2 && v.ny >= 2; }
        if (fits && tot < (1u << 30)) {
            std::vector<int2> hv(tot);
            for (int k = 0; k < nb; k++) {
                const int *b = &F.b4[4 * k];
                const DV &v = F.box[k]->d[0].v;
                const int EW = v.nx + 2 * G;
                for (int ej = 0; ej < v.ny + 2 * G; ej++)
                    for (int ei = 0; ei < EW; ei++) {
                        int iw = b[0] - G + ei, jw = b[1] - G + ej;
                        int2 h = int2{-1, 0};
                        if (wrap_cell(H, F, iw, jw)) {
                            const int o = F.index.find(iw, jw);
                            if (o >= 0) { const DV &vo = F.box[o]->d[0].v; h = int2{o, cidx(vo, iw - vo.i0, jw - vo.j0)}; }
                        }
                        hv[hb[k] + (size_t)ej * EW + ei] = h;
                    }
            }
            rc |= F.halo.upload(hv); rc |= F.hbase.upload(hb);
            F.halo_ok = true;
        }
    }
    rc |= F.ff_side.upload(ffs);
    { std::vector<CopyEnt> all(ffs); all.insert(all.end(), ffc.begin(), ffc.end()); rc |= F.ff_all.upload(all); } rc |= F.cf.upload(cf); rc |= F.pwl.upload(pwl);
    rc |= F.avg.upload(avg); rc |= F.wing.upload(wing); rc |= F.targets.upload(targets); rc |= F.faces.upload(faces);
    rc |= F.avg_cov.upload(avg_cov);
    for (auto &e : avg_cov) { F.cov_w = std::max(F.cov_w, e.w); F.cov_h = std::max(F.cov_h, e.h); }
    if (P) {
        rc |= F.avg_put.upload(avg_put); rc |= F.avg_get.upload(avg_get);
        F.put_stride = *std::max_element(putpos.begin(), putpos.end());
        for (auto &e : avg_put) { F.put_w = std::max(F.put_w, e.w); F.put_h = std::max(F.put_h, e.h); F.put_mine += (long)e.w * e.h; }
        for (auto &e : avg_get) { F.get_w = std::max(F.get_w, e.w); F.get_h = std::max(F.get_h, e.h); }
        std::vector<Xf> x_sides(x_side[0]); x_sides.insert(x_sides.end(), x_side[1].begin(), x_side[1].end());
        rc |= make_sync(H, F, x_side[0], F.sy_side[0]); rc |= make_sync(H, F, x_side[1], F.sy_side[1]); rc |= make_sync(H, F, x_sides, F.sy_sides);
        rc |= make_sync(H, F, x_all, F.sy_all); rc |= make_sync(H, F, x_fface, F.sy_fface);
        if (C.l >= 1) { rc |= make_sync(H, C, x_cread, F.sy_cread); rc |= make_sync(H, C, x_win, F.sy_win); }
        H->side_bytes[l] = 8 * (long)std::max(F.sy_side[0].send.n, F.sy_side[1].send.n);
    }
    if (C.l == 0) {
        std::sort(gcell.begin(), gcell.end());
        gcell.erase(std::unique(gcell.begin(), gcell.end()), gcell.end());
        std::vector<int2> gc;
        for (auto &q : gcell) if (q.first >= sv.j0 && q.first < sv.j0 + sv.ny) gc.push_back(int2{q.second, q.first - sv.j0});   // own rows, local (i, j)
        rc |= F.gcells.upload(gc); rc |= F.dirty0.upload(dirty0);
        F.dirty_w = F.dirty_h = 0;
        for (auto &r : dirty0) { F.dirty_w = std::max(F.dirty_w, r.z); F.dirty_h = std::max(F.dirty_h, r.w); }
    }
    if (rc) { suhmo_set_error("hier: plan upload failed"); return -2; }
    F.avg_w = F.avg_h = F.wing_w = F.wing_h = 0;
    for (auto &e : avg) { F.avg_w = std::max(F.avg_w, e.w); F.avg_h = std::max(F.avg_h, e.h); }
    for (auto &e : wing) { F.wing_w = std::max(F.wing_w, e.w); F.wing_h = std::max(F.wing_h, e.h); }
    {   // the pieces of a box's window are consecutive (the boxes were visited in order)
        std::vector<int> ws(nb + 1, 0);
        for (int b : wing_box) ws[b + 1]++;
        for (int k = 0; k < nb; k++) ws[k + 1] += ws[k];
        for (size_t t = 1; t < wing_box.size(); t++) if (wing_box[t] < wing_box[t - 1]) { suhmo_set_error("hier: internal: window pieces out of order"); return -4; }
        if (F.wstart.upload(ws)) { suhmo_set_error("hier: plan upload failed"); return -2; }
        for (const Win &w : F.win) F.win_max = std::max(F.win_max, w.nx * w.ny);
    }
    F.winelems = wtot;
    if (hipMalloc(&F.winbuf, std::max<size_t>(1, wtot) * sizeof(double)) != hipSuccess) { suhmo_set_error("hier: window allocation failed"); return -2; }
    (void)hipMemset(F.winbuf, 0, std::max<size_t>(1, wtot) * sizeof(double));
    if (hipMalloc(&F.d_win, std::max<size_t>(1, F.win.size()) * sizeof(Win)) != hipSuccess) return -2;
    if (hipMalloc(&F.d_wing_box, std::max<size_t>(1, wing_box.size()) * sizeof(int)) != hipSuccess) return -2;
    if (hipMemcpy(F.d_win, F.win.data(), F.win.size() * sizeof(Win), hipMemcpyHostToDevice) != hipSuccess) return -2;
    if (!wing_box.empty() && hipMemcpy(F.d_wing_box, wing_box.data(), wing_box.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -2;
    return 0;
}
}  // namespace

// ------------------------------------------------------------------ tables, launches of the plans
namespace {
#define HST(s) ((hipStream_t)(s))
inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// device table of the boxes' field pointers (levels >= 1).  The boxes relax in place, so a pointer changes only when a
// field is allocated for the first time: then the table is uploaded again (rare; synchronous).
int refresh_tables(suhmo_hier *H, int l, hipStream_t st)
{
    HLev &V = H->lev[l];
    if (l == 0) return 0;
    const size_t nb = V.box.size();
    // (63 boxes x 32 pointers compared before every launch was a third of the host's time per launch: no field pointer anywhere has
    //  changed since the last comparison -> the tables are current)
    const unsigned long epoch = suhmo_fp_epoch();
    if (V.d_fp && V.d_dv && V.h_fp.size() == nb && V.tab_epoch == epoch) return 0;
    V.tab_epoch = epoch;
    bool dirty = V.d_fp == nullptr;
    if (V.h_fp.size() != nb) { V.h_fp.assign(nb, FP{}); dirty = true; }
    for (size_t k = 0; k < nb; k++)
        if (memcmp(&V.h_fp[k], &V.box[k]->d[0].fp, sizeof(FP))) { V.h_fp[k] = V.box[k]->d[0].fp; dirty = true; }
    if (!dirty) return 0;
    HIPCHK(hipStreamSynchronize(st));
    if (!V.d_fp) { HIPCHK(hipMalloc(&V.d_fp, nb * sizeof(FP))); HIPCHK(hipMalloc(&V.d_fp_alt, nb * sizeof(FP))); }
    HIPCHK(hipMemcpy(V.d_fp, V.h_fp.data(), nb * sizeof(FP), hipMemcpyHostToDevice));
    V.h_fp_alt = V.h_fp;
    for (FP &f : V.h_fp_alt) std::swap(f.f[SUHMO_F_PHI], f.f[SUHMO_F_PHI2]);
    HIPCHK(hipMemcpy(V.d_fp_alt, V.h_fp_alt.data(), nb * sizeof(FP), hipMemcpyHostToDevice));
    if (!V.d_dv) {
        std::vector<DV> dv(nb);
        for (size_t k = 0; k < nb; k++) dv[k] = V.box[k]->d[0].v;
        HIPCHK(hipMalloc(&V.d_dv, nb * sizeof(DV)));
        HIPCHK(hipMemcpy(V.d_dv, dv.data(), nb * sizeof(DV), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&V.d_red, (64 * nb + 16) * sizeof(double)));
        for (size_t k = 0; k < nb; k++) { V.maxnx = std::max(V.maxnx, dv[k].nx); V.maxny = std::max(V.maxny, dv[k].ny); }
    }
    return 0;
}
// the two canvases of the head of a level of boxes trade places: in the boxes' handles and by switching to the table that lists them the other way
// round (no copy, nothing uploaded; requires SUHMO_F_PHI2 on every box and current tables)
void swap_head(suhmo_hier *H, int l)
{
    HLev &V = H->lev[l];
    for (suhmo_level *L : V.box) std::swap(L->d[0].fp.f[SUHMO_F_PHI], L->d[0].fp.f[SUHMO_F_PHI2]);
    std::swap(V.d_fp, V.d_fp_alt);
    V.h_fp.swap(V.h_fp_alt);
    V.swapped = !V.swapped;
}
int ensure_field(suhmo_hier *H, int l, int field)
{
    HLev &V = H->lev[l];
    if (V.ensured >> field & 1ull) return 0;
    for (size_t k = 0; k < V.box.size(); k++) {
        if (V.part && !V.held[k]) continue;                            // a box other ranks hold: no storage here
        if (!suhmo_field(V.box[k], 0, field)) { suhmo_set_error("field allocation failed"); return -2; }
    }
    V.ensured |= 1ull << field;
    return 0;
}
// ---- shadow of a level 0 cut into rank strips
constexpr int XF = 4;                                    // fields per all-gather
struct FList { const double *src[XF]; double *dst[XF]; int n; };
__global__ void k_need_pack(const int *__restrict__ need, int first, int n, int shift, FList fl, double *__restrict__ buf, long stride)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int off = need[first + t] - shift;             // the cell in this rank's strip canvas (same pitch, same ghost rows)
    for (int f = 0; f < fl.n; f++) buf[f * stride + t] = fl.src[f][off];
}
__global__ void k_need_unpack(const int *__restrict__ need /* offsets in the compact shadow */, const int2 *__restrict__ rl, int n, FList fl, const double *__restrict__ buf, long stride)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int2 q = rl[t];
    const int off = need[t];
    for (int f = 0; f < fl.n; f++) fl.dst[f][off] = buf[((long)q.x * fl.n + f) * stride + q.y];
}
double *shadow_field(suhmo_hier *H, int field)
{
    if (!H->shadow.f[field]) {
        double *p = nullptr;
        if (hipMalloc(&p, H->shadow_elems * sizeof(double)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, H->shadow_elems * sizeof(double));
        H->shadow.f[field] = p;
    }
    return H->shadow.f[field];
}
// the shadow's copies of `fields` of level 0 <- the owners' current values (collective over the ranks of level 0)
int refresh_base(suhmo_hier *H, const int *fields, int nf, hipStream_t st)
{
    if (!dist_base(H) || H->nlev < 2) return 0;
    // the head is read by several plans in a row (coarse-fine interpolation before every operator of level 1) while level 0 rests:
    // one all-gather serves them.  Whatever writes level 0's head clears the flag (the average from level 1, its own V-cycle, a
    // copy into it) and so does every entry point of the C-ABI (the caller may have loaded new data)
    if (nf == 1 && fields[0] == SUHMO_F_PHI && H->phi_shadow_fresh) return 0;
    SUHMO_TIME("hier: all-gather of the coarse cells level 1 reads");
    if (!H->ag) { suhmo_set_error("hier: level 0 is a rank strip and no all-gather is attached (suhmo_hier_attach_rccl / suhmo_hier_set_allgather)"); return -1; }
    ARG(nf >= 1 && nf <= XF);
    suhmo_level *B = base_of(H);
    FList fl;
    fl.n = nf;
    for (int f = 0; f < nf; f++) {
        fl.src[f] = suhmo_field(B, 0, fields[f]); fl.dst[f] = shadow_field(H, fields[f]);
        if (!fl.src[f] || !fl.dst[f]) { suhmo_set_error("field allocation failed"); return -2; }
    }
    const long stride = H->cnt_max, count = stride * nf;
    const size_t cap = (size_t)stride * XF;
    if (!H->xs) {
        HIPCHK(hipMalloc(&H->xs, std::max<size_t>(1, cap) * sizeof(double)));
        HIPCHK(hipMalloc(&H->xr, std::max<size_t>(1, cap * H->world) * sizeof(double)));
        HIPCHK(hipMemset(H->xs, 0, std::max<size_t>(1, cap) * sizeof(double)));
    }
    const int first = H->seg[H->rank], mine = H->seg[H->rank + 1] - first;
    const DV &sv = B->d[0].v;
    if (mine) hipLaunchKernelGGL(k_need_pack, g1(mine), dim3(256), 0, st, H->need.d, first, mine, sv.j0 * sv.P, fl, H->xs, stride);
    HIPCHK(hipGetLastError());
    int rc = H->ag(H->ag_user, H->xs, count, H->xr, (suhmo_stream_t)st);
    if (rc) return rc;
    H->gathers++;
    if (H->need.n) hipLaunchKernelGGL(k_need_unpack, g1(H->need.n), dim3(256), 0, st, H->need_c.d, H->need_rl.d, (int)H->need.n, fl, H->xr, stride);
    HIPCHK(hipGetLastError());
    for (int f = 0; f < nf; f++) if (fields[f] == SUHMO_F_PHI) H->phi_shadow_fresh = true;
    return 0;
}
inline int refresh_base1(suhmo_hier *H, int field, hipStream_t st) { return refresh_base(H, &field, 1, st); }

// ---- owner computes: packed cells from their owners to the mirrors of the ranks that read them
struct FIdx { int f[XF]; int n; };
__global__ void k_sync_pack(const Ref *__restrict__ e, int n, const FP *__restrict__ tab, FIdx fl, double *__restrict__ buf, long stride)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const Ref q = e[t];
    for (int f = 0; f < fl.n; f++) buf[f * stride + t] = tab[q.b].f[fl.f[f]][q.off];
}
__global__ void k_sync_unpack(const SyncRecv *__restrict__ e, int n, const FP *__restrict__ tab, FIdx fl, const double *__restrict__ buf, long stride)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const SyncRecv q = e[t];
    for (int f = 0; f < fl.n; f++) tab[q.d.b].f[fl.f[f]][q.d.off] = buf[((long)q.rank * fl.n + f) * stride + q.pos];
}
int part_staging(suhmo_hier *H, size_t doubles, hipStream_t st)
{
    if (doubles <= H->pcap) return 0;
    if (H->ps) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(H->ps); (void)hipFree(H->pr); H->ps = H->pr = nullptr; }
    H->pcap = doubles + doubles / 4 + 64;
    HIPCHK(hipMalloc(&H->ps, H->pcap * sizeof(double)));
    HIPCHK(hipMalloc(&H->pr, H->pcap * H->world * sizeof(double)));
    HIPCHK(hipMemsetAsync(H->ps, 0, H->pcap * sizeof(double), st));
    return 0;
}
// fields of level lt (the level whose cells S lists) from their owners into this rank's mirrors; collective over the ranks (skipped by all
// of them alike when nothing of this kind travels anywhere)
int sync_run(suhmo_hier *H, int lt, Sync &S, const int *fields, int nf, hipStream_t st)
{
    if (S.stride == 0) return 0;
    SUHMO_TIME("hier: exchange of packed cells between the owners of a level's boxes");
    if (!H->ag) { suhmo_set_error("hier: level %d is partitioned over the ranks and no all-gather is attached (suhmo_hier_attach_rccl / suhmo_hier_set_allgather)", lt); return -1; }
    ARG(nf >= 1 && nf <= XF);
    int rc;
    FIdx fl; fl.n = nf;
    for (int f = 0; f < nf; f++) { fl.f[f] = fields[f]; if ((rc = ensure_field(H, lt, fields[f]))) return rc; }
    if ((rc = refresh_tables(H, lt, st)) || (rc = part_staging(H, (size_t)nf * S.stride, st))) return rc;
    HLev &V = H->lev[lt];
    if (S.send.n) hipLaunchKernelGGL(k_sync_pack, g1(S.send.n), dim3(256), 0, st, S.send.d, (int)S.send.n, V.d_fp, fl, H->ps, S.stride);
    HIPCHK(hipGetLastError());
    if ((rc = H->ag(H->ag_user, H->ps, (long)nf * S.stride, H->pr, (suhmo_stream_t)st))) return rc;
    H->part_gathers++; H->part_bytes += 8L * nf * (long)S.send.n;
    if (S.recv.n) hipLaunchKernelGGL(k_sync_unpack, g1(S.recv.n), dim3(256), 0, st, S.recv.d, (int)S.recv.n, V.d_fp, fl, H->pr, S.stride);
    HIPCHK(hipGetLastError());
    return 0;
}
inline int sync1(suhmo_hier *H, int lt, Sync &S, int field, hipStream_t st) { return sync_run(H, lt, S, &field, 1, st); }

// coarse-side arguments of a kernel that reads / writes level l-1.  base / bdv: where the cells of level 0 are READ (the shadow
// of a cut level 0); dst / ddv: where they are written (the level, or this rank's strip of it)
struct CoarseArgs { const FP *tab; const DV *dv; FP base; DV bdv; FP dst; DV ddv; int use_base; };
int coarse_args(suhmo_hier *H, int lc, hipStream_t st, CoarseArgs &a)
{
    memset(&a, 0, sizeof(a));
    if (lc == 0) {
        a.dst = base_of(H)->d[0].fp; a.ddv = base_of(H)->d[0].v; a.use_base = 1;
        if (dist_base(H)) { a.base = H->shadow; a.bdv = H->vglob; } else { a.base = a.dst; a.bdv = a.ddv; }
        return 0;
    }
    int rc = refresh_tables(H, lc, st); if (rc) return rc;
    a.tab = H->lev[lc].d_fp; a.dv = H->lev[lc].d_dv; a.bdv = H->lev[lc].box[0]->d[0].v;
    return 0;
}

// all boxes of level l >= 1 as one launch target
int multi_of(suhmo_hier *H, int l, hipStream_t st, suhmo_multi &m)
{
    int rc = refresh_tables(H, l, st); if (rc) return rc;
    HLev &V = H->lev[l];
    m.dv = V.d_dv; m.fp = V.d_fp; m.nbox = (int)V.box.size(); m.maxnx = V.maxnx; m.maxny = V.maxny; m.red = V.d_red;
    m.push = V.push.d; m.pbase = V.pbase.d; m.merged = H->merged_launches;
    if (V.part) { m.dv += V.b0; m.fp += V.b0; m.nbox = V.nown; m.push = nullptr; m.pbase = nullptr; }   // owner computes: the tables from this rank's first box
    return 0;
}
inline const suhmo_phys_t &phys_of(suhmo_hier *H, int l) { return H->lev[l].box[0]->ph; }
inline bool has_alpha(suhmo_hier *H, int l) { return H->lev[l].box[0]->d[0].v.alpha != 0.0; }

// Copier::exchange of one or two cell fields of level l
// colour >= 0 (owner computes, after a colour pass of the head): only the side cells of that colour have changed and travel
int hier_ff(suhmo_hier *H, int l, int f0, int f1, bool corners, hipStream_t st, int colour = -1)
{
    if (l == 0) return 0;                                   // the base canvas: a neighbour's cell IS the ghost
    HLev &V = H->lev[l];
    int rc;
    if ((rc = ensure_field(H, l, f0)) || (f1 >= 0 && (rc = ensure_field(H, l, f1))) || (rc = refresh_tables(H, l, st))) return rc;
    const bool head_sides = f0 == SUHMO_F_PHI && f1 < 0 && !corners;
    if (head_sides && H->ff_seen[l] == H->phi_ver[l]) return 0;              // the side ghosts of the head are current
    if (head_sides) H->ff_seen[l] = H->phi_ver[l];
    if (V.part) {                                           // the source cells other ranks own -> their mirrors here, then the copies below
        const int fl[2] = {f0, f1};
        Sync &S = corners ? V.sy_all : (colour >= 0 ? V.sy_side[colour & 1] : V.sy_sides);
        if ((rc = sync_run(H, l, S, fl, f1 >= 0 ? 2 : 1, st))) return rc;
    }
    // (a corner ghost's source is a valid cell, never a ghost: sides and corners do not depend on each other)
    const DevVec<CopyEnt> &list = corners ? V.ff_all : V.ff_side;
    if (list.n) hipLaunchKernelGGL(k_ff, g1(list.n), dim3(256), 0, st, list.d, (int)list.n, V.d_fp, f0, f1);
    HIPCHK(hipGetLastError());
    return 0;
}
// QuadCFInterp: coarse-fine ghosts of field ff of level l <- field fc of level l-1
int hier_cf(suhmo_hier *H, int l, int ff, int fc, hipStream_t st, int ff1 = -1, int fc1 = -1)
{
    SUHMO_TIME("QuadCFInterp::coarseFineInterp");
    if (l == 0) return 0;
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l, ff)) || (rc = ensure_field(H, l - 1, fc)) || (rc = refresh_tables(H, l, st))) return rc;
    if (ff1 >= 0 && ((rc = ensure_field(H, l, ff1)) || (rc = ensure_field(H, l - 1, fc1)) || (rc = refresh_tables(H, l, st)))) return rc;
    { const int fl[2] = {fc, fc1};
      if (l == 1) rc = refresh_base(H, fl, ff1 >= 0 ? 2 : 1, st); else rc = V.part ? sync_run(H, l - 1, V.sy_cread, fl, ff1 >= 0 ? 2 : 1, st) : 0;
      if (rc) return rc; }
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    if (V.cf.n) hipLaunchKernelGGL(k_cf, g1(V.cf.n), dim3(256), 0, st, V.cf.d, (int)V.cf.n, V.d_fp, ff, ca.tab, ca.base, ca.use_base, fc, ff1, fc1);
    HIPCHK(hipGetLastError());
    return 0;
}
// hier_cf + hier_ff of the same field(s) in one launch (levels held whole by this process)
int hier_cf_ff(suhmo_hier *H, int l, int ff, int fc, int ff1, int fc1, bool corners, hipStream_t st)
{
    SUHMO_TIME("QuadCFInterp::coarseFineInterp + exchange");
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l, ff)) || (rc = ensure_field(H, l - 1, fc)) || (rc = refresh_tables(H, l, st))) return rc;
    if (ff1 >= 0 && ((rc = ensure_field(H, l, ff1)) || (rc = ensure_field(H, l - 1, fc1)) || (rc = refresh_tables(H, l, st)))) return rc;
    { const int fl[2] = {fc, fc1};
      if (l == 1 && (rc = refresh_base(H, fl, ff1 >= 0 ? 2 : 1, st))) return rc; }
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    const DevVec<CopyEnt> &list = corners ? V.ff_all : V.ff_side;
    const int nbcf = (int)g1(V.cf.n).x, nbff = (int)g1(list.n).x;
    if (nbcf + nbff > 0)
        hipLaunchKernelGGL(k_cf_ff, dim3(nbcf + nbff), dim3(256), 0, st, V.cf.d, (int)V.cf.n, nbcf, list.d, (int)list.n, V.d_fp, ff, ca.tab, ca.base, ca.use_base, fc, ff1, fc1);
    HIPCHK(hipGetLastError());
    return 0;
}
int hier_pwl(suhmo_hier *H, int l, int ff, int fc, hipStream_t st)
{
    if (l == 0) return 0;
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l, ff)) || (rc = ensure_field(H, l - 1, fc)) || (rc = refresh_tables(H, l, st))) return rc;
    if (l == 1 && (rc = refresh_base1(H, fc, st))) return rc;
    if (l > 1 && V.part && (rc = sync1(H, l - 1, V.sy_cread, fc, st))) return rc;
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    if (V.pwl.n) hipLaunchKernelGGL(k_pwl, g1(V.pwl.n), dim3(256), 0, st, V.pwl.d, (int)V.pwl.n, V.d_fp, ff, ca.tab, ca.base, ca.use_base, fc);
    HIPCHK(hipGetLastError());
    return 0;
}
// FORT_AVERAGE of field ff of level l into the covered cells of field fc of level l-1 (mode 0) / covered cells <- val (mode 1)
int hier_avg(suhmo_hier *H, int l, int ff, int fc, int mode, double val, hipStream_t st)
{
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l, ff)) || (rc = ensure_field(H, l - 1, fc)) || (rc = refresh_tables(H, l, st)) || (rc = coarse_args(H, l - 1, st, ca))) return rc;
    if (fc == SUHMO_F_PHI) { for (suhmo_level *L : H->lev[l - 1].box) L->d[0].phi_fresh = 0; H->phi_ver[l - 1]++; }
    if (fc == SUHMO_F_PHI && l == 1) H->phi_shadow_fresh = false;
    if (mode == 1) {                                        // geometry only: every holder of coarse cells marks / zeroes its own
        if (V.avg_cov.n) {
            dim3 grd((V.cov_w + 63) / 64, (V.cov_h + 3) / 4, (unsigned)V.avg_cov.n);
            hipLaunchKernelGGL(k_avg, grd, dim3(64, 4), 0, st, V.avg_cov.d, V.d_fp, V.d_dv, ff, ca.tab, ca.dv, ca.dst, ca.ddv, ca.use_base, fc, mode, val);
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (V.avg.n) {
        dim3 grd((V.avg_w + 63) / 64, (V.avg_h + 3) / 4, (unsigned)V.avg.n);
        hipLaunchKernelGGL(k_avg, grd, dim3(64, 4), 0, st, V.avg.d, V.d_fp, V.d_dv, ff, ca.tab, ca.dv, ca.dst, ca.ddv, ca.use_base, fc, mode, val);
    }
    HIPCHK(hipGetLastError());
    if (V.part && V.put_stride) {
        // owner computes: averages of this rank's fine boxes over coarse cells another rank holds travel in this rank's segment of one
        // all-gather; the holder of the coarse cells takes its rectangles from there (FORT_AVERAGE's sum, the same bits)
        SUHMO_TIME("hier: averages onto coarse cells other ranks hold");
        if (!H->ag) { suhmo_set_error("hier: level %d is partitioned over the ranks and no all-gather is attached", l); return -1; }
        if ((rc = part_staging(H, (size_t)V.put_stride, st))) return rc;
        if (V.avg_put.n) {
            dim3 grd((V.put_w + 63) / 64, (V.put_h + 3) / 4, (unsigned)V.avg_put.n);
            hipLaunchKernelGGL(k_avg_put, grd, dim3(64, 4), 0, st, V.avg_put.d, V.d_fp, V.d_dv, ff, H->ps);
        }
        HIPCHK(hipGetLastError());
        if ((rc = H->ag(H->ag_user, H->ps, V.put_stride, H->pr, (suhmo_stream_t)st))) return rc;
        H->part_gathers++;
        H->part_bytes += 8L * V.put_mine;
        if (V.avg_get.n) {
            dim3 grd((V.get_w + 63) / 64, (V.get_h + 3) / 4, (unsigned)V.avg_get.n);
            hipLaunchKernelGGL(k_put_unpack, grd, dim3(64, 4), 0, st, V.avg_get.d, ca.tab, ca.dv, ca.dst, ca.ddv, ca.use_base, fc, H->pr, V.put_stride);
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// AMRProlongS_2 (:1143-1206): PHI of level l += PROLONG_2_NL(field_c of level l-1), the coarse field gathered per box with
// its physical-BC ghosts (inhomogeneous in FAS mode)
// minus_saved: the coarse field is field_c minus what hier_window_save kept of it (the correction of a FAS cycle: only the
// windows of it are ever formed)
int hier_window_save(suhmo_hier *H, int l, int field_c, hipStream_t st)
{
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l - 1, field_c)) || (rc = refresh_tables(H, l, st))) return rc;
    if (l == 1 && (rc = refresh_base1(H, field_c, st))) return rc;
    if (l > 1 && V.part && (rc = sync1(H, l - 1, V.sy_win, field_c, st))) return rc;
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    if (!V.winold) {
        HIPCHK(hipMalloc(&V.winold, std::max<size_t>(1, V.winelems) * sizeof(double)));
        HIPCHK(hipMemsetAsync(V.winold, 0, std::max<size_t>(1, V.winelems) * sizeof(double), st));
    }
    if (V.wing.n) {
        dim3 grd((V.wing_w + 63) / 64, (V.wing_h + 3) / 4, (unsigned)V.wing.n);
        hipLaunchKernelGGL(k_win_gather, grd, dim3(64, 4), 0, st, V.wing.d, V.winold, ca.tab, ca.dv, ca.base, ca.bdv, ca.use_base, field_c, V.d_win, V.d_wing_box);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// leave_below (l - 1 >= 1): the level below leaves its FAS problem in the same launch, field_c = PHI minus PHIOLD formed on the fly
int hier_prolong2(suhmo_hier *H, int l, int field_c, hipStream_t st, bool minus_saved = false, bool leave_below = false)
{
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if (leave_below && !(H->fused_prolong && V.win_max <= 6144 && !V.part && !H->lev[l - 1].part && l - 1 >= 1)) {     // two launches after all
        suhmo_multi mc;
        if ((rc = multi_of(H, l - 1, st, mc)) || (rc = suhmo_multi_fas_leave(mc, st))) return rc;
        return hier_prolong2(H, l, SUHMO_F_CORR, st);
    }
    if ((rc = ensure_field(H, l - 1, field_c)) || (rc = refresh_tables(H, l, st))) return rc;
    if (l == 1 && (rc = refresh_base1(H, field_c, st))) return rc;
    if (l > 1 && V.part && (rc = sync1(H, l - 1, V.sy_win, field_c, st))) return rc;
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    const int nb = (int)V.box.size();
    for (suhmo_level *L : V.box) L->d[0].phi_fresh = 0;
    H->phi_ver[l]++;
    const int k0 = V.part ? V.b0 : 0, nk = V.part ? V.nown : nb;       // (owner computes: the windows of this rank's boxes)
    if (nk <= 0) return 0;
    if (leave_below) {
        suhmo_multi mc;
        if ((rc = multi_of(H, l - 1, st, mc))) return rc;
        const int cgx = (mc.maxnx + 2 + 63) / 64, cgy = (mc.maxny + 2 + 3) / 4;
        hipLaunchKernelGGL(k_prolong2_fused, dim3(nk + cgx * cgy * mc.nbox), dim3(256), (size_t)V.win_max * sizeof(double), st, V.wing.d, V.wstart.d, V.d_win, k0,
                           ca.tab, ca.dv, ca.base, ca.bdv, ca.use_base, (int)SUHMO_F_PHI, (const double *)nullptr, V.d_fp, V.d_dv, (int)SUHMO_F_PHIOLD, nk, cgx, cgy);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (H->fused_prolong && V.win_max <= 6144) {                       // gather + BC + PROLONG_2_NL of a box in one workgroup, the window in LDS
        hipLaunchKernelGGL(k_prolong2_fused, dim3(nk), dim3(256), (size_t)V.win_max * sizeof(double), st, V.wing.d, V.wstart.d, V.d_win, k0,
                           ca.tab, ca.dv, ca.base, ca.bdv, ca.use_base, field_c, minus_saved ? V.winold : nullptr, V.d_fp, V.d_dv, -1, nk, 1, 1);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (V.wing.n) {
        dim3 grd((V.wing_w + 63) / 64, (V.wing_h + 3) / 4, (unsigned)V.wing.n);
        hipLaunchKernelGGL(k_win_gather, grd, dim3(64, 4), 0, st, V.wing.d, V.winbuf, ca.tab, ca.dv, ca.base, ca.bdv, ca.use_base, field_c, V.d_win, V.d_wing_box,
                           minus_saved ? V.winold : nullptr);
    }
    int maxp = 0, maxx = 0, maxy = 0;
    for (const Win &w : V.win) maxp = std::max(maxp, 2 * (w.nx - 2) + 2 * (w.ny - 2));
    for (suhmo_level *L : V.box) { maxx = std::max(maxx, L->d[0].v.nx); maxy = std::max(maxy, L->d[0].v.ny); }
    hipLaunchKernelGGL(k_win_bc, dim3((maxp + 255) / 256, nk), dim3(256), 0, st, V.d_win + k0, nk, V.winbuf, ca.bdv);
    hipLaunchKernelGGL(k_prolong2_win, dim3((maxx + 63) / 64, (maxy + 3) / 4, nk), dim3(64, 4), 0, st, V.d_win + k0, V.winbuf, V.d_fp + k0, V.d_dv + k0);
    HIPCHK(hipGetLastError());
    return 0;
}
// reflux (src/VCAMRNonLinearPoissonOp.cpp:555-652): field_c of level l-1 (holding L(phi)) += the flux mismatch on the
// coarse-fine faces of level l
int hier_reflux(suhmo_hier *H, int l, int field_c, hipStream_t st, int residual = 0)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::reflux");
    HLev &V = H->lev[l];
    int rc;
    CoarseArgs ca;
    if ((rc = ensure_field(H, l - 1, field_c)) || (rc = refresh_tables(H, l, st))) return rc;
    { const int fl[3] = {SUHMO_F_PHI, SUHMO_F_BX, SUHMO_F_BY};
      if (l == 1) { if ((rc = refresh_base(H, fl, 3, st))) return rc; }
      else if (V.part && (rc = sync_run(H, l - 1, V.sy_cread, fl, 3, st))) return rc;
      // owner computes: the register is added up where the coarse cell lives; the fine cells and faces next to the coarse-fine faces come along
      if (V.part && (rc = sync_run(H, l, V.sy_fface, fl, 3, st))) return rc; }
    if ((rc = coarse_args(H, l - 1, st, ca))) return rc;
    const DV &vc = H->lev[l - 1].box[0]->d[0].v;
    if (V.targets.n)
        hipLaunchKernelGGL(k_reflux, g1(V.targets.n), dim3(256), 0, st, V.targets.d, (int)V.targets.n, V.faces.d, V.d_fp, V.d_dv, ca.tab, ca.base, ca.dst,
                           ca.use_base, field_c, vc.dx, vc.dy, vc.beta, residual);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ operator methods of a level
// LoadBalance(procIDs, grids) (src/AmrHydro.cpp:4283, 4929): here the boxes in the order given, cut into `world` runs of about equal cell
// counts (a box goes to the rank its middle cell falls to): deterministic, contiguous, the same on every rank.  Called BEFORE the plans are
// built: every plan entry is kept by the rank that executes it
int part_setup(suhmo_hier *H, int l)
{
    HLev &V = H->lev[l];
    const int nb = (int)V.box.size(), W = H->world;
    V.part = H->part;
    if (!V.part) return 0;
    long total = 0;
    for (int k = 0; k < nb; k++) { const int *b = &V.b4[4 * k]; total += (long)(b[2] - b[0] + 1) * (b[3] - b[1] + 1); }
    V.owner.assign(nb, 0);
    long before = 0;
    int prev = 0;
    for (int k = 0; k < nb; k++) {
        const int *b = &V.b4[4 * k];
        const long c = (long)(b[2] - b[0] + 1) * (b[3] - b[1] + 1);
        const int r = std::max(prev, std::min(W - 1, (int)(((before + c / 2) * W) / total)));
        V.owner[k] = prev = r;
        if (r == H->rank) V.owned_cells += c;
        before += c;
    }
    V.own.assign(W + 1, nb);
    for (int r = 0; r < W; r++) V.own[r] = (int)(std::lower_bound(V.owner.begin(), V.owner.end(), r) - V.owner.begin());
    V.b0 = V.own[H->rank]; V.nown = V.own[H->rank + 1] - V.b0;
    V.held.assign(nb, 0);
    for (int k = V.b0; k < V.b0 + V.nown; k++) V.held[k] = 1;
    return 0;
}

int hier_gsrb(suhmo_hier *H, int l, int sweeps, suhmo_stream_t s, bool may_swap = false)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::relaxNF");
    if (l == 0) { H->phi_shadow_fresh = false; H->phi_ver[0]++; H->base_full_ver++; return suhmo_level_gsrb(base_of(H), 0, sweeps, s); }
    int rc;
    suhmo_multi m;
    if ((rc = multi_of(H, l, HST(s), m))) return rc;
    HLev &Vf = H->lev[l];
    if (sweeps > 0 && H->fused_relax && Vf.halo_ok && !Vf.part) {
        // two sweeps per launch: a box's workgroup relaxes the box and, redundantly, the 4 cells around it that belong to its neighbours,
        // reading their canvases directly -- no exchange between the colour passes, a quarter of the launches (suhmo_gsrb.hip:k_gsrb_box_m)
        if ((rc = ensure_field(H, l, SUHMO_F_PHI2)) || (rc = multi_of(H, l, HST(s), m))) return rc;
        if (Vf.self_wrap < 0) {                            // a box that is its own periodic neighbour: its physical ghost is another tile's cell
            Vf.self_wrap = 0;
            for (suhmo_level *L : Vf.box) { const DV &v = L->d[0].v; if ((v.per[0] && !v.cfx[0]) || (v.per[1] && !v.ext[0])) Vf.self_wrap = 1; }
        }
        const int bcg = H->merged_launches && !Vf.self_wrap;             // the closing ghost fill (:757-759) rides in the relaxation launches
        int src = SUHMO_F_PHI, dst = SUHMO_F_PHI2;
        const int per = H->box_sweeps;                      // sweeps per launch: 4 (the whole smoothing of num_smooth = 4) or 2
        for (int done = 0; done < sweeps; done += per) {
            const int npass = 2 * std::min(per, sweeps - done);
            if ((rc = suhmo_multi_gsrb_box(m, phys_of(H, l), has_alpha(H, l), Vf.halo.d, Vf.hbase.d, src, dst, npass, bcg, HST(s)))) return rc;
            std::swap(src, dst);
            H->phi_ver[l]++; H->n_fused_relax++;
        }
        if (src != SUHMO_F_PHI) {                          // an odd number of launches: the result is on the second canvas
            if (may_swap && Vf.d_fp_alt) swap_head(H, l);  // (inside a V-cycle: that canvas becomes the head; vcycle_amr puts things back)
            else if ((rc = suhmo_multi_copy(m, SUHMO_F_PHI, SUHMO_F_PHI2, HST(s)))) return rc;
        }
        return bcg ? 0 : suhmo_multi_fill_ghosts(m, SUHMO_F_PHI, 1, HST(s));                                  // :757-759
    }
    // exchange() before every colour pass (:692, :751): once here (unless the side ghosts are current), then every pass pushes
    // its new side cells into the ghost cells they feed
    if (sweeps > 0 && (rc = hier_ff(H, l, SUHMO_F_PHI, -1, false, HST(s)))) return rc;
    // owner computes: a pass relaxes this rank's boxes, the side cells of the colour it advanced travel to the ranks whose boxes lie across
    // those sides, and the exchange launch copies them into the ghost cells: the reference's own pattern, exchange() + pass (:692, :751)
    HLev &V = H->lev[l];
    const bool push = H->push_ghosts && !V.part;
    for (int it = 0; it < sweeps; it++)
        for (int pass = 0; pass < 2; pass++) {
            if ((rc = suhmo_multi_colour_pass(m, phys_of(H, l), has_alpha(H, l), pass, HST(s), push))) return rc;
            H->phi_ver[l]++;
            if (!push && (rc = hier_ff(H, l, SUHMO_F_PHI, -1, false, HST(s), V.part ? pass : -1))) return rc;
        }
    if (sweeps > 0 && (rc = suhmo_multi_fill_ghosts(m, SUHMO_F_PHI, 1, HST(s)))) return rc;                        // :757-759
    if (sweeps > 0 && push) H->ff_seen[l] = H->phi_ver[l];          // every pass pushed its side cells: the ghosts are current
    return 0;
}
int hier_level_residual(suhmo_hier *H, int l, suhmo_stream_t s)   // residualI: RES
{
    if (l == 0) return suhmo_level_residual(base_of(H), 0, s);
    int rc;
    suhmo_multi m;
    if ((rc = hier_ff(H, l, SUHMO_F_PHI, -1, false, HST(s))) || (rc = multi_of(H, l, HST(s), m))) return rc;
    return suhmo_multi_apply(m, phys_of(H, l), has_alpha(H, l), 1, HST(s));            // (owner computes: m = this rank's boxes)
}
int hier_axby(suhmo_hier *H, int l, int dst, int x, int y, double a, double b, suhmo_stream_t s)
{
    if (l == 0) return suhmo_level_axby(base_of(H), 0, dst, x, y, a, b, s);
    int rc;
    suhmo_multi m;
    if ((rc = ensure_field(H, l, dst)) || (rc = ensure_field(H, l, x)) || (rc = ensure_field(H, l, y)) || (rc = multi_of(H, l, HST(s), m))) return rc;
    return suhmo_multi_axby(m, dst, x, y, a, b, HST(s));
}
int hier_copy(suhmo_hier *H, int l, int dst, int src, suhmo_stream_t s)
{
    int rc;
    if ((rc = ensure_field(H, l, dst)) || (rc = ensure_field(H, l, src))) return rc;
    if (dst == SUHMO_F_PHI) { for (suhmo_level *L : H->lev[l].box) L->d[0].phi_fresh = 0; H->phi_ver[l]++; }
    if (dst == SUHMO_F_PHI && l == 0) H->phi_shadow_fresh = false;
    if (l == 0) H->base_full_ver++;
    if (l == 0) {
        suhmo_level *L = base_of(H);
        HIPCHK(hipMemcpyAsync(L->d[0].fp.f[dst], L->d[0].fp.f[src], L->d[0].elems * sizeof(double), hipMemcpyDeviceToDevice, HST(s)));
        return 0;
    }
    suhmo_multi m;
    if ((rc = multi_of(H, l, HST(s), m))) return rc;
    return suhmo_multi_copy(m, dst, src, HST(s));
}
// head of level l: its coarse-fine ghosts from level l-1
int cf_phi(suhmo_hier *H, int l, suhmo_stream_t s)
{
    if (l == 0) return 0;
    if (H->cf_seen[l][0] == H->phi_ver[l] && H->cf_seen[l][1] == H->phi_ver[l - 1]) return 0;       // the ghosts are current
    if (H->merged_launches && !H->lev[l].part && H->ff_seen[l] != H->phi_ver[l]) {
        // the side ghosts between the boxes are stale as well (whoever reads the head next asks for them): both in one launch
        int rc = hier_cf_ff(H, l, SUHMO_F_PHI, SUHMO_F_PHI, -1, -1, false, HST(s));
        if (!rc) { H->cf_seen[l][0] = H->phi_ver[l]; H->cf_seen[l][1] = H->phi_ver[l - 1]; H->ff_seen[l] = H->phi_ver[l]; }
        return rc;
    }
    int rc = hier_cf(H, l, SUHMO_F_PHI, SUHMO_F_PHI, HST(s));
    if (!rc) { H->cf_seen[l][0] = H->phi_ver[l]; H->cf_seen[l][1] = H->phi_ver[l - 1]; }
    return rc;
}

// ---- several levels per launch (option merged_launches; this process holds every box and the whole of level 0)
inline bool levels_mergeable(const suhmo_hier *H) { return H->merged_launches && !H->part && !dist_base(H) && H->nlev <= SUHMO_LVMAX + 1; }
// coarse-fine and fine-fine side ghosts of the head of the levels llo .. lhi (>= 1) that are stale: ONE launch
int ghosts_levels(suhmo_hier *H, int llo, int lhi, suhmo_stream_t s)
{
    LvGhosts g;
    memset(&g, 0, sizeof(g));
    int rc, nb = 0;
    for (int l = std::max(1, llo); l <= lhi; l++) {
        HLev &V = H->lev[l];
        const bool cf_ok = H->cf_seen[l][0] == H->phi_ver[l] && H->cf_seen[l][1] == H->phi_ver[l - 1], ff_ok = H->ff_seen[l] == H->phi_ver[l];
        if (cf_ok && ff_ok) continue;
        if ((rc = ensure_field(H, l, SUHMO_F_PHI)) || (rc = ensure_field(H, l - 1, SUHMO_F_PHI)) || (rc = refresh_tables(H, l, HST(s)))) return rc;
        if (l - 1 >= 1 && (rc = refresh_tables(H, l - 1, HST(s)))) return rc;
        const int q = g.n++;
        g.cf[q] = V.cf.d; g.ncf[q] = (int)V.cf.n; g.nbcf[q] = (int)g1(V.cf.n).x;
        g.ff[q] = V.ff_side.d; g.nff[q] = (int)V.ff_side.n;
        g.nb[q] = g.nbcf[q] + (int)g1(V.ff_side.n).x;
        g.ftab[q] = V.d_fp; g.ctab[q] = l - 1 >= 1 ? H->lev[l - 1].d_fp : nullptr; g.use_base[q] = l - 1 == 0;
        nb += g.nb[q];
        H->cf_seen[l][0] = H->phi_ver[l]; H->cf_seen[l][1] = H->phi_ver[l - 1]; H->ff_seen[l] = H->phi_ver[l];
    }
    if (nb > 0) {
        SUHMO_TIME("QuadCFInterp::coarseFineInterp + exchange");
        hipLaunchKernelGGL(k_cf_ff_lv, dim3(nb), dim3(256), 0, HST(s), g, base_of(H)->d[0].fp, (int)SUHMO_F_PHI);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// the boxes of the levels lhi, lhi - 1, .. llo (>= 1) as one launch's table; mode_top for lhi, mode_rest for the others
int levels_boxes(suhmo_hier *H, int llo, int lhi, int mode_top, int mode_rest, suhmo_stream_t s, suhmo_lvboxes &lv)
{
    memset(&lv, 0, sizeof(lv));
    int rc;
    for (int l = lhi; l >= std::max(1, llo); l--) {
        suhmo_multi m;
        if ((rc = multi_of(H, l, HST(s), m))) return rc;
        const int q = lv.n++;
        lv.dv[q] = m.dv; lv.fp[q] = m.fp; lv.nbox[q] = m.nbox; lv.mode[q] = l == lhi ? mode_top : mode_rest;
        lv.maxnx = std::max(lv.maxnx, m.maxnx); lv.maxny = std::max(lv.maxny, m.maxny);
    }
    return 0;
}
// L(phi) and the residual of level 0 inside a composite residual: as it is (left behind by the cycle's last launch), on the rectangles the
// average from level 1 changed, or over the whole level
int base_apply_residual(suhmo_hier *H, bool whole_level_follows, suhmo_stream_t s)
{
    int rc;
    HLev &V1 = H->lev[1];
    if (!whole_level_follows && H->base_fused_ver == H->base_full_ver)
        { rc = 0; H->n_fused_residual++; }                                // L(phi) and rhs - L(phi) of the head as it is: written by the cycle's last launch
    else if (H->incremental && whole_level_follows && H->base_res_seen == H->base_full_ver)
        { rc = suhmo_apply_and_residual_rects(base_of(H), 0, V1.dirty0.d, (int)V1.dirty0.n, V1.dirty_w, V1.dirty_h, HST(s)); H->n_incr_residual++; }   // only what the average changed
    else rc = suhmo_apply_and_residual(base_of(H), 0, HST(s));
    H->base_res_seen = whole_level_follows ? 0 : H->base_full_ver;       // (the solve loop's evaluation is the one the next cycle can build on)
    return rc;
}
// RES of the levels llo .. lhi: level lhi's own residual (residualI), the composite residual with the reflux from the level above on the others
// (what hier_level_residual(lhi) and composite_residual(lhi), .., composite_residual(llo + 1) leave) in ONE launch per kind: ghosts, operator on
// the levels of boxes, [level 0], refluxes.  Nothing of one level's part reads what another's writes (ghosts from valid cells; L(phi) from the
// level's own head; a reflux adds fluxes of two heads to L(phi) of its coarse cells).
int levels_residual(suhmo_hier *H, int lhi, int llo, bool whole_level_follows, suhmo_stream_t s, bool average_down = false)
{
    int rc;
    if ((rc = ghosts_levels(H, llo, lhi, s))) return rc;
    for (int l = std::max(1, llo); l < lhi; l++) if ((rc = ensure_field(H, l, SUHMO_F_LPHI))) return rc;
    suhmo_lvboxes lv;
    if ((rc = levels_boxes(H, llo, lhi, 1, 3, s, lv)) || (rc = suhmo_levels_apply(lv, phys_of(H, lhi), has_alpha(H, lhi), HST(s)))) return rc;
    if (llo == 0 && (rc = base_apply_residual(H, whole_level_follows, s))) return rc;
    SUHMO_TIME("VCAMRNonLinearPoissonOp::reflux");
    LvReflux r;
    memset(&r, 0, sizeof(r));
    int nb = 0;
    for (int l = lhi; l > llo; l--) {
        HLev &V = H->lev[l];
        if (!V.targets.n) continue;
        if ((rc = refresh_tables(H, l, HST(s))) || (l - 1 >= 1 && (rc = refresh_tables(H, l - 1, HST(s))))) return rc;
        const DV &vc = H->lev[l - 1].box[0]->d[0].v;
        const int q = r.n++;
        r.tg[q] = V.targets.d; r.ntg[q] = (int)V.targets.n; r.nb[q] = (int)g1(V.targets.n).x; r.faces[q] = V.faces.d;
        r.ftab[q] = V.d_fp; r.fdv[q] = V.d_dv; r.ctab[q] = l - 1 >= 1 ? H->lev[l - 1].d_fp : nullptr; r.use_base[q] = l - 1 == 0;
        r.dxc[q] = vc.dx; r.dyc[q] = vc.dy; r.beta[q] = vc.beta;
        nb += r.nb[q];
    }
    AvgPart av;
    memset(&av, 0, sizeof(av));
    if (average_down) {                                    // AMRRestrictS of the residual of level lhi rides along (hier_avg(H, lhi, RES, RES, 0))
        HLev &V = H->lev[lhi];
        CoarseArgs ca;
        if ((rc = refresh_tables(H, lhi, HST(s))) || (rc = coarse_args(H, lhi - 1, HST(s), ca))) return rc;
        av.e = V.avg.d; av.n = (int)V.avg.n; av.ftab = V.d_fp; av.fdv = V.d_dv; av.ctab = ca.tab; av.cdv = ca.dv; av.use_base = ca.use_base;
        av.gx = (V.avg_w + 63) / 64; av.gy = (V.avg_h + 3) / 4; av.nb0 = nb;
        nb += av.gx * av.gy * av.n;
    }
    if (nb > 0) hipLaunchKernelGGL(k_reflux_lv, dim3(nb), dim3(256), 0, HST(s), r, base_of(H)->d[0].fp, base_of(H)->d[0].fp, (int)SUHMO_F_RES, 1, av, base_of(H)->d[0].v);
    HIPCHK(hipGetLastError());
    return 0;
}

// cell-centred gradient of level l (compGradientCC) with its domain-side ghosts
int hier_grad_cc(suhmo_hier *H, int l, suhmo_stream_t s)
{
    if (l == 0) return suhmo_grad_cc(base_of(H), 0, HST(s));
    int rc;
    suhmo_multi m;
    if ((rc = hier_ff(H, l, SUHMO_F_PHI, -1, false, HST(s)))) return rc;              // UpdateOperator :47
    if ((rc = ensure_field(H, l, SUHMO_F_GRADX)) || (rc = ensure_field(H, l, SUHMO_F_GRADY)) || (rc = ensure_field(H, l, SUHMO_F_RE)) || (rc = multi_of(H, l, HST(s), m))) return rc;
    return suhmo_multi_grad_cc(m, phys_of(H, l).use_mask_gradients, HST(s));
}
// UpdateOperator of level l >= 1 with its coarser level (src/VCAMRNonLinearPoissonOp.cpp:34-64, src/AmrHydro.cpp:1415-1539)
int hier_update_operator(suhmo_hier *H, int l, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::UpdateOperator(AMR)");
    int rc;
    if (l - 1 >= 1 && levels_mergeable(H)) {
        // the gradients of this level and of the coarser one: their ghosts in one launch, the two gradients in one launch
        suhmo_lvboxes lv;
        if ((rc = ghosts_levels(H, l - 1, l, s))) return rc;
        for (int q = l - 1; q <= l; q++) for (int f : {SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_RE}) if ((rc = ensure_field(H, q, f))) return rc;
        if ((rc = levels_boxes(H, l - 1, l, 0, 0, s, lv)) || (rc = suhmo_levels_grad_cc(lv, phys_of(H, l).use_mask_gradients, HST(s)))) return rc;
        if ((rc = hier_cf_ff(H, l, SUHMO_F_GRADX, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_GRADY, true, HST(s)))) return rc;
        suhmo_multi m;
        if ((rc = multi_of(H, l, HST(s), m))) return rc;
        return suhmo_multi_re_bcoef(m, phys_of(H, l), HST(s));
    }
    if ((rc = cf_phi(H, l - 1, s))) return rc;                    // the coarser level's own coarse-fine ghosts (its gradient reads them)
    if ((rc = hier_grad_cc(H, l, s))) return rc;
    // the coarse gradient is read by the coarse-fine interpolation below and by nothing else: on level 0 only the cells those
    // stencils touch are evaluated (a pass over the whole level otherwise, 86 us at 4096^2)
    if (l - 1 == 0 && H->incremental) { rc = suhmo_grad_cc_list(base_of(H), 0, H->lev[1].gcells.d, (int)H->lev[1].gcells.n, HST(s)); H->n_sparse_grad++; }
    else rc = hier_grad_cc(H, l - 1, s);
    if (rc) return rc;
    if (H->merged_launches && !H->lev[l].part) {
        if ((rc = hier_cf_ff(H, l, SUHMO_F_GRADX, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_GRADY, true, HST(s)))) return rc;
    } else {
        if ((rc = hier_cf(H, l, SUHMO_F_GRADX, SUHMO_F_GRADX, HST(s), SUHMO_F_GRADY, SUHMO_F_GRADY))) return rc;
        if ((rc = hier_ff(H, l, SUHMO_F_GRADX, SUHMO_F_GRADY, true, HST(s)))) return rc;  // lvlgradH.exchange() src/AmrHydro.cpp:1490
    }
    suhmo_multi m;
    if ((rc = multi_of(H, l, HST(s), m))) return rc;
    return suhmo_multi_re_bcoef(m, phys_of(H, l), HST(s));
}
// RES of level l-1 = rhs - [applyOpI(phi) + reflux from level l]; LPHI of level l-1 keeps the plain L(phi)
// whole_level_follows: called from inside a V-cycle (what follows turns RES of level l-1 into a FAS right-hand side); false: the
// residual evaluation of the solve loop
int composite_residual(suhmo_hier *H, int l, suhmo_stream_t s, bool whole_level_follows = true)
{
    int rc;
    // one pass writes LPHI and rhs - LPHI; the cells next to the coarse-fine faces then get rhs - (LPHI + flux mismatch): the
    // values of copy, reflux, axby(RES, RHS, -1, 1) over the whole level, without two of its three passes
    if ((rc = cf_phi(H, l - 1, s))) return rc;
    if (l - 1 == 0) rc = base_apply_residual(H, whole_level_follows, s);
    else {
        suhmo_multi m;
        if ((rc = hier_ff(H, l - 1, SUHMO_F_PHI, -1, false, HST(s))) || (rc = ensure_field(H, l - 1, SUHMO_F_LPHI)) || (rc = multi_of(H, l - 1, HST(s), m))) return rc;
        rc = suhmo_multi_apply(m, phys_of(H, l - 1), has_alpha(H, l - 1), 3, HST(s));
    }
    if (rc) return rc;
    if ((rc = cf_phi(H, l, s))) return rc;
    return hier_reflux(H, l, SUHMO_F_RES, HST(s), 1);
}
int vcycle_amr(suhmo_hier *H, int l, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    if (l == 0) { H->phi_shadow_fresh = false; H->phi_ver[0]++; H->base_full_ver++; return suhmo_level_vcycle(base_of(H), sp, s); }
    int rc;
    if ((rc = cf_phi(H, l, s))) return rc;
    if (sp->bcoeff_otf && (rc = hier_update_operator(H, l, s))) return rc;
    if ((rc = hier_gsrb(H, l, sp->num_smooth, s, true))) return rc;                           // relaxNF
    if ((rc = hier_avg(H, l, SUHMO_F_PHI, SUHMO_F_PHI, 0, 0.0, HST(s)))) return rc;           // AMRRestrictS(skip_res)
    if (levels_mergeable(H)) { if ((rc = levels_residual(H, l, l - 1, true, s, true))) return rc; }      // (the average of the residual rides in the reflux launch)
    else {
        if ((rc = cf_phi(H, l, s))) return rc;
        if ((rc = hier_level_residual(H, l, s))) return rc;
        if ((rc = composite_residual(H, l, s))) return rc;
        if ((rc = hier_avg(H, l, SUHMO_F_RES, SUHMO_F_RES, 0, 0.0, HST(s)))) return rc;
    }
    // the right-hand side of level l-1 is set aside while its FAS problem runs: two canvases trade places on level 0 (its
    // pointers travel by value), a copy on a level of boxes (their pointers sit in a device table)
    SwapGuard rhs_aside;                                   // (trades back on every way out of this scope)
    if (l - 1 == 0) {
        rhs_aside.arm(&base_of(H)->d[0].fp.f[SUHMO_F_RHS], &base_of(H)->d[0].fp.f[SUHMO_F_RHS0]);
        if ((rc = hier_axby(H, 0, SUHMO_F_RHS, SUHMO_F_RES, SUHMO_F_LPHI, 1.0, 1.0, s))) return rc;
        if (dist_base(H) && (rc = suhmo_level_exchange(base_of(H), 0, SUHMO_F_RHS, s))) return rc;      // rank strips: rhs halo rows (relaxed redundantly)
        // the head of level 0 before its FAS problem: level 1 reads the correction of level 0 only through the windows of its boxes,
        // so only those cells are kept (and only their differences formed)
        if ((rc = hier_window_save(H, l, SUHMO_F_PHI, HST(s)))) return rc;
    } else {
        // a level of boxes: its right-hand side set aside, the FAS right-hand side formed, a copy of its head kept -- one launch
        // (copy RHS0 <- RHS, axby RHS <- RES + LPHI, copy PHIOLD <- PHI: the same expressions on the same operands)
        suhmo_multi mc;
        for (int f : {SUHMO_F_RHS0, SUHMO_F_PHIOLD, SUHMO_F_LPHI}) if ((rc = ensure_field(H, l - 1, f))) return rc;
        if ((rc = multi_of(H, l - 1, HST(s), mc)) || (rc = suhmo_multi_fas_enter(mc, HST(s)))) return rc;
    }
    if (l - 1 == 0) {
        // level 0's own V-cycle runs against the FAS right-hand side; its last launch is asked to leave L(phi) and TRUE rhs - L(phi) behind
        // (the true right-hand side waits on the second canvas): what the solve loop's residual evaluation computes next
        suhmo_level *B = base_of(H);
        B->resout_req = 3; B->resout_rhs = B->d[0].fp.f[SUHMO_F_RHS0]; B->resout_done = 0;
        rc = vcycle_amr(H, 0, sp, s);
        if (!rc && B->resout_done) H->base_fused_ver = H->base_full_ver;
        B->resout_req = 0; B->resout_rhs = nullptr; B->resout_done = 0;
        if (rc) return rc;
    } else if ((rc = vcycle_amr(H, l - 1, sp, s))) return rc;
    if (l - 1 == 0) { rhs_aside.back(); rc = hier_prolong2(H, l, SUHMO_F_PHI, HST(s), true); }                // AMRProlongS_2 of phi - phi_saved
    else {                                                                                    // RHS <- RHS0, CORR <- PHI - PHIOLD: one launch
        if ((rc = ensure_field(H, l - 1, SUHMO_F_CORR))) return rc;
        if (H->merged_launches) rc = hier_prolong2(H, l, SUHMO_F_PHI, HST(s), false, true);      // leaving and prolongation in one launch
        else {
            suhmo_multi mc;
            if ((rc = multi_of(H, l - 1, HST(s), mc)) || (rc = suhmo_multi_fas_leave(mc, HST(s)))) return rc;
            rc = hier_prolong2(H, l, SUHMO_F_CORR, HST(s));
        }
    }
    if (rc) return rc;
    if ((rc = cf_phi(H, l, s))) return rc;
    if ((rc = hier_gsrb(H, l, sp->num_smooth, s, true))) return rc;
    if (H->lev[l].swapped) {                               // (an odd number of odd relaxations: the head goes back to its own canvas by a copy after all)
        suhmo_multi m;
        if ((rc = multi_of(H, l, HST(s), m)) || (rc = suhmo_multi_copy(m, SUHMO_F_PHI2, SUHMO_F_PHI, HST(s)))) return rc;
        swap_head(H, l);
    }
    return 0;
}
}  // namespace
// MAX over the ranks of a value every rank computed on the boxes it owns (through the all-reduce of the base strip)
int suhmo_hier_allreduce_max_(suhmo_hier *H, double *v)
{
    suhmo_level *B = H->lev[0].box[0];
    if (H->world <= 1) return 0;
    if (!B->ar) { suhmo_set_error("hier: the levels are partitioned over the ranks and level 0 has no all-reduce hook"); return -1; }
    return B->ar(B->user, v);
}
namespace {
int check_hier(suhmo_hier *H)
{
    ARG(H && H->nlev >= 1);
    // a V-cycle that failed half way may have left a level with the canvases of its head trading places (swap_head): the head goes back to its
    // own canvas before anything else looks at the level
    for (int l = 1; l < H->nlev; l++)
        if (H->lev[l].swapped) {
            suhmo_multi m;
            int rc;
            HIPCHK(hipSetDevice(H->device));
            HIPCHK(hipDeviceSynchronize());
            if ((rc = multi_of(H, l, nullptr, m)) || (rc = suhmo_multi_copy(m, SUHMO_F_PHI2, SUHMO_F_PHI, nullptr))) return rc;
            HIPCHK(hipDeviceSynchronize());
            swap_head(H, l);
        }
    H->phi_shadow_fresh = false;
    for (int l = 0; l < 8; l++) H->phi_ver[l]++;
    H->base_full_ver++;
    return 0;
}     // every C-ABI entry: the caller may have loaded new data
}  // namespace

// ------------------------------------------------------------------ C-ABI
extern "C" int suhmo_hier_destroy(suhmo_hier_t *H)
{
    if (!H) return 0;
    (void)hipSetDevice(H->device);
    (void)hipDeviceSynchronize();
    if (H->gap) { (void)suhmo_hier_destroy(H->gap); H->gap = nullptr; }
    for (int f = 0; f < SUHMO_F_COUNT; f++) if (H->shadow.f[f]) (void)hipFree(H->shadow.f[f]);
    H->need.release(); H->need_c.release(); H->need_rl.release(); H->cover_full.release();
    if (H->cover_whole) (void)hipFree(H->cover_whole);
    if (H->red_all) (void)hipFree(H->red_all);
    if (H->xs) (void)hipFree(H->xs);
    if (H->xr) (void)hipFree(H->xr);
    for (int l = 0; l < 8; l++) {
        HLev &V = H->lev[l];
        for (suhmo_level *L : V.box) (void)suhmo_level_destroy(L);
        V.ff_side.release(); V.ff_all.release(); V.push.release(); V.pbase.release(); V.cf.release(); V.pwl.release(); V.avg.release(); V.wing.release();
            V.wstart.release(); V.halo.release(); V.hbase.release();
        V.targets.release(); V.faces.release(); V.dirty0.release(); V.gcells.release();
        if (V.winbuf) (void)hipFree(V.winbuf);
        if (V.winold) (void)hipFree(V.winold);
        if (V.d_win) (void)hipFree(V.d_win);
        if (V.d_wing_box) (void)hipFree(V.d_wing_box);
        if (V.d_fp) (void)hipFree(V.d_fp);
        if (V.d_fp_alt) (void)hipFree(V.d_fp_alt);
        if (V.d_dv) (void)hipFree(V.d_dv);
        if (V.d_red) (void)hipFree(V.d_red);
        for (Sync *S : {&V.sy_side[0], &V.sy_side[1], &V.sy_sides, &V.sy_all, &V.sy_cread, &V.sy_win, &V.sy_fface}) S->release();
        V.avg_cov.release(); V.avg_put.release(); V.avg_get.release();
    }
    if (H->ps) (void)hipFree(H->ps);
    if (H->pr) (void)hipFree(H->pr);
    delete H;
    return 0;
}

extern "C" int suhmo_hier_create(suhmo_hier_t **out, const suhmo_level_desc_t *base, int nlev, const int *nbox, const int *boxes)
{
    return suhmo_hier_create_opts(out, base, nlev, nbox, boxes, nullptr);
}
// value of `key` in a "key=value,key=value" list; dflt when absent
static long hier_opt(const char *opts, const char *key, long dflt)
{
    if (!opts) return dflt;
    const size_t n = strlen(key);
    for (const char *p = opts; *p;) {
        while (*p == ',' || *p == ' ') p++;
        if (!strncmp(p, key, n) && p[n] == '=') return atol(p + n + 1);
        while (*p && *p != ',') p++;
    }
    return dflt;
}
extern "C" int suhmo_hier_create_opts(suhmo_hier_t **out, const suhmo_level_desc_t *base, int nlev, const int *nbox, const int *boxes, const char *options)
{
    ARG(out && base && nlev >= 1 && nlev <= 8);
    ARG(nlev == 1 || (nbox && boxes));
    ARG(base->i0 == 0 && (base->nx_global == 0 || base->nx_global == base->nx));
    const bool cut = !(base->j0 == 0 && base->ny == base->ny_global);
    if (cut && (base->ny_global % base->ny || base->j0 % base->ny)) { suhmo_set_error("hier: level 0 must be cut into EQUAL rank strips"); return -1; }
    suhmo_hier *H = new suhmo_hier();
    if (cut) { H->world = base->ny_global / base->ny; H->rank = base->j0 / base->ny; }
    H->shadowed = cut;
    if (hier_opt(options, "shadow", 0) != 0) H->shadowed = true;          // an uncut level 0 read through the shadow path all the same (tests)
    H->push_ghosts = hier_opt(options, "push_ghosts", 1) != 0;
    H->incremental = hier_opt(options, "incremental_residual", 1) != 0;
    H->fused_prolong = hier_opt(options, "fused_prolong", 1) != 0;
    H->merged_launches = hier_opt(options, "merged_launches", 1) != 0;
    H->box_sweeps = hier_opt(options, "box_sweeps", 4) >= 4 ? 4 : 2;
    H->fused_relax = hier_opt(options, "fused_relax", 1) != 0;
    H->part_min_cells = std::max(1L, hier_opt(options, "partition_min_cells", H->part_min_cells));
    H->nlev = nlev; H->device = base->device; H->bc = base->bc; H->base_desc = *base; H->base_desc.boxes = nullptr; H->base_desc.nbox = 0;
    if (options) H->options = options;
    suhmo_level *B = nullptr;
    int rc = suhmo_level_create(&B, base);
    if (rc) { delete H; return rc; }
    H->lev[0].l = 0; H->lev[0].nxd = base->nx; H->lev[0].nyd = base->ny_global; H->lev[0].box.push_back(B);
    H->vglob = B->d[0].v;
    if (H->shadowed) {
        DV &g = H->vglob;
        g.ny = g.nyg; g.j0 = 0; g.rows = g.ny + 2 * g.gy;
        g.ext[0] = g.ext[1] = g.rk[0] = g.rk[1] = 0;
    }
    {   // owner computes: the levels >= 1 are dealt to the ranks -- all of them or none -- when the largest holds at least partition_min_cells
        // cells per rank (per AMR cycle a level's ~26 passes shrink by (1 - 1/W) x 14 us per million cells each and ~28
        // collectives of ~30 us + 8 B x W x the packed cells / the link rate are added: even at ~2.6 M cells per level on 8 ranks)
        long most = 0;
        const int *qq = boxes;
        for (int l = 1; l < nlev; l++) {
            long cells = 0;
            for (int k = 0; k < nbox[l]; k++, qq += 4) cells += (long)(qq[2] - qq[0] + 1) * (qq[3] - qq[1] + 1);
            most = std::max(most, cells);
        }
        H->part = H->world > 1 && nlev > 1 && most >= H->part_min_cells * H->world;
    }
    const int *q = boxes;
    for (int l = 1; l < nlev; l++) {
        HLev &V = H->lev[l];
        V.l = l; V.nxd = H->lev[l - 1].nxd * 2; V.nyd = H->lev[l - 1].nyd * 2;
        if (nbox[l] < 1) { suhmo_set_error("hier: level %d has no box", l); suhmo_hier_destroy(H); return -1; }
        V.b4.assign(q, q + 4 * (size_t)nbox[l]);
        q += 4 * (size_t)nbox[l];
        long cells = 0;
        for (int k = 0; k < nbox[l]; k++) {
            const int *b = &V.b4[4 * k];
            if ((b[0] & 1) || (b[1] & 1) || !(b[2] & 1) || !(b[3] & 1) || b[0] < 0 || b[1] < 0 || b[2] >= V.nxd || b[3] >= V.nyd || b[2] < b[0] || b[3] < b[1]) {
                suhmo_set_error("hier: box %d of level %d is not a coarse-aligned box of the refined domain", k, l); suhmo_hier_destroy(H); return -1; }
            cells += (long)(b[2] - b[0] + 1) * (b[3] - b[1] + 1);
        }
        V.index.build(V.b4, V.nxd, V.nyd);
        {   // disjoint: every cell of every box is found in that box
            long seen = 0;
            for (int k = 0; k < nbox[l]; k++) {
                const int *b = &V.b4[4 * k];
                for (int c = 0; c < 4; c++) {                       // overlap of rectangles shows at a corner of one of them
                    int i = (c & 1) ? b[2] : b[0], j = (c & 2) ? b[3] : b[1];
                    size_t qb = (size_t)(j / V.index.bs) * V.index.nbx + i / V.index.bs;
                    for (int p = V.index.start[qb]; p < V.index.start[qb + 1]; p++) {
                        const int o = V.index.items[p];
                        const int *ob = &V.b4[4 * o];
                        if (o != k && i >= ob[0] && i <= ob[2] && j >= ob[1] && j <= ob[3]) seen = -1;
                    }
                }
                // a box may also cross another without containing a corner: compare against every box sharing a bucket
                for (int by = b[1] / V.index.bs; by <= b[3] / V.index.bs && seen >= 0; by++)
                    for (int bx = b[0] / V.index.bs; bx <= b[2] / V.index.bs; bx++) {
                        size_t qb = (size_t)by * V.index.nbx + bx;
                        for (int p = V.index.start[qb]; p < V.index.start[qb + 1]; p++) {
                            const int o = V.index.items[p];
                            const int *ob = &V.b4[4 * o];
                            if (o != k && std::max(b[0], ob[0]) <= std::min(b[2], ob[2]) && std::max(b[1], ob[1]) <= std::min(b[3], ob[3])) seen = -1;
                        }
                    }
                if (seen < 0) break;
            }
            if (seen < 0) { suhmo_set_error("hier: boxes of level %d overlap", l); suhmo_hier_destroy(H); return -1; }
        }
        for (int k = 0; k < nbox[l]; k++) {
            const int *b = &V.b4[4 * k];
            suhmo_level_desc_t d = *base;
            d.nx = b[2] - b[0] + 1; d.ny = b[3] - b[1] + 1;
            d.i0 = b[0]; d.nx_global = V.nxd; d.j0 = b[1]; d.ny_global = V.nyd;
            d.dx = base->dx / (double)(1 << l); d.dy = base->dy / (double)(1 << l);
            d.nbox = 0; d.boxes = nullptr; d.max_box = std::max(d.nx, d.ny);
            d.halo_rows = 1; d.patch_j0 = 0; d.patch_ny = 0;
            suhmo_level *L = nullptr;
            rc = suhmo_level_create_(&L, &d, H->part);           // (owner computes: geometry first, storage once the plans say which boxes this rank holds)
            if (rc) { suhmo_hier_destroy(H); return rc; }
            L->gsrb_variant = 0; L->gsrb_tile = 0;               // in-place colour passes: the canvases of a box never move
            V.box.push_back(L);
        }
        (void)cells;
    }
    static const int need[] = {SUHMO_F_LPHI, SUHMO_F_GRADX, SUHMO_F_GRADY, SUHMO_F_RE, SUHMO_F_RHS0, SUHMO_F_PHIOLD, SUHMO_F_CORR};
    for (int l = 1; l < nlev; l++) if ((rc = part_setup(H, l))) { suhmo_hier_destroy(H); return rc; }
    for (int l = 1; l < nlev; l++) if ((rc = build_plans(H, l))) { suhmo_hier_destroy(H); return rc; }
    for (int l = 1; l < nlev && H->part; l++) {              // storage for the boxes this rank owns or mirrors; every other box stays a stub
        HLev &V = H->lev[l];
        for (size_t k = 0; k < V.box.size(); k++)
            if (V.held[k]) { V.held_boxes++; if ((rc = suhmo_level_materialize_(V.box[k]))) { suhmo_hier_destroy(H); return rc; } }
    }
    for (int l = 0; l < nlev; l++) for (int f : need) if ((rc = ensure_field(H, l, f))) { suhmo_hier_destroy(H); return rc; }
    for (int l = 1; l < nlev; l++) if ((rc = refresh_tables(H, l, nullptr))) { suhmo_hier_destroy(H); return rc; }
    // SUHMO_F_COVER: 1 under a finer level, 0 elsewhere
    for (int l = 0; l < nlev; l++) for (suhmo_level *L : H->lev[l].box) if (!L->stub && (rc = suhmo_level_set_value(L, 0, SUHMO_F_COVER, 0.0, nullptr))) { suhmo_hier_destroy(H); return rc; }
    for (int l = 1; l < nlev; l++) if ((rc = hier_avg(H, l, SUHMO_F_COVER, SUHMO_F_COVER, 1, 1.0, nullptr))) { suhmo_hier_destroy(H); return rc; }
    if (H->shadowed && nlev > 1) {                          // COVER of the whole level 0 (geometry only): the moulin integrals run over all of it
        const size_t welems = (size_t)H->vglob.P * (size_t)(H->vglob.rows + 1);
        if (hipMalloc(&H->cover_whole, welems * sizeof(double)) != hipSuccess) { suhmo_set_error("field allocation failed"); suhmo_hier_destroy(H); return -2; }
        HIPCHK(hipMemset(H->cover_whole, 0, welems * sizeof(double)));
        FP whole{}; whole.f[SUHMO_F_COVER] = H->cover_whole;
        HLev &V = H->lev[1];
        if (H->cover_full.n) {
            int w = 0, h = 0;
            std::vector<RectEnt> tmp(H->cover_full.n);
            HIPCHK(hipMemcpy(tmp.data(), H->cover_full.d, tmp.size() * sizeof(RectEnt), hipMemcpyDeviceToHost));
            for (auto &e : tmp) { w = std::max(w, e.w); h = std::max(h, e.h); }
            dim3 grd((w + 63) / 64, (h + 3) / 4, (unsigned)H->cover_full.n);
            hipLaunchKernelGGL(k_avg, grd, dim3(64, 4), 0, nullptr, H->cover_full.d, V.d_fp, V.d_dv, (int)SUHMO_F_COVER, (const FP *)nullptr, (const DV *)nullptr,
                               whole, H->vglob, 1, (int)SUHMO_F_COVER, 1, 1.0);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipDeviceSynchronize());
    *out = H;
    return 0;
}
// internal interface for the time step (suhmo_step.hip)
int suhmo_hier_nlev_(const suhmo_hier *H) { return H->nlev; }
const std::vector<suhmo_level *> &suhmo_hier_boxes_(suhmo_hier *H, int l) { return H->lev[l].box; }
int suhmo_hier_device_(const suhmo_hier *H) { return H->device; }
int suhmo_hier_ff_(suhmo_hier *H, int l, int f0, int f1, bool corners, hipStream_t st) { return hier_ff(H, l, f0, f1, corners, st); }
int suhmo_hier_cf_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st) { return hier_cf(H, l, ff, fc, st); }
int suhmo_hier_cf2_(suhmo_hier *H, int l, int ff0, int fc0, int ff1, int fc1, hipStream_t st) { return hier_cf(H, l, ff0, fc0, st, ff1, fc1); }
int suhmo_hier_pwl_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st) { return hier_pwl(H, l, ff, fc, st); }
int suhmo_hier_avg_(suhmo_hier *H, int l, int ff, int fc, hipStream_t st) { return hier_avg(H, l, ff, fc, 0, 0.0, st); }
int suhmo_hier_multi_(suhmo_hier *H, int l, hipStream_t st, suhmo_multi *m) { return multi_of(H, l, st, *m); }
int suhmo_hier_ensure_(suhmo_hier *H, int l, int field) { return ensure_field(H, l, field); }
void suhmo_hier_invalidate_(suhmo_hier *H)
{
    H->phi_shadow_fresh = false;
    for (int l = 0; l < 8; l++) H->phi_ver[l]++;
    H->base_full_ver++;
    if (H->gap) suhmo_hier_invalidate_(H->gap);
}
bool suhmo_hier_partitioned_(const suhmo_hier *H) { return H->part; }
int suhmo_hier_world_(const suhmo_hier *H) { return H->world; }
void suhmo_hier_owned_(const suhmo_hier *H, int l, int *first, int *n)
{
    const HLev &V = H->lev[l];
    if (V.part) { *first = V.b0; *n = V.nown; } else { *first = 0; *n = (int)V.box.size(); }
}
int suhmo_hier_allgather_(suhmo_hier *H, const double *send, long count, double *recv, hipStream_t st)
{
    if (!H->ag) { suhmo_set_error("hier: no all-gather is attached (suhmo_hier_attach_rccl / suhmo_hier_set_allgather)"); return -1; }
    H->part_gathers++;
    return H->ag(H->ag_user, send, count, recv, (suhmo_stream_t)st);
}
const double *suhmo_hier_base_cover_(suhmo_hier *H, DV *whole)
{
    if (!dist_base(H)) return nullptr;
    *whole = H->vglob;
    return H->cover_whole;
}
int suhmo_hier_gap_(suhmo_hier *H, const suhmo_model_params_t *mp, double dt, suhmo_hier **gap)
{
    if (H->gap && H->gap_dt != dt) {                       // a new time step size: beta = dt diffFactor of every operator; boxes, plans and tables stay
        for (int l = 0; l < H->nlev; l++)
            for (suhmo_level *L : H->gap->lev[l].box) { int rc = suhmo_level_set_alpha_beta(L, 1.0, dt * mp->diffFactor); if (rc) return rc; }
        HIPCHK(hipDeviceSynchronize());
        for (int l = 1; l < H->nlev; l++) {                // the device copies of the boxes' views carry beta: uploaded again at the next use
            HLev &V = H->gap->lev[l];
            if (V.d_dv) { (void)hipFree(V.d_dv); V.d_dv = nullptr; }
            if (V.d_red) { (void)hipFree(V.d_red); V.d_red = nullptr; }
            V.h_fp.clear();
        }
        H->gap->vglob.beta = H->gap->lev[0].box[0]->d[0].v.beta;
        H->gap_dt = dt;
    }
    if (!H->gap) {
        suhmo_level_desc_t d = H->base_desc;
        const suhmo_level *B = H->lev[0].box[0];
        d.boxes = B->boxes.data(); d.nbox = (int)(B->boxes.size() / 4);
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { d.bc.type[a][b] = 1; d.bc.value[a][b] = 0.0; }
        d.phys.use_NL = 0; d.alpha = 1.0; d.beta = dt * mp->diffFactor;
        std::vector<int> nbox(H->nlev, 0), flat;
        for (int l = 1; l < H->nlev; l++) { nbox[l] = (int)H->lev[l].box.size(); flat.insert(flat.end(), H->lev[l].b4.begin(), H->lev[l].b4.end()); }
        int rc = suhmo_hier_create_opts(&H->gap, &d, H->nlev, nbox.data(), flat.data(), H->options.c_str()); if (rc) return rc;
        H->gap_dt = dt;
        H->gap->ag = H->ag; H->gap->ag_user = H->ag_user;                                  // same strips, same ranks
        { suhmo_level *G0 = H->gap->lev[0].box[0]; G0->ex = B->ex; G0->ar = B->ar; G0->ar2 = B->ar2; G0->ard = B->ard; G0->user = B->user;
            G0->ex_begin = B->ex_begin; G0->ex_end = B->ex_end; G0->ipc = B->ipc;
          G0->ag = B->ag; G0->ag_user = B->ag_user; G0->agg_min_cells = B->agg_min_cells; if ((rc = suhmo_agg_setup(G0))) return rc; }
        for (int l = 0; l < H->nlev; l++)
            for (suhmo_level *L : H->gap->lev[l].box) if (!L->stub && (rc = suhmo_level_set_value(L, 0, SUHMO_F_ACOEF, 1.0, nullptr))) return rc;   // aCoeff_GH :1820-1828
    }
    *gap = H->gap;
    return 0;
}

extern "C" int suhmo_hier_set_option(suhmo_hier_t *H, const char *key, long value)
{
    ARG(H && key);
    if (!strcmp(key, "incremental_residual")) {      // 0: every composite residual / coarse gradient over the whole of level 0 (A/B runs, tests)
        H->incremental = value != 0;
        H->base_res_seen = 0;
        if (H->gap) return suhmo_hier_set_option(H->gap, key, value);
        return 0;
    }
    if (!strcmp(key, "fused_prolong")) { H->fused_prolong = value != 0; if (H->gap) return suhmo_hier_set_option(H->gap, key, value); return 0; }
    if (!strcmp(key, "merged_launches")) { H->merged_launches = value != 0; if (H->gap) return suhmo_hier_set_option(H->gap, key, value); return 0; }
    if (!strcmp(key, "box_sweeps")) { H->box_sweeps = value >= 4 ? 4 : 2; if (H->gap) return suhmo_hier_set_option(H->gap, key, value); return 0; }
    if (!strcmp(key, "fused_relax")) { H->fused_relax = value != 0; if (H->gap) return suhmo_hier_set_option(H->gap, key, value); return 0; }
    if (!strcmp(key, "push_ghosts")) {
        H->push_ghosts = value != 0;
        for (int l = 0; l < 8; l++) H->ff_seen[l] = 0;
        if (H->gap) return suhmo_hier_set_option(H->gap, key, value);
        return 0;
    }
    suhmo_set_error("unknown hierarchy option '%s' (push_ghosts, incremental_residual; shadow is a creation option of suhmo_hier_create_opts)", key);
    return -1;
}
extern "C" int suhmo_hier_get_option(const suhmo_hier_t *H, const char *key, long *value)
{
    ARG(H && key && value);
    if (!strcmp(key, "push_ghosts")) { *value = H->push_ghosts; return 0; }
    if (!strcmp(key, "fused_prolong")) { *value = H->fused_prolong; return 0; }
    if (!strcmp(key, "merged_launches")) { *value = H->merged_launches; return 0; }
    if (!strcmp(key, "box_sweeps")) { *value = H->box_sweeps; return 0; }
    if (!strcmp(key, "fused_relax")) { *value = H->fused_relax; return 0; }
    if (!strcmp(key, "fused_relax_launches")) { *value = H->n_fused_relax + (H->gap ? H->gap->n_fused_relax : 0); return 0; }
    if (!strcmp(key, "incremental_residual")) { *value = H->incremental; return 0; }
    if (!strcmp(key, "shadow")) { *value = H->shadowed; return 0; }
    if (!strcmp(key, "partition_min_cells")) { *value = H->part_min_cells; return 0; }
    if (!strcmp(key, "incremental_residual_passes")) { *value = H->n_incr_residual; return 0; }
    if (!strcmp(key, "residuals_left_by_relax")) { *value = H->n_fused_residual; return 0; }
    if (!strcmp(key, "sparse_gradient_passes")) { *value = H->n_sparse_grad; return 0; }
    if (!strcmp(key, "partition_gathers")) { *value = H->part_gathers + (H->gap ? H->gap->part_gathers : 0); return 0; }
    if (!strncmp(key, "partitioned_level_", 18) || !strncmp(key, "own_boxes_level_", 16)) {      // e.g. own_boxes_level_2: boxes of level 2 this rank relaxes
        const bool own = key[0] == 'o';
        const int l = atoi(key + (own ? 16 : 18));
        if (l < 0 || l >= H->nlev) { suhmo_set_error("no level %d", l); return -1; }
        const HLev &V = H->lev[l];
        *value = own ? (V.part ? V.nown : (long)V.box.size()) : (V.part ? 1 : 0);
        return 0;
    }
    if (!strcmp(key, "partition_bytes")) { *value = H->part_bytes + (H->gap ? H->gap->part_bytes : 0); return 0; }     // bytes this rank contributed to the partition's collectives
    {   // per level l >= 1: <key>_level_<l>
        static const char *keys[] = {"ghost_exchange_bytes_level_", "held_boxes_level_", "owned_cells_level_", "canvas_bytes_level_", "ghost_exchange_bound_bytes_level_"};
        for (int q = 0; q < 5; q++) {
            const size_t n = strlen(keys[q]);
            if (strncmp(key, keys[q], n)) continue;
            const int l = atoi(key + n);
            if (l < 1 || l >= H->nlev) { suhmo_set_error("no level %d", l); return -1; }
            const HLev &V = H->lev[l];
            if (q == 0) *value = H->side_bytes[l];                                   // what this rank sends per colour-pass exchange (the larger colour)
            else if (q == 1) *value = V.part ? V.held_boxes : (long)V.box.size();
            else if (q == 2) { long c = 0; if (V.part) c = V.owned_cells; else for (size_t k = 0; k < V.box.size(); k++) { const int *b = &V.b4[4 * k];
                c += (long)(b[2] - b[0] + 1) * (b[3] - b[1] + 1); } *value = c; }
            else if (q == 3) { long c = 0; for (suhmo_level *L : V.box) for (int f = 0; f < SUHMO_F_COUNT; f++) if (L->d[0].fp.f[f]) c += (long)L->d[0].elems * 8; *value = c; }
            // 4 sides x 8 B of the owned boxes
            else { long c = 0; for (int k = V.part ? V.b0 : 0; k < (V.part ? V.b0 + V.nown : 0); k++) { const int *b = &V.b4[4 * k];
                c += 8L * 2 * ((b[2] - b[0] + 1) + (b[3] - b[1] + 1)); } *value = c; }
            return 0;
        }
    }
    suhmo_set_error("unknown hierarchy option '%s'", key);
    return -1;
}
extern "C" int suhmo_hier_set_allgather(suhmo_hier_t *H, suhmo_hier_allgather_fn fn, void *user)
{
    ARG(H);
    H->ag = fn; H->ag_user = user;
    if (H->gap) { H->gap->ag = fn; H->gap->ag_user = user; }
    return 0;
}
int suhmo_rccl_allgather_hook(void *user, const double *send, long count, double *recv, suhmo_stream_t s);   // suhmo_rccl.hip; user = the level's Strip
extern "C" int suhmo_hier_attach_rccl(suhmo_hier_t *H)
{
    ARG(H);
    suhmo_level *B = base_of(H);
    if (!B->rccl) { suhmo_set_error("hier: attach the base strip first (suhmo_level_attach_rccl on suhmo_hier_box(H, 0, 0))"); return -1; }
    return suhmo_hier_set_allgather(H, suhmo_rccl_allgather_hook, B->rccl);
}
extern "C" long suhmo_hier_gathers(const suhmo_hier_t *H) { return H ? H->gathers + (H->gap ? H->gap->gathers : 0) : -1; }
extern "C" int suhmo_hier_num_levels(const suhmo_hier_t *H) { return H ? H->nlev : -1; }
extern "C" int suhmo_hier_num_boxes(const suhmo_hier_t *H, int l) { return (H && l >= 0 && l < H->nlev) ? (int)H->lev[l].box.size() : -1; }
extern "C" suhmo_level_t *suhmo_hier_box(suhmo_hier_t *H, int l, int k)
{
    if (!H || l < 0 || l >= H->nlev || k < 0 || k >= (int)H->lev[l].box.size()) return nullptr;
    return H->lev[l].box[k];
}
// which rank holds box k of level l: -1 = every rank (a replicated level, level 0's strip); `held`: this rank keeps storage for it (its own
// box, or a mirror of a neighbour's whose cells its plans read) -- a box that is neither is a stub: suhmo_level_set_field etc. refuse it
extern "C" int suhmo_hier_box_owner(const suhmo_hier_t *H, int l, int k, int *held)
{
    if (!H || l < 0 || l >= H->nlev || k < 0 || k >= (int)H->lev[l].box.size()) return -2;
    const HLev &V = H->lev[l];
    if (held) *held = V.part ? (int)V.held[k] : 1;
    return V.part ? V.owner[k] : -1;
}
extern "C" int suhmo_hier_exchange(suhmo_hier_t *H, int l, int field, int corners, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 0 && l < H->nlev && field >= 0 && field < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(H->device));
    return hier_ff(H, l, field, -1, corners != 0, HST(s));
}
extern "C" int suhmo_hier_cf_interp(suhmo_hier_t *H, int l, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 1 && l < H->nlev && field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(H->device));
    return hier_cf(H, l, field_f, field_c, HST(s));
}
extern "C" int suhmo_hier_pwl_fill(suhmo_hier_t *H, int l, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 1 && l < H->nlev && field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(H->device));
    return hier_pwl(H, l, field_f, field_c, HST(s));
}
extern "C" int suhmo_hier_average(suhmo_hier_t *H, int l, int field_f, int field_c, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 1 && l < H->nlev && field_f >= 0 && field_f < SUHMO_F_COUNT && field_c >= 0 && field_c < SUHMO_F_COUNT);
    HIPCHK(hipSetDevice(H->device));
    return hier_avg(H, l, field_f, field_c, 0, 0.0, HST(s));
}
extern "C" int suhmo_hier_gsrb(suhmo_hier_t *H, int l, int sweeps, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 0 && l < H->nlev && sweeps >= 0);
    HIPCHK(hipSetDevice(H->device));
    return hier_gsrb(H, l, sweeps, s);
}
extern "C" int suhmo_hier_update_operator(suhmo_hier_t *H, int l, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    ARG(l >= 0 && l < H->nlev);
    HIPCHK(hipSetDevice(H->device));
    if (l == 0) return suhmo_level_update_operator(base_of(H), 0, s);
    return hier_update_operator(H, l, s);
}
// composite residual of the hierarchy (RES of every level, covered cells zeroed) and its max norm (AMRNorm :1222-1264)
// the composite residual of all levels and its max norm over the cells no finer level covers (AMRResidual + AMRNorm)
static int hier_residual_(suhmo_hier *H, double *norm, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::AMRResidual");
    int rc;
    const int top = H->nlev - 1;
    if (levels_mergeable(H)) {
        // every level's ghosts, operator and reflux in one launch per kind; the covered cells are zeroed by the pass that takes the first stage
        // of the max norm; ONE launch over all partial maxima and one read-back (a maximum is the same in any order)
        if ((rc = levels_residual(H, top, 0, false, s))) return rc;
        if (!H->red_all) {
            size_t nb = 0;
            for (int l = 1; l <= top; l++) nb += H->lev[l].box.size();
            HIPCHK(hipMalloc(&H->red_all, (64 * nb + 16) * sizeof(double)));
        }
        const double *lists[2];
        int np[2];
        suhmo_lvboxes lv;
        // level 0: a large one is not read a second time for its cover -- its covered rectangles are zeroed by their list (one small launch), then the
        // plain first stage; a small one zeroes as it reduces, like the levels of boxes
        if (base_of(H)->d[0].elems > (1 << 20)) {
            if ((rc = hier_avg(H, 1, SUHMO_F_RES, SUHMO_F_RES, 1, 0.0, HST(s))) || (rc = suhmo_level_norm_max_partials(base_of(H), SUHMO_F_RES, &lists[0], &np[0], HST(s)))) return rc;
        } else if ((rc = suhmo_level_norm_max_cover_partials(base_of(H), SUHMO_F_RES, &lists[0], &np[0], HST(s)))) return rc;
        if ((rc = levels_boxes(H, 1, top, 0, 1, s, lv)) || (rc = suhmo_levels_norm_max_cover_partials(lv, SUHMO_F_RES, H->red_all, &np[1], HST(s)))) return rc;
        lists[1] = H->red_all;
        return norm ? suhmo_norm_max_of_lists(base_of(H), lists, np, 2, norm, HST(s)) : 0;
    }
    if ((rc = cf_phi(H, top, s))) return rc;
    if ((rc = hier_level_residual(H, top, s))) return rc;                                  // AMRResidualNF on the finest level
    for (int l = top; l >= 1; l--) if ((rc = composite_residual(H, l, s, false))) return rc;
    for (int l = top; l >= 1; l--) if ((rc = hier_avg(H, l, SUHMO_F_RES, SUHMO_F_RES, 1, 0.0, HST(s)))) return rc;
    if (norm && H->merged_launches && !H->part && !dist_base(H)) {
        // one process holds everything: the first stages of every level's max norm, then ONE launch over their partial maxima and one read-back
        // (a maximum is the same in any order) instead of a reduction and a read-back per level
        const double *lists[8];
        int np[8];
        if ((rc = suhmo_level_norm_max_partials(base_of(H), SUHMO_F_RES, &lists[0], &np[0], HST(s)))) return rc;
        for (int l = 1; l <= top; l++) {
            suhmo_multi mv;
            if ((rc = multi_of(H, l, HST(s), mv)) || (rc = suhmo_multi_norm_max_partials(mv, SUHMO_F_RES, &lists[l], &np[l], HST(s)))) return rc;
        }
        return suhmo_norm_max_of_lists(base_of(H), lists, np, top + 1, norm, HST(s));
    }
    if (norm) {
        double m = 0.0;
        if ((rc = suhmo_level_norm(base_of(H), 0, SUHMO_F_RES, 0, &m, s))) return rc;
        double mp = 0.0;
        for (int l = 1; l <= top; l++) {
            double a = 0.0;
            suhmo_multi mv;
            if ((rc = multi_of(H, l, HST(s), mv)) || (rc = suhmo_multi_norm_max(mv, base_of(H), SUHMO_F_RES, &a, HST(s)))) return rc;
            if (a > mp) mp = a;
        }
        if (H->part && (rc = suhmo_hier_allreduce_max_(H, &mp))) return rc;     // owner computes: every rank saw its own boxes only
        if (mp > m) m = mp;
        *norm = m;
    }
    return 0;
}
extern "C" int suhmo_hier_residual(suhmo_hier_t *H, double *norm, suhmo_stream_t s)
{
    int rc = check_hier(H); if (rc) return rc;
    HIPCHK(hipSetDevice(H->device));
    return hier_residual_(H, norm, s);
}
extern "C" int suhmo_hier_vcycle(suhmo_hier_t *H, const suhmo_solver_params_t *sp, suhmo_stream_t s)
{
    SUHMO_TIME("AMRFASMultiGrid::VCycle(AMR)");
    int rc = check_hier(H); if (rc) return rc;
    ARG(sp);
    HIPCHK(hipSetDevice(H->device));
    return vcycle_amr(H, H->nlev - 1, sp, s);
}
extern "C" int suhmo_hier_solve(suhmo_hier_t *H, const suhmo_solver_params_t *sp, int *iters, double *hist, suhmo_stream_t s)
{
    SUHMO_TIME("AMRFASMultiGrid::solve(AMR)");
    ARG(sp);
    int rc;
    if (H && H->nlev == 1) return suhmo_level_solve(base_of(H), sp, iters, hist, s);
    // one invalidation at the entry (the caller may have loaded data); inside the loop every writer keeps the version counters, so
    // ghosts interpolated for the residual serve the cycle that follows, and the residual of level 0 is re-evaluated only where the
    // average from level 1 changed its head
    if ((rc = check_hier(H))) return rc;
    HIPCHK(hipSetDevice(H->device));
    double rnorm = 0.0;
    if ((rc = hier_residual_(H, &rnorm, s))) return rc;
    double initial_rnorm = rnorm, norm_last = 2.0 * initial_rnorm;
    int iter = 0;
    if (hist) hist[0] = rnorm;
    bool goNorm = rnorm > sp->norm_thresh, goRedu = rnorm > sp->eps * initial_rnorm, goIter = iter < sp->max_iter;
    bool goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last, goMin = iter < sp->iter_min;
    while (goMin || (goIter && goRedu && goHang && goNorm)) {
        norm_last = rnorm;
        { SUHMO_TIME("AMRFASMultiGrid::VCycle(AMR)"); rc = vcycle_amr(H, H->nlev - 1, sp, s); }
        if (rc) return rc;
        if ((rc = hier_residual_(H, &rnorm, s))) return rc;
        iter++;
        if (hist) hist[iter] = rnorm;
        goNorm = rnorm > sp->norm_thresh; goRedu = rnorm > sp->eps * initial_rnorm; goIter = iter < sp->max_iter;
        goHang = iter < sp->imin || rnorm < (1.0 - sp->hang) * norm_last; goMin = iter < sp->iter_min;
    }
    if (iters) *iters = iter;
    return 0;
}
