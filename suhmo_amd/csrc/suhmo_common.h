// suhmo_common.h -- internal types shared by the HIP translation units of libsuhmo_hip.so.
// gfx950 (MI355X) only; compiled with -ffp-contract=off so that every kernel reproduces
// the reference's Fortran expression association bit for bit ("Parity").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <string>
#include <vector>
#include "../../include/suhmo_hip.h"

#define SUHMO_XOFF 16      // column of cell i = 0 in a canvas row (128-byte aligned)
#define SUHMO_MAXDEPTH 16

// Level canvas geometry + boundary description of one multigrid depth, passed by value to
// kernels.  Canvas element of cell/face (i,j): (j + gy) * P + XOFF + i.  Every field of a
// depth (cell-centred, x-faces, y-faces) uses the same canvas shape.
struct DV {
    int nx, ny;        // cells of this strip at this depth
    int P;             // pitch in doubles (multiple of 16)
    int gy;            // ghost rows below / above
    int rows;          // ny + 2*gy
    int j0, nyg;       // global row of local j = 0; rows of the whole level (colour parity, faces)
    double dx, dy;     // cell size
    double rdx, rdy;   // one/(dx*dx), one/(dy*dy)   (src/VCAMRNonLinearPoissonOpF.ChF:110)
    double fdx, fdy;   // one/dx, one/dy             (util/GradientF.ChF:57)
    double alpha, beta;
    int bct[2][2];     // [dir][side] 0 Dirichlet, 1 Neumann
    double two_v[2][2];// 2.0*value               (DiriBC order 1: 2*value - near)
    double neu[2][2];  // (sign*dx)*value         (NeumBC: near + sign*dx*value)
    int per[2];        // periodic
    int ext[2];        // y side is a rank boundary or a coarse-fine side: ghost rows hold data
    int i0, nxg;       // AMR patch: global column of local i = 0, columns of the whole (refined) domain
    int cfx[2];        // x side is a coarse-fine side: ghost columns hold interpolated data
    int rk[2];         // y side is a RANK boundary (halo rows exchanged with a neighbour); ext && !rk = coarse-fine side
};

struct FP { double *f[SUHMO_F_COUNT]; };

__host__ __device__ __forceinline__ int cidx(const DV &v, int i, int j)
{
    return (j + v.gy) * v.P + SUHMO_XOFF + i;
}

// ---- neighbour values of phi with the physical BC / periodic wrap applied on the fly ----
// mixBCValues (src/AmrHydro.cpp:248-309): ghost = 2*value - near (Dirichlet, order 1) or
// near + sign*dx*value (Neumann); the ghost depends only on the adjacent interior cell, so
// it is recomputed from registers instead of being stored between colour passes.
__device__ __forceinline__ double phiW(const DV &v, const double *__restrict__ p, int idx, int i, double c, bool homog)
{
    if (i > 0 || v.cfx[0]) return p[idx - 1];
    if (v.per[0]) return p[idx + v.nx - 1];
    if (v.bct[0][0] == 0) return (homog ? 0.0 : v.two_v[0][0]) - c;
    return homog ? c : c + v.neu[0][0];
}
__device__ __forceinline__ double phiE(const DV &v, const double *__restrict__ p, int idx, int i, double c, bool homog)
{
    if (i < v.nx - 1 || v.cfx[1]) return p[idx + 1];
    if (v.per[0]) return p[idx - (v.nx - 1)];
    if (v.bct[0][1] == 0) return (homog ? 0.0 : v.two_v[0][1]) - c;
    return homog ? c : c + v.neu[0][1];
}
__device__ __forceinline__ double phiS(const DV &v, const double *__restrict__ p, int idx, int j, double c, bool homog)
{
    if (j > 0 || v.ext[0]) return p[idx - v.P];
    if (v.per[1]) return p[idx + (v.ny - 1) * v.P];
    if (v.bct[1][0] == 0) return (homog ? 0.0 : v.two_v[1][0]) - c;
    return homog ? c : c + v.neu[1][0];
}
__device__ __forceinline__ double phiN(const DV &v, const double *__restrict__ p, int idx, int j, double c, bool homog)
{
    if (j < v.ny - 1 || v.ext[1]) return p[idx + v.P];
    if (v.per[1]) return p[idx - (v.ny - 1) * v.P];
    if (v.bct[1][1] == 0) return (homog ? 0.0 : v.two_v[1][1]) - c;
    return homog ? c : c + v.neu[1][1];
}

// COMPUTENONLINEARTERMS, src/AmrHydroF.ChF:38-65
__device__ __forceinline__ void nl_terms(const suhmo_phys_t &ph, double phi, double B, double Pi,
                                         double zb, double mask, double &nl, double &dnl)
{
    if (!ph.use_NL || mask < 0.0) { nl = 0.0; dnl = 0.0; return; }
    double N = Pi - ph.rho_w_g * (phi - zb);
    nl = -ph.A * B * N * N * N;
    dnl = 3.0 * ph.A * B * 1000.0 * ph.grav * N * N;
    if (ph.cutOffbr > B) {
        nl = nl * (1.0 - (ph.cutOffbr - B) / ph.cutOffbr);
        dnl = dnl * B / ph.cutOffbr;
    }
    if (ph.maxOffbr < B) {
        nl = nl * (1.0 - (ph.maxOffbr - B) / ph.maxOffbr);
        dnl = dnl * B / ph.maxOffbr;
    }
}

// L(phi) at one cell, src/VCAMRNonLinearPoissonOpF.ChF:130-152 / 257-279.  `aterm` is
// alpha*aCoef(i,j) (or alpha alone when alpha == 0 and aCoef is not read).
__device__ __forceinline__ double lofphi_cell(const DV &v, double aterm, double c, double e, double w,
                                              double n, double s, double bxE, double bxW, double byN,
                                              double byS, double nl)
{
    return aterm * c
           - v.beta * (bxE * (e - c) * v.rdx - bxW * (c - w) * v.rdx
                       + byN * (n - c) * v.rdy - byS * (c - s) * v.rdy)
           + nl;
}

// lambda, resetLambda + SUMFACESNL (src/VCAMRNonLinearPoissonOp.cpp:517-528, ...OpF.ChF:591-598)
__device__ __forceinline__ double lambda_cell(const DV &v, double aterm, double bxE, double bxW,
                                              double byN, double byS)
{
    double lam = aterm;
    lam = lam + v.rdx * v.beta * (bxE + bxW);
    lam = lam + v.rdy * v.beta * (byN + byS);
    return lam;
}

struct Depth {
    DV v;
    FP fp;
    size_t elems;      // doubles per canvas
    int nbox;
    int prolong_pending; // the next fused relax adds P(phi_c - phi_c,old) while loading phi (FAS prolongIncrement)
    int rhs_pending;     // the next tile relax forms the FAS right-hand side rhs = res + L(phi) while loading (and PHIOLD, LPHI)
    double *phi_alt;   // second phi canvas: the fused GSRB kernel writes out of place (ping-pong)
    int phi_fresh;     // strips: halo rows of phi (each rank-boundary side) that hold the neighbour's CURRENT values
};

struct VGraph {                 // a V-cycle captured for one set of solver parameters and one state of the phi ping-pong (suhmo_fas.hip)
    int key[4]; hipGraphExec_t exec;
    double *p0[SUHMO_MAXDEPTH], *a0[SUHMO_MAXDEPTH];     // PHI / second canvas of every depth when the cycle starts ...
    double *p1[SUHMO_MAXDEPTH], *a1[SUHMO_MAXDEPTH];     // ... and when it ends (an odd number of out-of-place launches on a depth swaps them)
    double *rhs;                                         // right-hand-side canvas of depth 0 the cycle was captured with
    int rout_req, rout_done, rout_np; const double *rout_rhs;     // the residual its last launch was asked to leave behind (resout_req, resout_rhs) and did
};
struct ProfEv { hipEvent_t a, b; long cells; int restricts; };   // restricts: the launch also did the restriction (RST)

struct suhmo_level {
    int ndepth;
    int stub;                   // geometry only: no canvas, no scratch (a box of a partitioned AMR level held by other ranks; suhmo_hier.hip)
    Depth d[SUHMO_MAXDEPTH];
    suhmo_level_desc_t desc;
    std::vector<int> boxes;     // nbox x 4, global indices, depth 0
    suhmo_phys_t ph;
    int device;
    double *scratch;            // reduction scratch (device)
    double *hscratch;           // pinned host scratch
    double *hscratch_dev;       // its device address: the last kernel of a reduction writes the result and a sequence number there
    unsigned long long hseq;    // and the host polls for the number instead of synchronising the stream (env SUHMO_POLL_READBACK, default 1)
    int poll_readback;
    size_t scratch_elems;
    suhmo_exchange_fn ex;
    suhmo_allreduce_max_fn ar;
    suhmo_allreduce_fn ar2;                            // n values, MAX or SUM, on the host (suhmo_level_set_reduce_hook)
    int (*ard)(void *user, double *dev_values, int n, int op, suhmo_stream_t s);   // the same in place on DEVICE values, enqueued on s (native transport)
    void *user;
    int (*ex_begin)(void *user);                       // optional: open / close a batch of exchanges that
    int (*ex_end)(void *user, suhmo_level *L, suhmo_stream_t s);   // travel as ONE message group (native transport)
    long graph_max_cells;        // V-cycles of levels up to this size are replayed as HIP graphs (env SUHMO_GRAPH_MAX_CELLS, 0 = off)
    std::vector<VGraph> vgraphs; int vgraph_seen[4]; hipStream_t gstream;
    suhmo_level *gap; double gap_dt;   // implicit gap-height operator of the time step (suhmo_step.hip), owned
    void *rccl;                 // native transport state (suhmo_rccl.hip), owned by the level
    int faces_deferred;         // rank strips: UpdateOperator left the halo rows of the depth-0 faces to the message that carries the coarse depths' (suhmo_average_operator_all)
    void *ipc; int ipc_owner;   // peer-direct halo transport (suhmo_ipc.hip): arena and neighbours' mappings; a gap-height handle borrows its level's
    // agglomeration of the coarse depths of a rank strip (suhmo_agg.hip): from depth agg_depth on (0 = none) the cycle runs on `agg`, a
    // handle of the WHOLE level at that depth held by every rank; all-gather transport ag (suhmo_level_set_allgather / attach_rccl)
    long agg_min_cells;         // depths whose strip holds fewer cells are agglomerated (env SUHMO_AGG_MIN_CELLS, default 100000: at 4096^2 cells per strip the two deepest of six depths; 0 = off)
    int agg_depth, agg_world, agg_rank;
    suhmo_level *agg;
    int agg_static_stale;       // A was (re)created after the last coefficient build: its B / Pi / zb / mask / aCoef / faces are gathered before the cycle enters it
    suhmo_allgather_fn ag; void *ag_user;
    double *agg_send, *agg_recv; size_t agg_cap; long agg_gathers;
    // the solve loops ask the cycle to leave the residual of its final phi behind (suhmo_gsrb.hip, RM = 2): resout_req bit 0 RES, bit 1 LPHI
    // too; resout_rhs: the right-hand side it is about (NULL: the level's); resout_armed: set by the cycle around its last relax of depth 0;
    // resout_done: the launch did it (else the caller runs its own pass); read-only option residual_in_relax_launches counts them
    int resout_req, resout_armed, resout_done; const double *resout_rhs; long resout_count;
    int resout_np;              // resout_req bit 2: the launch also left that many partial maxima of |RES| in scratch + 2 (k_norm_final's input)
    long frhs_stream, frhs_tile;   // launches that formed a coarse depth's FAS right-hand side themselves (streaming / tile kernel); read-only options
    int prof_on;
    std::vector<ProfEv> prof;
    int gsrb_variant;           // kernel selection (see suhmo_gsrb.hip); env SUHMO_GSRB_VARIANT
    long fused_min_cells;       // auto mode: use the fused kernel from this many cells (env SUHMO_FUSED_MIN_CELLS)
    int bcoef_fused;            // single-kernel WFlx_level (env SUHMO_BCOEF_FUSED, default 1)
    int bcoef_tile_x;           // its tile width: 62 (64 x 4 threads) or 126 (128 x 2 threads); env SUHMO_BCOEF_TILE_X
    int fused_nt;               // threads per workgroup of the fused kernel: 256 or 64 (env SUHMO_FUSED_NT)
    int fused_restrict;         // the last pre-smoothing launch also restricts (env SUHMO_FUSED_RESTRICT, default 1)
    int fused_hc;               // rows per chunk of the fused kernel (0 = auto); env SUHMO_FUSED_HC
    int fas_rhs_fused;          // coarse FAS right-hand side rhs = res + L(R phi), the copy of R phi and L(phi) in ONE pass (k_apply<., 2>) instead of
                                // copy + applyOp + axby (env SUHMO_FAS_RHS_FUSED, default 1)
    int strips_rhs_local;       // rank strips: R phi and RES travel together, the coarse right-hand side of the halo rows is computed locally
                                // (env SUHMO_STRIPS_RHS_LOCAL, default 1)
    int skip_mask;              // the streaming relaxation skips the ice-mask array in a V-cycle whose UpdateOperator found no negative cell
                                // (env SUHMO_SKIP_MASK, default 1)
    int mask_reported;          // the last suhmo_level_update_operator(depth 0) made that report
    int coarse_mask_ok;         // the ice masks of the depths > 0 are MGnewOp's averages of depth 0's current one (suhmo_build_mg_coefficients; any
                                // write to the field clears it): the report about depth 0 then covers them (an average of non-negative cells)
    unsigned mask_epoch, maskflag_epoch;   // number of the last k_bcoef_fused call on depth 0; the call whose report is current (0: none)
    int overlap_halo;           // rank strips, streaming kernel: the halo exchange travels on a second stream while the chunks that do not
                                // read halo rows relax; the two end chunks follow (env SUHMO_OVERLAP_HALO; 1 = default: with the native, stream-ordered
                                // RCCL transport only; 2 = with any hook, the caller vouches that it orders against the stream it is given; 0 = off)
    hipStream_t xstream; hipEvent_t xev[2]; long overlapped;   // ... its stream and events; launches that overlapped so far
    int tile_strips;            // tile kernel on rank strips (env SUHMO_TILE_STRIPS, default 1)
    int tile_chunks;            // a level that is one tile relaxes all its sweeps in one launch (env SUHMO_TILE_CHUNKS, default 1)
    int resid_in_relax;         // the solve loops' residual evaluation rides on the launch that ends the V-cycle (streaming kernel at depth 0; default 1)
    int fas_rhs_in_relax;       // coarse FAS right-hand side formed by the first relax of the depth: bit 0 in the tile kernel, bit 1 in the streaming kernel (env SUHMO_FAS_RHS_IN_RELAX, default 3)
    long tile_max_cells;        // auto mode: levels below this many cells relax on the tile kernel (env SUHMO_TILE_MAX_CELLS)
    int tile_restrict;          // the tile kernel's last pre-smoothing launch also restricts: 0 never (a separate kernel restricts: faster on one GPU and
                                // on a rank strip alike, profiles/r02_h_tile_restrict_ab.txt), 1 always, 2 on rank strips only (env SUHMO_TILE_RESTRICT, default 0)
    int tile_order;             // workgroup -> tile map of the tile kernel: 0 as launched, 1 a contiguous run of tiles per XCD, 2 the same in panels of 8 tile rows (env SUHMO_TILE_ORDER)
    int gsrb_tile, tile_t, tile_s;      // cache-resident depths: S sweeps per launch on LDS tiles (env SUHMO_GSRB_TILE, default 1); tile edge 16 / 32
                                // (env SUHMO_TILE_T, 0 = by size)
};

void suhmo_set_error(const char *fmt, ...);
// Named scoped timers with the reference's CH_TIME labels (src/VCAMRNonLinearPoissonOp.cpp:40,69,103,277,390,660; report =
// CH_TIMER_REPORT, exec/A_SHMIP/Suhmo.cpp:136).  Off by default (one relaxed load per scope); suhmo_timers_enable(1): host wall
// time per scope, (2): the device is synchronised at both ends, so the time includes the kernels the scope launched.
struct SuhmoTimer { const char *name; double t0; int mode; explicit SuhmoTimer(const char *n); ~SuhmoTimer(); };
#define SUHMO_TIME(label) SuhmoTimer suhmo_tm_(label)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    suhmo_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return -2; } } while (0)
#define ARG(cond) do { if (!(cond)) { suhmo_set_error("%s:%d bad argument: %s", __FILE__, __LINE__, #cond); return -1; } } while (0)

double *suhmo_field(suhmo_level *L, int depth, int field);   // lazily allocates
int suhmo_level_create_(suhmo_level_t **out, const suhmo_level_desc_t *desc, bool stub);   // suhmo_level.hip
int suhmo_level_materialize_(suhmo_level *L);
// Two canvases of a level trade places while its FAS problem runs (the level's own right-hand side is set aside): whatever way the
// scope is left -- an exchange or all-gather hook failing in between included -- they trade back, so a caller that catches the
// error still holds the problem it posed
// every change of a field POINTER of any level (a field allocated on first use, phi canvases trading places) advances this count: tables of
// pointers kept elsewhere (suhmo_hier.hip: the boxes of a level as one launch target) are compared again only after it moved
void suhmo_fp_changed();
unsigned long suhmo_fp_epoch();
struct SwapGuard {
    double **a, **b; bool armed;
    SwapGuard() : a(nullptr), b(nullptr), armed(false) {}
    void arm(double **a_, double **b_) { a = a_; b = b_; std::swap(*a, *b); armed = true; suhmo_fp_changed(); }
    void back() { if (armed) { std::swap(*a, *b); armed = false; suhmo_fp_changed(); } }
    ~SwapGuard() { back(); }
};
int suhmo_average_operator_all(suhmo_level *L, int nd, hipStream_t st);      // suhmo_bcoef.hip
int suhmo_restrict_both(suhmo_level *L, int depth, hipStream_t st);                 // suhmo_ops.hip
bool suhmo_gsrb_can_fuse_prolong(suhmo_level *L, int depth, int sweeps);           // suhmo_gsrb.hip
int suhmo_level_residual_and_norm(suhmo_level *L, double *out, hipStream_t st);                   // suhmo_ops.hip
int suhmo_level_norm_from_partials(suhmo_level *L, int np, double *out, hipStream_t st);   // suhmo_ops.hip
bool suhmo_gsrb_can_fuse_rhs(suhmo_level *L, int depth, int sweeps, bool rhs_local = false);               // suhmo_gsrb.hip
int suhmo_launch_gsrb(suhmo_level *L, int depth, int sweeps, int tail, hipStream_t st, int *restricted = nullptr);   // suhmo_gsrb.hip; tail = halo
                                                    // rows worth keeping valid at exit; restricted: see there
void suhmo_level_drop_graphs(suhmo_level *L);                                     // suhmo_fas.hip
int suhmo_ensure_phi_halo(suhmo_level *L, int depth, int need, hipStream_t st);      // suhmo_ops.hip
int suhmo_fas_coarse_rhs(suhmo_level *L, int depth, hipStream_t st, int hcomp = 0);  // suhmo_ops.hip; hcomp: halo rows that get rhs computed too
// Result of a reduction whose last kernel was launched with suhmo_host_slot(L): 8 bytes back on the host.
struct HostSlot { double *val; unsigned long long *flag; unsigned long long seq; };
HostSlot suhmo_host_slot(suhmo_level *L);                                           // suhmo_ops.hip; call right before the launch
int suhmo_readback(suhmo_level *L, hipStream_t st, double *out, double *out2 = nullptr);   // after the launch; out2: a second value (scratch[1] / val[1])
// n = 1 or 2 values a reduction's last kernel left in L->scratch[0..n-1] (launched with suhmo_reduce_slot(L)): combined over the ranks of a
// strip partition (op 0 MAX, 1 SUM) and brought to the host.  On a strip with the native transport the all-reduce runs on the device, on
// the kernels' stream, and a publishing kernel follows; with host hooks the local values are read back first.
HostSlot suhmo_reduce_slot(suhmo_level *L);                                         // = suhmo_host_slot, or an empty slot when the values still travel
int suhmo_reduce_finish(suhmo_level *L, hipStream_t st, int n, int op, double *out, double *out2 = nullptr);
__device__ __forceinline__ void suhmo_publish(const HostSlot &h, double v)
{
    if (!h.val) return;
    h.val[0] = v;
    __hip_atomic_store(h.flag, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int suhmo_build_mg_coefficients(suhmo_level *L, bool with_faces, hipStream_t st);   // suhmo_bcoef.hip
// suhmo_agg.hip: agglomeration of the coarse multigrid depths of a rank strip
void suhmo_ipc_release(suhmo_level *L);   // suhmo_ipc.hip
int suhmo_ipc_batch(suhmo_level *L, int open, hipStream_t st);
int suhmo_agg_setup(suhmo_level *L);
void suhmo_agg_release(suhmo_level *L);
int suhmo_agg_gather_static(suhmo_level *L, bool with_faces, hipStream_t st);
int suhmo_agg_gather_faces(suhmo_level *L, int nd, hipStream_t st);
int suhmo_agg_gather_state(suhmo_level *L, hipStream_t st);
int suhmo_agg_scatter(suhmo_level *L, hipStream_t st);
int suhmo_exchange_list(suhmo_level *L, int depth, const int *fields, int n, hipStream_t st);   // LevelData::exchange across rank boundaries
static inline int suhmo_halo_rows(const DV &v) { return v.gy < v.ny ? v.gy : v.ny; }
