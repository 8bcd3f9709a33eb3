// suhmo_level.hip -- the level handle: named timers, lifecycle (canvases, views, stubs), options, LevelData traffic (fields and boxes in and
// out), the strip-halo row packers of host transports, profiling.  The operator kernels live in suhmo_ops.hip, UpdateOperator / AverageOperator /
// the MG coefficients in suhmo_bcoef.hip, the relaxation in suhmo_gsrb.hip, the FAS driver in suhmo_fas.hip.
// gfx950 only.  Reference citations: file:line in the SUHMO checkout.
#include "suhmo_hier.h"
#include "suhmo_level_int.h"
#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <initializer_list>

static thread_local char g_err[512] = "";
void suhmo_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char *suhmo_last_error(void) { return g_err; }
// ------------------------------------------------------------------ named timers (CH_TIME / CH_TIMER_REPORT)
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
namespace {
std::atomic<int> g_timer_mode{-1};
std::mutex g_timer_mu;
struct TimerRec { long count = 0; double total = 0.0; };
std::map<std::string, TimerRec> g_timers;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int timer_mode()
{
    int m = g_timer_mode.load(std::memory_order_relaxed);
    if (m < 0) { const char *e = getenv("SUHMO_TIMERS"); m = e ? atoi(e) : 0; g_timer_mode.store(m); }
    return m;
}
}  // namespace
SuhmoTimer::SuhmoTimer(const char *n) : name(n), t0(0.0), mode(timer_mode())
{
    if (!mode) return;
    if (mode >= 2) (void)hipDeviceSynchronize();
    t0 = now_s();
}
SuhmoTimer::~SuhmoTimer()
{
    if (!mode) return;
    if (mode >= 2) (void)hipDeviceSynchronize();
    double dt = now_s() - t0;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    TimerRec &r = g_timers[name];
    r.count++; r.total += dt;
}
extern "C" int suhmo_timers_enable(int mode) { g_timer_mode.store(mode < 0 ? 0 : mode); return 0; }
extern "C" int suhmo_timers_reset(void) { std::lock_guard<std::mutex> lk(g_timer_mu); g_timers.clear(); return 0; }
// "label  calls  total [s]  mean [us]" per line, most expensive first; returns the number of bytes the full report needs
extern "C" long suhmo_timers_report(char *buf, long size)
{
    std::vector<std::pair<std::string, TimerRec>> v;
    { std::lock_guard<std::mutex> lk(g_timer_mu); v.assign(g_timers.begin(), g_timers.end()); }
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.second.total > b.second.total; });
    std::string out;
    char line[256];
    for (auto &e : v) {
        snprintf(line, sizeof(line), "%-56s %10ld %14.6f %12.2f\n", e.first.c_str(), e.second.count, e.second.total, 1e6 * e.second.total / std::max(1L, e.second.count));
        out += line;
    }
    if (buf && size > 0) { long n = std::min<long>(size - 1, (long)out.size()); memcpy(buf, out.data(), n); buf[n] = 0; }
    return (long)out.size() + 1;
}

extern "C" int suhmo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------ level lifecycle
static bool boxes_coarsenable(const std::vector<int> &b, int r)
{
    for (size_t k = 0; k < b.size(); k += 4)
        if (b[k] % r || b[k + 1] % r || (b[k + 2] + 1) % r || (b[k + 3] + 1) % r) return false;
    return true;
}

static void make_dv(DV &v, const suhmo_level_desc_t &d, int depth)
{
    int c = 1 << depth;
    memset(&v, 0, sizeof(v));
    v.nx = d.nx / c; v.ny = d.ny / c;
    v.gy = d.halo_rows < 1 ? 1 : d.halo_rows;
    v.rows = v.ny + 2 * v.gy;
    v.P = ((v.nx + 2 * SUHMO_XOFF + 15) / 16) * 16;
    v.j0 = d.j0 / c; v.nyg = d.ny_global / c;
    v.i0 = d.i0 / c; v.nxg = (d.nx_global > 0 ? d.nx_global : d.nx) / c;
    v.dx = d.dx * c; v.dy = d.dy * c;
    v.rdx = 1.0 / (v.dx * v.dx); v.rdy = 1.0 / (v.dy * v.dy);
    v.fdx = 1.0 / v.dx; v.fdy = 1.0 / v.dy;
    v.alpha = d.alpha; v.beta = d.beta;
    for (int dir = 0; dir < 2; dir++) {
        v.per[dir] = d.bc.periodic[dir];
        for (int s = 0; s < 2; s++) {
            v.bct[dir][s] = d.bc.type[dir][s];
            v.two_v[dir][s] = 2.0 * d.bc.value[dir][s];
            double isign = s == 0 ? -1.0 : 1.0;
            v.neu[dir][s] = isign * (dir == 0 ? v.dx : v.dy) * d.bc.value[dir][s];
        }
    }
    bool whole = (d.j0 == 0 && d.ny == d.ny_global);
    v.ext[0] = (!whole) && (d.j0 > 0 || d.bc.periodic[1]);
    v.ext[1] = (!whole) && (d.j0 + d.ny < d.ny_global || d.bc.periodic[1]);
    v.cfx[0] = v.i0 > 0 || (v.per[0] && v.nx < v.nxg);     // a box that does not span a periodic domain keeps stored ghost columns there
    v.cfx[1] = v.i0 + v.nx < v.nxg || (v.per[0] && v.nx < v.nxg);
    if (d.nx_global > 0) {               // AMR patch: rank boundaries only inside the patch's own row range
        const int pj0 = d.patch_ny > 0 ? d.patch_j0 : d.j0, pj1 = d.patch_ny > 0 ? d.patch_j0 + d.patch_ny : d.j0 + d.ny;
        v.rk[0] = d.j0 > pj0; v.rk[1] = d.j0 + d.ny < pj1;
    } else { v.rk[0] = v.ext[0]; v.rk[1] = v.ext[1]; }
}

static std::atomic<unsigned long> g_fp_epoch{1};
void suhmo_fp_changed() { g_fp_epoch.fetch_add(1, std::memory_order_relaxed); }
unsigned long suhmo_fp_epoch() { return g_fp_epoch.load(std::memory_order_relaxed); }

double *suhmo_field(suhmo_level *L, int depth, int field)
{
    Depth &D = L->d[depth];
    if (!D.fp.f[field]) {
        double *p = nullptr;
        if (L->stub) return nullptr;                              // geometry only (a box another rank holds)
        if (hipMalloc(&p, D.elems * sizeof(double)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, D.elems * sizeof(double));
        D.fp.f[field] = p;
        suhmo_fp_changed();
    }
    return D.fp.f[field];
}

// stub: geometry, options and boxes only -- no canvas, no scratch: a box of a partitioned AMR level this rank neither owns nor reads
// (suhmo_hier.hip); suhmo_level_materialize_ turns it into a level with storage
static int level_storage(suhmo_level *L);
int suhmo_level_create_(suhmo_level_t **out, const suhmo_level_desc_t *desc, bool stub);
extern "C" int suhmo_level_create(suhmo_level_t **out, const suhmo_level_desc_t *desc) { return suhmo_level_create_(out, desc, false); }
int suhmo_level_materialize_(suhmo_level *L) { return L->stub ? level_storage(L) : 0; }
int suhmo_level_create_(suhmo_level_t **out, const suhmo_level_desc_t *desc, bool stub)
{
    ARG(out && desc);
    ARG(desc->nx >= 2 && desc->ny >= 2 && desc->dx > 0 && desc->dy > 0);
    ARG(desc->ny_global >= desc->ny && desc->j0 >= 0 && desc->j0 + desc->ny <= desc->ny_global);
    ARG(desc->i0 >= 0 && (desc->nx_global == 0 ? desc->i0 == 0 : desc->i0 + desc->nx <= desc->nx_global));
    ARG(desc->i0 % 2 == 0);                  // colour parity is taken from the local column
    ARG(desc->patch_ny == 0 || (desc->nx_global > 0 && desc->patch_j0 <= desc->j0 && desc->j0 + desc->ny <= desc->patch_j0 + desc->patch_ny));
    ARG((long)(desc->nx + 64) * (long)(desc->ny + 2 * (desc->halo_rows < 1 ? 1 : desc->halo_rows)) < (1L << 31));
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { suhmo_set_error("no HIP device: the product path has no CPU fallback"); return -3; }
    ARG(desc->device >= 0 && desc->device < ndev);
    HIPCHK(hipSetDevice(desc->device));
    suhmo_level *L = new suhmo_level();
    L->desc = *desc; L->ph = desc->phys; L->device = desc->device;
    L->ex = nullptr; L->ar = nullptr; L->ar2 = nullptr; L->ard = nullptr; L->user = nullptr; L->ex_begin = nullptr; L->ex_end = nullptr; L->rccl = nullptr;
        L->ipc = nullptr; L->ipc_owner = 0; L->faces_deferred = 0; L->gap = nullptr; L->gap_dt = 0.0; L->prof_on = 0; L->gsrb_variant = -1; L->fused_hc = 0;
    // defaults of the kernel-selection knobs (each with the measurement it comes from where it is declared, suhmo_common.h) ...
    L->bcoef_fused = 1;
    L->bcoef_tile_x = 62;
    L->fused_nt = 64;                        // one wave per workgroup: 316 vs 308 V-cycles/s at 4096^2 (profiles/r01_i_nt_ab.txt)
    L->fused_restrict = 1;
    L->gsrb_tile = 1; L->tile_t = 0; L->tile_s = 4;
    L->tile_max_cells = 3000000;     // 2048^2 (4.2 M cells) streams: 45 us per sweep at its one-round chunk height against 51 on tiles (profiles/r03_stream2048.txt)
    L->fas_rhs_in_relax = 3;
    L->tile_chunks = 1;
    L->tile_order = 2; L->tile_restrict = 0;
    L->tile_strips = 1;
    L->overlap_halo = 1; L->xstream = nullptr; L->xev[0] = L->xev[1] = nullptr; L->overlapped = 0;
    L->strips_rhs_local = 1;
    L->fas_rhs_fused = 1;
    L->agg_min_cells = 100000; L->agg_depth = 0; L->agg_world = 1; L->agg_rank = 0; L->agg = nullptr; L->ag = nullptr; L->ag_user = nullptr;
    L->agg_send = L->agg_recv = nullptr; L->agg_cap = 0; L->agg_gathers = 0; L->agg_static_stale = 0;
    L->frhs_stream = L->frhs_tile = 0;
    L->resout_np = 0;
    L->resout_req = L->resout_armed = L->resout_done = 0; L->resout_rhs = nullptr; L->resout_count = 0; L->resid_in_relax = 1;
    L->graph_max_cells = 1500000; L->gstream = nullptr; memset(L->vgraph_seen, 0, sizeof(L->vgraph_seen));
    L->fused_min_cells = 1000000;
    L->skip_mask = 1; L->poll_readback = 1;
    // ... and ONE place where the environment may override them, read when a level is created: the A/B runs DESIGN.md quotes and the tests that
    // force a kernel onto a small level.  Nothing else of a level is taken from the environment.
    {
        struct IntKnob { const char *env; int suhmo_level::*field; };
        struct LongKnob { const char *env; long suhmo_level::*field; };
        static const IntKnob ints[] = {
            {"SUHMO_GSRB_VARIANT", &suhmo_level::gsrb_variant}, {"SUHMO_FUSED_HC", &suhmo_level::fused_hc}, {"SUHMO_BCOEF_FUSED", &suhmo_level::bcoef_fused},
            {"SUHMO_BCOEF_TILE_X", &suhmo_level::bcoef_tile_x}, {"SUHMO_FUSED_NT", &suhmo_level::fused_nt}, {"SUHMO_OVERLAP_HALO", &suhmo_level::overlap_halo},
            {"SUHMO_STRIPS_RHS_LOCAL", &suhmo_level::strips_rhs_local}, {"SUHMO_FAS_RHS_FUSED", &suhmo_level::fas_rhs_fused},
            {"SUHMO_RESID_IN_RELAX", &suhmo_level::resid_in_relax}, {"SUHMO_TILE_STRIPS", &suhmo_level::tile_strips}, {"SUHMO_TILE_CHUNKS", &suhmo_level::tile_chunks},
            {"SUHMO_TILE_RESTRICT", &suhmo_level::tile_restrict}, {"SUHMO_TILE_ORDER", &suhmo_level::tile_order}, {"SUHMO_FAS_RHS_IN_RELAX", &suhmo_level::fas_rhs_in_relax},
            {"SUHMO_TILE_S", &suhmo_level::tile_s}, {"SUHMO_GSRB_TILE", &suhmo_level::gsrb_tile}, {"SUHMO_TILE_T", &suhmo_level::tile_t},
            {"SUHMO_FUSED_RESTRICT", &suhmo_level::fused_restrict}, {"SUHMO_SKIP_MASK", &suhmo_level::skip_mask}, {"SUHMO_POLL_READBACK", &suhmo_level::poll_readback}};
        static const LongKnob longs[] = {
            {"SUHMO_AGG_MIN_CELLS", &suhmo_level::agg_min_cells}, {"SUHMO_TILE_MAX_CELLS", &suhmo_level::tile_max_cells},
            {"SUHMO_GRAPH_MAX_CELLS", &suhmo_level::graph_max_cells}, {"SUHMO_FUSED_MIN_CELLS", &suhmo_level::fused_min_cells}};
        for (const IntKnob &k : ints) if (const char *e = getenv(k.env)) L->*(k.field) = atoi(e);
        for (const LongKnob &k : longs) if (const char *e = getenv(k.env)) L->*(k.field) = atol(e);
        L->fas_rhs_fused = L->fas_rhs_fused != 0;
        if (L->tile_order < 0 || L->tile_order > 2) L->tile_order = 0;
        if (L->tile_t != 16 && L->tile_t != 32) L->tile_t = 0;          // (most sweeps per tile launch tile_s: 4, 2, 1)
    }
    if (desc->boxes && desc->nbox > 0) {
        L->boxes.assign(desc->boxes, desc->boxes + 4 * (size_t)desc->nbox);
    } else {
        int mb = desc->max_box > 0 ? desc->max_box : 64;
        for (int bj = 0; bj * mb < desc->ny; bj++)
            for (int bi = 0; bi * mb < desc->nx; bi++) {
                int lo0 = desc->i0 + bi * mb, lo1 = desc->j0 + bj * mb;
                int hi0 = std::min(lo0 + mb, desc->i0 + desc->nx) - 1, hi1 = std::min(lo1 + mb, desc->j0 + desc->ny) - 1;
                int b[4] = {lo0, lo1, hi0, hi1};
                L->boxes.insert(L->boxes.end(), b, b + 4);
            }
    }
    L->desc.boxes = nullptr;
    L->desc.nbox = (int)(L->boxes.size() / 4);
    // boxes must tile the strip exactly
    {
        long cells = 0;
        for (size_t k = 0; k < L->boxes.size(); k += 4) {
            const int *b = &L->boxes[k];
            if (b[0] < desc->i0 || b[2] >= desc->i0 + desc->nx || b[1] < desc->j0 || b[3] >= desc->j0 + desc->ny || b[0] > b[2] || b[1] > b[3]) {
                suhmo_set_error("box %zu outside the strip", k / 4); delete L; return -1;
            }
            cells += (long)(b[2] - b[0] + 1) * (b[3] - b[1] + 1);
        }
        if (cells != (long)desc->nx * desc->ny) { suhmo_set_error("boxes do not tile the level"); delete L; return -1; }
    }
    // MGnewOp depth rule (src/VCAMRNonLinearPoissonOp.cpp:1044-1060; s_maxCoarse = 2)
    L->ndepth = 1;
    for (int dep = 1; dep < SUHMO_MAXDEPTH; dep++) {
        if (!boxes_coarsenable(L->boxes, (1 << dep) * 2)) break;
        if ((desc->j0 % (1 << dep)) || (desc->ny_global % (1 << dep))) break;
        if ((desc->i0 % (2 << dep)) || (desc->nx_global % (1 << dep))) break;
        L->ndepth = dep + 1;
    }
    for (int dep = 0; dep < L->ndepth; dep++) {
        Depth &D = L->d[dep];
        make_dv(D.v, L->desc, dep);
        D.elems = (size_t)D.v.P * (size_t)(D.v.rows + 1);
        D.nbox = L->desc.nbox;
        memset(&D.fp, 0, sizeof(D.fp));
        D.phi_alt = nullptr; D.prolong_pending = 0; D.rhs_pending = 0; D.phi_fresh = 0;
    }
    L->scratch = nullptr; L->scratch_elems = 0; L->hscratch = nullptr; L->hscratch_dev = nullptr; L->hseq = 0;
    L->mask_epoch = 0; L->maskflag_epoch = 0; L->mask_reported = 0; L->coarse_mask_ok = 0;
    L->stub = 1;
    if (!stub) { int rc = level_storage(L); if (rc) { suhmo_level_destroy(L); return rc; } }
    *out = L;
    return 0;
}
static int level_storage(suhmo_level *L)
{
    static const int eager[] = {SUHMO_F_PHI, SUHMO_F_RHS, SUHMO_F_ACOEF, SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB,
                                SUHMO_F_MASK, SUHMO_F_BX, SUHMO_F_BY, SUHMO_F_RES, SUHMO_F_LPHI};
    HIPCHK(hipSetDevice(L->device));
    L->stub = 0;
    for (int dep = 0; dep < L->ndepth; dep++) {
        for (int f : eager) {
            if (dep == 0 && f == SUHMO_F_LPHI) continue;       // lazily (only tests / AMR use it at depth 0)
            if (!suhmo_field(L, dep, f)) { suhmo_set_error("hipMalloc failed (depth %d field %d)", dep, f); return -2; }
        }
        if (dep > 0) { suhmo_field(L, dep, SUHMO_F_PHIOLD); suhmo_field(L, dep, SUHMO_F_CORR); }
    }
    L->scratch_elems = 16384;
    HIPCHK(hipMalloc(&L->scratch, L->scratch_elems * sizeof(double)));
    HIPCHK(hipMemset(L->scratch, 0, L->scratch_elems * sizeof(double)));                   // (its last word: the negative-mask report of k_bcoef_fused)
    HIPCHK(hipHostMalloc(&L->hscratch, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    memset(L->hscratch, 0, 64 * sizeof(double));
    if (L->poll_readback && hipHostGetDevicePointer((void **)&L->hscratch_dev, L->hscratch, 0) != hipSuccess) { L->hscratch_dev = nullptr; L->poll_readback = 0; (void)hipGetLastError(); }
    return 0;
}

extern "C" int suhmo_level_destroy(suhmo_level_t *L)
{
    if (!L) return 0;
    (void)hipSetDevice(L->device);
    (void)hipDeviceSynchronize();
    if (L->rccl) (void)suhmo_level_detach_rccl(L);
    if (L->gap) { (void)suhmo_level_destroy(L->gap); L->gap = nullptr; }
    suhmo_ipc_release(L);
    suhmo_agg_release(L);
    suhmo_level_drop_graphs(L);
    for (int dep = 0; dep < L->ndepth; dep++)
        for (int f = 0; f < SUHMO_F_COUNT; f++)
            if (L->d[dep].fp.f[f]) (void)hipFree(L->d[dep].fp.f[f]);
    for (int dep = 0; dep < L->ndepth; dep++)
        if (L->d[dep].phi_alt) (void)hipFree(L->d[dep].phi_alt);
    for (auto &pe : L->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    if (L->xstream) { (void)hipStreamDestroy(L->xstream); (void)hipEventDestroy(L->xev[0]); (void)hipEventDestroy(L->xev[1]); }
    if (L->scratch) (void)hipFree(L->scratch);
    if (L->hscratch) (void)hipHostFree(L->hscratch);
    delete L;
    return 0;
}

// setAlphaAndBeta / setBC of the operator (src/VCAMRNonLinearPoissonOp.cpp:462-469, src/AMRNonLinearPoissonOp.cpp:1275-1278):
// every multigrid depth of the level takes the new values (the reference keeps them per operator object; its factory
// hands the same alpha, beta, BCHolder to all of them)
static int remake_views(suhmo_level *L)
{
    for (int dep = 0; dep < L->ndepth; dep++) make_dv(L->d[dep].v, L->desc, dep);
    suhmo_level_drop_graphs(L);                      // captured launches carry the old view by value
    return 0;
}
extern "C" int suhmo_level_set_alpha_beta(suhmo_level_t *L, double alpha, double beta)
{
    ARG(L);
    L->desc.alpha = alpha; L->desc.beta = beta;
    if (L->agg) { int rc = suhmo_level_set_alpha_beta(L->agg, alpha, beta); if (rc) return rc; }   // the agglomerated depths are depths of this operator
    return remake_views(L);
}
extern "C" int suhmo_level_set_bc(suhmo_level_t *L, const suhmo_bc_t *bc)
{
    ARG(L && bc);
    for (int d = 0; d < 2; d++)
        if ((bc->periodic[d] != 0) != (L->desc.bc.periodic[d] != 0)) { suhmo_set_error("setBC cannot change the periodicity of the domain"); return -1; }
    for (int d = 0; d < 2; d++) for (int s = 0; s < 2; s++) ARG(bc->type[d][s] == 0 || bc->type[d][s] == 1);
    L->desc.bc = *bc;
    if (L->agg) { int rc = suhmo_level_set_bc(L->agg, bc); if (rc) return rc; }
    return remake_views(L);
}

// Kernel selection of a level as an API (the SUHMO_* environment variables read at creation remain as an override for A/B
// runs).  Keys = the variable names without the SUHMO_ prefix, lower case.  Captured V-cycle graphs are dropped: they carry
// the old choice.
static long *option_slot_long(suhmo_level *L, const char *key)
{
    if (!strcmp(key, "fused_min_cells")) return &L->fused_min_cells;
    if (!strcmp(key, "tile_max_cells")) return &L->tile_max_cells;
    if (!strcmp(key, "graph_max_cells")) return &L->graph_max_cells;
    if (!strcmp(key, "agg_min_cells")) return &L->agg_min_cells;
    return nullptr;
}
static int *option_slot_int(suhmo_level *L, const char *key)
{
    static const struct { const char *k; int suhmo_level::*m; } tab[] = {
        {"gsrb_variant", &suhmo_level::gsrb_variant}, {"fused_hc", &suhmo_level::fused_hc}, {"bcoef_fused", &suhmo_level::bcoef_fused}, {"bcoef_tile_x", &suhmo_level::bcoef_tile_x},
        {"fused_nt", &suhmo_level::fused_nt}, {"fused_restrict", &suhmo_level::fused_restrict}, {"strips_rhs_local", &suhmo_level::strips_rhs_local},
        {"tile_strips", &suhmo_level::tile_strips}, {"overlap_halo", &suhmo_level::overlap_halo}, {"skip_mask", &suhmo_level::skip_mask}, {"tile_chunks",
            &suhmo_level::tile_chunks}, {"tile_order", &suhmo_level::tile_order}, {"tile_restrict", &suhmo_level::tile_restrict}, {"fas_rhs_in_relax",
            &suhmo_level::fas_rhs_in_relax}, {"resid_in_relax", &suhmo_level::resid_in_relax}, {"fas_rhs_fused", &suhmo_level::fas_rhs_fused},
        {"tile_s", &suhmo_level::tile_s}, {"gsrb_tile", &suhmo_level::gsrb_tile}, {"tile_t", &suhmo_level::tile_t}, {"poll_readback", &suhmo_level::poll_readback}};
    for (const auto &e : tab) if (!strcmp(key, e.k)) return &(L->*(e.m));
    return nullptr;
}
extern "C" int suhmo_level_set_option(suhmo_level_t *L, const char *key, long value)
{
    ARG(L && key);
    if (long *p = option_slot_long(L, key)) {
        *p = value; suhmo_level_drop_graphs(L);
        if (!strcmp(key, "agg_min_cells")) return suhmo_agg_setup(L);     // (every rank of the partition must set the same value)
        return 0;
    }
    if (int *p = option_slot_int(L, key)) {
        if (!strcmp(key, "tile_t") && value != 0 && value != 16 && value != 32) { suhmo_set_error("tile_t: 0 (by size), 16 or 32"); return -1; }
        if (!strcmp(key, "fused_nt") && value != 64 && value != 256) { suhmo_set_error("fused_nt: 64 or 256"); return -1; }
        if (!strcmp(key, "poll_readback") && value && !L->hscratch_dev) { suhmo_set_error("poll_readback: no device address of the host slot"); return -1; }
        *p = (int)value; suhmo_level_drop_graphs(L); return 0;
    }
    suhmo_set_error("unknown option '%s'", key);
    return -1;
}
extern "C" int suhmo_level_get_option(const suhmo_level_t *L, const char *key, long *value)
{
    ARG(L && key && value);
    if (long *p = option_slot_long(const_cast<suhmo_level *>(L), key)) { *value = *p; return 0; }
    if (int *p = option_slot_int(const_cast<suhmo_level *>(L), key)) { *value = *p; return 0; }
    if (!strcmp(key, "overlapped_launches")) { *value = L->overlapped; return 0; }       // read-only counter (overlap_halo)
    if (!strcmp(key, "agg_gathers")) { *value = L->agg_gathers; return 0; }              // read-only counter (agglomeration)
    if (!strcmp(key, "rhs_in_streaming_launches")) { *value = L->frhs_stream; return 0; }   // read-only counters (fas_rhs_in_relax)
    if (!strcmp(key, "rhs_in_tile_launches")) { *value = L->frhs_tile; return 0; }
    if (!strcmp(key, "residual_in_relax_launches")) { *value = L->resout_count; return 0; }
    suhmo_set_error("unknown option '%s'", key);
    return -1;
}

extern "C" int suhmo_level_num_depths(const suhmo_level_t *L) { return L ? L->ndepth : -1; }
extern "C" int suhmo_level_synchronize(suhmo_level_t *L, suhmo_stream_t s)
{
    ARG(L);
    HIPCHK(hipStreamSynchronize((hipStream_t)s));
    return 0;
}
extern "C" int suhmo_level_set_hooks(suhmo_level_t *L, suhmo_exchange_fn ex, suhmo_allreduce_max_fn ar, void *user)
{
    ARG(L);
    L->ex = ex; L->ar = ar; L->ar2 = nullptr; L->ard = nullptr; L->user = user; L->ex_begin = nullptr; L->ex_end = nullptr;
    return 0;
}
extern "C" int suhmo_level_set_allgather(suhmo_level_t *L, suhmo_allgather_fn fn, void *user)
{
    ARG(L);
    HIPCHK(hipSetDevice(L->device));
    L->ag = fn; L->ag_user = user;
    return suhmo_agg_setup(L);
}
extern "C" int suhmo_level_agglomerated_depth(const suhmo_level_t *L) { return L ? L->agg_depth : -1; }
extern "C" int suhmo_level_set_reduce_hook(suhmo_level_t *L, suhmo_allreduce_fn fn)
{
    ARG(L);
    L->ar2 = fn;
    return 0;
}

extern "C" int suhmo_level_exchange(suhmo_level_t *L, int depth, int field, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(field >= 0 && field < SUHMO_F_COUNT);
    const DV &v = L->d[depth].v;
    if (L->ex && (v.ext[0] || v.ext[1])) {
        if (!suhmo_field(L, depth, field)) return -2;
        int rc = L->ex(L->user, L, depth, &field, 1, s);
        if (!rc && field == SUHMO_F_PHI) L->d[depth].phi_fresh = suhmo_halo_rows(v);
        return rc;
    }
    return 0;
}
extern "C" int suhmo_level_halo_info(const suhmo_level_t *L, int depth, int *halo_rows, int *ext_lo, int *ext_hi, int *nx, int *ny)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    const DV &v = L->d[depth].v;
    if (halo_rows) *halo_rows = v.gy;
    if (ext_lo) *ext_lo = v.ext[0];
    if (ext_hi) *ext_hi = v.ext[1];
    if (nx) *nx = v.nx;
    if (ny) *ny = v.ny;
    return 0;
}

// ------------------------------------------------------------------ LevelData traffic

extern "C" int suhmo_level_field_view(suhmo_level_t *L, int depth, int field, double **base, long *pitch, long *origin)
{
    CHECK_DF(L, depth, field);
    double *p = suhmo_field(L, depth, field);
    if (!p) { suhmo_set_error("field allocation failed"); return -2; }
    if (base) *base = p;
    if (pitch) *pitch = L->d[depth].v.P;
    if (origin) *origin = cidx(L->d[depth].v, 0, 0);
    return 0;
}

static int copy2d(suhmo_level *L, int depth, int field, double *host, long hpitch, int i0, int j0, int ni, int nj,
                  bool to_canvas, bool on_device, hipStream_t st)
{
    if (ni <= 0 || nj <= 0) return 0;
    double *p = suhmo_field(L, depth, field);
    if (!p) { suhmo_set_error("field allocation failed"); return -2; }
    const DV &v = L->d[depth].v;
    double *c = p + cidx(v, i0, j0);
    hipMemcpyKind kind = to_canvas ? (on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice)
                                   : (on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost);
    if (to_canvas)
        HIPCHK(hipMemcpy2DAsync(c, (size_t)v.P * 8, host, (size_t)hpitch * 8, (size_t)ni * 8, nj, kind, st));
    else
        HIPCHK(hipMemcpy2DAsync(host, (size_t)hpitch * 8, c, (size_t)v.P * 8, (size_t)ni * 8, nj, kind, st));
    return 0;
}

static int field_io(suhmo_level *L, int depth, int field, double *buf, int ghosted, int on_device, bool set, hipStream_t st)
{
    const DV &v = L->d[depth].v;
    int rc;
    if (is_xface(field)) rc = copy2d(L, depth, field, buf, v.nx + 1, 0, 0, v.nx + 1, v.ny, set, on_device, st);
    else if (is_yface(field)) rc = copy2d(L, depth, field, buf, v.nx, 0, 0, v.nx, v.ny + 1, set, on_device, st);
    else if (ghosted) rc = copy2d(L, depth, field, buf, v.nx + 2, -1, -1, v.nx + 2, v.ny + 2, set, on_device, st);
    else rc = copy2d(L, depth, field, buf, v.nx, 0, 0, v.nx, v.ny, set, on_device, st);
    if (rc) return rc;
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int suhmo_level_set_field(suhmo_level_t *L, int depth, int field, const double *src, int ghosted,
                                     int on_device, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(src);
    HIPCHK(hipSetDevice(L->device));
    if (field == SUHMO_F_PHI) phi_changed(L, depth);
    if (field == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }
    return field_io(L, depth, field, (double *)src, ghosted, on_device, true, (hipStream_t)s);
}
extern "C" int suhmo_level_get_field(suhmo_level_t *L, int depth, int field, double *dst, int ghosted,
                                     int on_device, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(dst);
    HIPCHK(hipSetDevice(L->device));
    return field_io(L, depth, field, dst, ghosted, on_device, false, (hipStream_t)s);
}

// valid region of box ibox at `depth`, in strip-local indices; faces: surroundingNodes
static void box_region(const suhmo_level *L, int depth, int field, int ibox, int r[4])
{
    const int *b = &L->boxes[4 * (size_t)ibox];
    int c = 1 << depth;
    r[0] = b[0] / c - L->d[depth].v.i0; r[1] = b[1] / c - L->d[depth].v.j0;
    r[2] = (b[2] + 1) / c - 1 - L->d[depth].v.i0; r[3] = (b[3] + 1) / c - 1 - L->d[depth].v.j0;
    if (is_xface(field)) r[2] += 1;
    if (is_yface(field)) r[3] += 1;
}

extern "C" int suhmo_level_put_box(suhmo_level_t *L, int depth, int field, int ibox, const double *fab,
                                   int flo0, int flo1, int fhi0, int fhi1, int with_domain_ghosts, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(fab); ARG(ibox >= 0 && ibox < L->desc.nbox);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    const DV &v = L->d[depth].v;
    if (field == SUHMO_F_PHI) phi_changed(L, depth);
    if (field == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }
    int r[4]; box_region(L, depth, field, ibox, r);
    int j0 = v.j0;                         // fab indices are global: local j = global j - j0
    flo0 -= v.i0; fhi0 -= v.i0;            // ... and local i = global i - i0 (AMR patch)
    const bool patch = L->desc.nx_global > 0;   // patch sides inside the domain: the fab's ghosts are the coarse-fine data
    long fp = fhi0 - flo0 + 1;
    ARG(flo0 <= r[0] && fhi0 >= r[2] && flo1 - j0 <= r[1] && fhi1 - j0 >= r[3]);
    double *h = (double *)fab;
    int rc = copy2d(L, depth, field, h + (long)(r[1] + j0 - flo1) * fp + (r[0] - flo0), fp, r[0], r[1],
                    r[2] - r[0] + 1, r[3] - r[1] + 1, true, false, st);
    if (rc) return rc;
    if (with_domain_ghosts && !is_face(field)) {
        // ghost strips of the fab that fall outside the problem domain (1 layer)
        if (r[0] == 0 && flo0 <= -1)
            rc |= copy2d(L, depth, field, h + (long)(r[1] + j0 - flo1) * fp + (-1 - flo0), fp, -1, r[1], 1, r[3] - r[1] + 1, true, false, st);
        if (r[2] == v.nx - 1 && fhi0 >= v.nx)
            rc |= copy2d(L, depth, field, h + (long)(r[1] + j0 - flo1) * fp + (v.nx - flo0), fp, v.nx, r[1], 1, r[3] - r[1] + 1, true, false, st);
        if (r[1] + j0 == 0 && flo1 <= -1)
            rc |= copy2d(L, depth, field, h + (long)(-1 - flo1) * fp + (r[0] - flo0), fp, r[0], -1 - j0, r[2] - r[0] + 1, 1, true, false, st);
        else if (patch && r[1] == 0 && flo1 <= j0 - 1)
            rc |= copy2d(L, depth, field, h + (long)(j0 - 1 - flo1) * fp + (r[0] - flo0), fp, r[0], -1, r[2] - r[0] + 1, 1, true, false, st);
        if (r[3] + j0 == v.nyg - 1 && fhi1 >= v.nyg)
            rc |= copy2d(L, depth, field, h + (long)(v.nyg - flo1) * fp + (r[0] - flo0), fp, r[0], v.nyg - j0, r[2] - r[0] + 1, 1, true, false, st);
        else if (patch && r[3] == v.ny - 1 && fhi1 >= j0 + v.ny)
            rc |= copy2d(L, depth, field, h + (long)(j0 + v.ny - flo1) * fp + (r[0] - flo0), fp, r[0], v.ny, r[2] - r[0] + 1, 1, true, false, st);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int suhmo_level_get_box(suhmo_level_t *L, int depth, int field, int ibox, double *fab,
                                   int flo0, int flo1, int fhi0, int fhi1, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(fab); ARG(ibox >= 0 && ibox < L->desc.nbox);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    const DV &v = L->d[depth].v;
    int j0 = v.j0;
    flo0 -= v.i0; fhi0 -= v.i0;
    long fp = fhi0 - flo0 + 1;
    // clip the fab box to the stored canvas region (1 ghost layer in x, gy rows in y)
    int lo0 = std::max(flo0, -1), hi0 = std::min(fhi0, v.nx + (field == SUHMO_F_BX ? 0 : 0));
    int lo1 = std::max(flo1 - j0, -v.gy), hi1 = std::min(fhi1 - j0, v.ny + v.gy - 1);
    int rc = copy2d(L, depth, field, fab + (long)(lo1 + j0 - flo1) * fp + (lo0 - flo0), fp, lo0, lo1,
                    hi0 - lo0 + 1, hi1 - lo1 + 1, false, false, st);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------ multi-GPU strip halos
// rows are shipped (nx + 1) wide so that x-face fields (nx + 1 faces per row) travel whole
__global__ void k_pack_rows(DV v, const double *__restrict__ p, int jstart, int rows, double *__restrict__ buf)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > v.nx || r >= rows) return;
    buf[(size_t)r * (v.nx + 1) + i] = p[cidx(v, i, jstart + r)];
}
__global__ void k_unpack_rows(DV v, double *__restrict__ p, int jstart, int rows, const double *__restrict__ buf)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > v.nx || r >= rows) return;
    p[cidx(v, i, jstart + r)] = buf[(size_t)r * (v.nx + 1) + i];
}
extern "C" int suhmo_level_pack_rows(suhmo_level_t *L, int depth, int field, int side, int rows, double *dev_buf, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(dev_buf); ARG(side == 0 || side == 1);
    const DV &v = L->d[depth].v;
    ARG(rows >= 1 && rows <= v.gy && rows <= v.ny);
    HIPCHK(hipSetDevice(L->device));
    int jstart = side == 0 ? 0 : v.ny - rows;     // owned rows next to that side, ascending j
    // y-faces: face row 0 of this strip IS face row ny of the lower neighbour (both own it), so
    // the rows the lower neighbour lacks start at face row 1
    if (field == SUHMO_F_BY && side == 0) jstart = 1;
    hipLaunchKernelGGL(k_pack_rows, grid2d(v.nx + 1, rows), BLK2D, 0, (hipStream_t)s, v, suhmo_field(L, depth, field), jstart, rows, dev_buf);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_unpack_rows(suhmo_level_t *L, int depth, int field, int side, int rows, const double *dev_buf, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(dev_buf); ARG(side == 0 || side == 1);
    const DV &v = L->d[depth].v;
    ARG(rows >= 1 && rows <= v.gy);
    HIPCHK(hipSetDevice(L->device));
    int jstart = side == 0 ? -rows : v.ny;        // ghost rows of that side, ascending j
    if (field == SUHMO_F_BY && side == 1) jstart = v.ny + 1;   // face row ny is owned (see pack)
    hipLaunchKernelGGL(k_unpack_rows, grid2d(v.nx + 1, rows), BLK2D, 0, (hipStream_t)s, v, suhmo_field(L, depth, field), jstart, rows, dev_buf);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ profiling helper
extern "C" int suhmo_level_profile_enable(suhmo_level_t *L, int on) { ARG(L); L->prof_on = on; return 0; }
extern "C" int suhmo_level_profile_reset(suhmo_level_t *L)
{
    ARG(L);
    for (auto &pe : L->prof) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    L->prof.clear();
    return 0;
}
static int profile_read_kind(suhmo_level_t *L, suhmo_stream_t s, int kind, double *ms_total, long *launches, long *cells);
extern "C" int suhmo_level_profile_read(suhmo_level_t *L, suhmo_stream_t s, double *ms_total, long *launches, long *cells)
{ return profile_read_kind(L, s, 0, ms_total, launches, cells); }
extern "C" int suhmo_level_profile_read_restricting(suhmo_level_t *L, suhmo_stream_t s, double *ms_total, long *launches, long *cells)
{ return profile_read_kind(L, s, 1, ms_total, launches, cells); }
static int profile_read_kind(suhmo_level_t *L, suhmo_stream_t s, int kind, double *ms_total, long *launches, long *cells)
{
    ARG(L);
    HIPCHK(hipStreamSynchronize((hipStream_t)s));
    double tot = 0.0; long n = 0, c = 0;
    for (auto &pe : L->prof) {
        if (pe.restricts != kind) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
        tot += ms; n++; c += pe.cells;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    if (cells) *cells = c;
    return 0;
}

