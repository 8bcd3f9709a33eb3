// suhmo_level.hip -- level canvas management, LevelData traffic and every operator
// kernel except the GSRB relaxation (suhmo_gsrb.hip) and the FAS driver (suhmo_fas.hip).
// gfx950 only.  Reference citations: file:line in the SUHMO checkout.
#include "suhmo_hier.h"
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
    L->ex = nullptr; L->ar = nullptr; L->ar2 = nullptr; L->ard = nullptr; L->user = nullptr; L->ex_begin = nullptr; L->ex_end = nullptr; L->rccl = nullptr; L->ipc = nullptr; L->ipc_owner = 0; L->faces_deferred = 0; L->gap = nullptr; L->gap_dt = 0.0; L->prof_on = 0; L->gsrb_variant = -1; L->fused_hc = 0;
    if (const char *e = getenv("SUHMO_GSRB_VARIANT")) L->gsrb_variant = atoi(e);
    if (const char *e = getenv("SUHMO_FUSED_HC")) L->fused_hc = atoi(e);
    L->bcoef_fused = 1;
    if (const char *e = getenv("SUHMO_BCOEF_FUSED")) L->bcoef_fused = atoi(e);
    L->bcoef_tile_x = 62;
    if (const char *e = getenv("SUHMO_BCOEF_TILE_X")) L->bcoef_tile_x = atoi(e);
    L->fused_nt = 64;                        // one wave per workgroup: 316 vs 308 V-cycles/s at 4096^2 (profiles/r01_i_nt_ab.txt)
    if (const char *e = getenv("SUHMO_FUSED_NT")) L->fused_nt = atoi(e);
    L->fused_restrict = 1;
    L->gsrb_tile = 1; L->tile_t = 0; L->tile_s = 4;
    L->tile_max_cells = 3000000;     // 2048^2 (4.2 M cells) streams: 45 us per sweep at its one-round chunk height against 51 on tiles (profiles/r03_stream2048.txt)
    L->fas_rhs_in_relax = 3;
    L->tile_chunks = 1;
    L->tile_order = 2; L->tile_restrict = 0;
    L->tile_strips = 1;
    L->overlap_halo = 1; L->xstream = nullptr; L->xev[0] = L->xev[1] = nullptr; L->overlapped = 0;
    if (const char *e = getenv("SUHMO_OVERLAP_HALO")) L->overlap_halo = atoi(e);
    L->strips_rhs_local = 1;
    if (const char *e = getenv("SUHMO_STRIPS_RHS_LOCAL")) L->strips_rhs_local = atoi(e);
    L->fas_rhs_fused = 1;
    if (const char *e = getenv("SUHMO_FAS_RHS_FUSED")) L->fas_rhs_fused = atoi(e) != 0;
    L->agg_min_cells = 100000; L->agg_depth = 0; L->agg_world = 1; L->agg_rank = 0; L->agg = nullptr; L->ag = nullptr; L->ag_user = nullptr;
    L->agg_send = L->agg_recv = nullptr; L->agg_cap = 0; L->agg_gathers = 0; L->agg_static_stale = 0;
    L->frhs_stream = L->frhs_tile = 0;
    L->resout_np = 0;
    L->resout_req = L->resout_armed = L->resout_done = 0; L->resout_rhs = nullptr; L->resout_count = 0; L->resid_in_relax = 1;
    if (const char *e = getenv("SUHMO_RESID_IN_RELAX")) L->resid_in_relax = atoi(e);
    if (const char *e = getenv("SUHMO_AGG_MIN_CELLS")) L->agg_min_cells = atol(e);
    if (const char *e = getenv("SUHMO_TILE_STRIPS")) L->tile_strips = atoi(e);
    if (const char *e = getenv("SUHMO_TILE_CHUNKS")) L->tile_chunks = atoi(e);
    if (const char *e = getenv("SUHMO_TILE_RESTRICT")) L->tile_restrict = atoi(e);
    if (const char *e = getenv("SUHMO_TILE_ORDER")) { L->tile_order = atoi(e); if (L->tile_order < 0 || L->tile_order > 2) L->tile_order = 0; }
    if (const char *e = getenv("SUHMO_FAS_RHS_IN_RELAX")) L->fas_rhs_in_relax = atoi(e);
    if (const char *e = getenv("SUHMO_TILE_MAX_CELLS")) L->tile_max_cells = atol(e);
    if (const char *e = getenv("SUHMO_TILE_S")) L->tile_s = atoi(e);        // most sweeps per tile launch (4, 2, 1)
    if (const char *e = getenv("SUHMO_GSRB_TILE")) L->gsrb_tile = atoi(e);
    if (const char *e = getenv("SUHMO_TILE_T")) { L->tile_t = atoi(e); if (L->tile_t != 16 && L->tile_t != 32) L->tile_t = 0; }
    if (const char *e = getenv("SUHMO_FUSED_RESTRICT")) L->fused_restrict = atoi(e);
    L->graph_max_cells = 1500000; L->gstream = nullptr; memset(L->vgraph_seen, 0, sizeof(L->vgraph_seen));
    if (const char *e = getenv("SUHMO_GRAPH_MAX_CELLS")) L->graph_max_cells = atol(e);
    L->fused_min_cells = 1000000;
    if (const char *e = getenv("SUHMO_FUSED_MIN_CELLS")) L->fused_min_cells = atol(e);
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
    L->scratch = nullptr; L->scratch_elems = 0; L->hscratch = nullptr; L->hscratch_dev = nullptr; L->hseq = 0; L->poll_readback = 1;
    L->mask_epoch = 0; L->maskflag_epoch = 0; L->mask_reported = 0; L->skip_mask = 1; L->coarse_mask_ok = 0;
    if (const char *e = getenv("SUHMO_SKIP_MASK")) L->skip_mask = atoi(e);
    if (const char *e = getenv("SUHMO_POLL_READBACK")) L->poll_readback = atoi(e);
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
        {"tile_strips", &suhmo_level::tile_strips}, {"overlap_halo", &suhmo_level::overlap_halo}, {"skip_mask", &suhmo_level::skip_mask}, {"tile_chunks", &suhmo_level::tile_chunks}, {"tile_order", &suhmo_level::tile_order}, {"tile_restrict", &suhmo_level::tile_restrict}, {"fas_rhs_in_relax", &suhmo_level::fas_rhs_in_relax}, {"resid_in_relax", &suhmo_level::resid_in_relax}, {"fas_rhs_fused", &suhmo_level::fas_rhs_fused},
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
static inline void phi_changed(suhmo_level *L, int depth) { L->d[depth].phi_fresh = 0; }   // see suhmo_ensure_phi_halo
static bool is_xface(int f) { return f == SUHMO_F_BX || f == SUHMO_F_QWX || f == SUHMO_F_DCX; }
static bool is_yface(int f) { return f == SUHMO_F_BY || f == SUHMO_F_QWY || f == SUHMO_F_DCY; }
static bool is_face(int f) { return is_xface(f) || is_yface(f); }
#define CHECK_DF(L, depth, field) ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(field >= 0 && field < SUHMO_F_COUNT); \
    if (L->stub) { suhmo_set_error("this box of a partitioned AMR level is held by another rank (suhmo_hier_box_owner)"); return -7; }

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

// ------------------------------------------------------------------ kernels
#define BLK2D dim3(64, 4)
static inline dim3 grid2d(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4); }

// exchange (periodic wrap) + mixBCValues into the stored ghost ring of a cell field
// (src/AmrHydro.cpp:248-309).  One thread per perimeter cell.
__device__ __forceinline__ void d_fill_ghosts(const DV &v, double *__restrict__ p, int homog)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * v.ny) {                       // x sides
        int side = t / v.ny, j = t % v.ny;
        if (v.cfx[side]) return;              // coarse-fine side: ghost columns hold interpolated data
        int i = side ? v.nx - 1 : 0;
        int idx = cidx(v, i, j);
        double c = p[idx];
        if (side) p[idx + 1] = phiE(v, p, idx, i, c, homog); else p[idx - 1] = phiW(v, p, idx, i, c, homog);
        return;
    }
    t -= 2 * v.ny;
    if (t < 2 * v.nx) {                       // y sides
        int side = t / v.nx, i = t % v.nx;
        if (v.ext[side]) return;              // rank boundary: ghost rows hold exchanged data
        int j = side ? v.ny - 1 : 0;
        int idx = cidx(v, i, j);
        double c = p[idx];
        if (side) p[idx + v.P] = phiN(v, p, idx, j, c, homog); else p[idx - v.P] = phiS(v, p, idx, j, c, homog);
    }
}
__global__ void k_fill_ghosts(DV v, double *__restrict__ p, int homog)
{
    d_fill_ghosts(v, p, homog);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_fill_ghosts_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field, int homog)
{
    d_fill_ghosts(vt[blockIdx.z], ft[blockIdx.z].f[field], homog);
}

extern "C" int suhmo_level_fill_ghosts(suhmo_level_t *L, int depth, int field, int homogeneous, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field); ARG(!is_face(field));
    HIPCHK(hipSetDevice(L->device));
    const DV &v = L->d[depth].v;
    double *p = suhmo_field(L, depth, field);
    int n = 2 * v.ny + 2 * v.nx;
    hipLaunchKernelGGL(k_fill_ghosts, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, v, p, homogeneous);
    HIPCHK(hipGetLastError());
    return 0;
}

// VCNLCOMPUTEOP2D / VCNLCOMPUTERES2D with BC, NL fused.  MODE 0: LPHI = L(phi); 1: RES = rhs - L(phi);
// 2: the FAS right-hand side of a coarse depth in one pass: LPHI = L(phi), RHS = axby(RES, LPHI, 1, 1), PHIOLD = phi
// 3: LPHI = L(phi) and RES = axby(LPHI, RHS, -1, 1) in one pass (the composite residual of an AMR level, suhmo_hier.hip)
// (returns what it stored in RES, MODE 1 / 3; 0 for a thread outside the level)
template <bool HAS_ALPHA, int MODE>
__device__ __forceinline__ double d_apply_at(const DV &v, const FP &fp, suhmo_phys_t ph, int homog, int halo, int hcomp, int i, int j)
{
    if (i >= v.nx || j >= v.ny + halo) return 0.0;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    if (MODE == 2 && (j < 0 || j >= v.ny)) {
        // rank strip: the first hcomp halo rows beyond a rank boundary get the neighbour's right-hand side computed here (its phi
        // and RES were exchanged together), which saves the exchange of RHS; the rows further out only copy phi
        const bool comp = (j < 0 && v.rk[0] && j >= -hcomp) || (j >= v.ny && v.rk[1] && j < v.ny + hcomp);
        if (!comp) { fp.f[SUHMO_F_PHIOLD][idx] = phi[idx]; return 0.0; }
    }
    double c = phi[idx];
    double e = phiE(v, phi, idx, i, c, homog), w = phiW(v, phi, idx, i, c, homog);
    double n = phiN(v, phi, idx, j, c, homog), s = phiS(v, phi, idx, j, c, homog);
    double bxW = fp.f[SUHMO_F_BX][idx], bxE = fp.f[SUHMO_F_BX][idx + 1];
    double byS = fp.f[SUHMO_F_BY][idx], byN = fp.f[SUHMO_F_BY][idx + v.P];
    double nl, dnl;
    nl_terms(ph, c, fp.f[SUHMO_F_B][idx], fp.f[SUHMO_F_PI][idx], fp.f[SUHMO_F_ZB][idx], fp.f[SUHMO_F_MASK][idx], nl, dnl);
    double aterm = HAS_ALPHA ? v.alpha * fp.f[SUHMO_F_ACOEF][idx] : v.alpha;
    double lofphi = lofphi_cell(v, aterm, c, e, w, n, s, bxE, bxW, byN, byS, nl);
    double res = 0.0;
    if (MODE == 0) fp.f[SUHMO_F_LPHI][idx] = lofphi;
    else if (MODE == 1) fp.f[SUHMO_F_RES][idx] = res = fp.f[SUHMO_F_RHS][idx] - lofphi;
    else if (MODE == 3) { fp.f[SUHMO_F_LPHI][idx] = lofphi; fp.f[SUHMO_F_RES][idx] = res = -1.0 * lofphi + 1.0 * fp.f[SUHMO_F_RHS][idx]; }
    else {
        fp.f[SUHMO_F_LPHI][idx] = lofphi;
        fp.f[SUHMO_F_RHS][idx] = 1.0 * fp.f[SUHMO_F_RES][idx] + 1.0 * lofphi;
        fp.f[SUHMO_F_PHIOLD][idx] = c;
    }
    return res;
}
template <bool HAS_ALPHA, int MODE>
__device__ __forceinline__ void d_apply(const DV &v, const FP &fp, suhmo_phys_t ph, int homog, int halo, int hcomp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - halo;     // MODE 2 on a rank strip: the halo rows only copy phi
    d_apply_at<HAS_ALPHA, MODE>(v, fp, ph, homog, halo, hcomp, i, j);
}
template <bool HAS_ALPHA, int MODE>
__global__ __launch_bounds__(256) void k_apply(DV v, FP fp, suhmo_phys_t ph, int homog, int halo = 0, int hcomp = 0)
{
    d_apply<HAS_ALPHA, MODE>(v, fp, ph, homog, halo, hcomp);
}
// RES = rhs - L(phi) and, in the same pass, the first stage of its max norm (one partial per workgroup, as k_norm_partial leaves them for
// k_norm_final): the solve loop's residual evaluation on levels whose cycle's last launch cannot leave it behind (the tile-kernel sizes)
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_residual_norm(DV v, FP fp, suhmo_phys_t ph, double *__restrict__ partial)
{
    __shared__ double sm[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    double r = fabs(d_apply_at<HAS_ALPHA, 1>(v, fp, ph, 0, 0, 0, i, j));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r = fmax(r, __shfl_xor(r, o));
    if (threadIdx.x == 0) sm[threadIdx.y] = r;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
// LPHI and RES = rhs - LPHI on a list of rectangles (x = first column, y = first row, z = columns, w = rows) of the level: the part of
// a composite residual that has changed since the whole level was evaluated (suhmo_hier.hip); overlapping rectangles write the same values
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_apply_rects(DV v, FP fp, suhmo_phys_t ph, const int4 *__restrict__ rects)
{
    const int4 r = rects[blockIdx.z];
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
    if (a >= r.z || b >= r.w) return;
    d_apply_at<HAS_ALPHA, 3>(v, fp, ph, 0, 0, 0, r.x + a, r.y + b);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
template <bool HAS_ALPHA, int MODE>
__global__ __launch_bounds__(256) void k_apply_m(const DV *__restrict__ vt, const FP *__restrict__ ft, suhmo_phys_t ph, int homog)
{
    d_apply<HAS_ALPHA, MODE>(vt[blockIdx.z], ft[blockIdx.z], ph, homog, 0, 0);
}

static int exchange_fields(suhmo_level *L, int depth, std::initializer_list<int> fields, hipStream_t st)
{
    const DV &v = L->d[depth].v;
    if (!(L->ex && (v.ext[0] || v.ext[1]))) return 0;
    for (int f : fields) if (!suhmo_field(L, depth, f)) return -2;
    int rc = L->ex(L->user, L, depth, fields.begin(), (int)fields.size(), (suhmo_stream_t)st);
    if (!rc) for (int f : fields) if (f == SUHMO_F_PHI) L->d[depth].phi_fresh = suhmo_halo_rows(v);
    return rc;
}
int suhmo_exchange_list(suhmo_level *L, int depth, const int *fields, int n, hipStream_t st)    // for suhmo_step.hip
{
    const DV &v = L->d[depth].v;
    if (!(L->ex && (v.ext[0] || v.ext[1]))) return 0;
    for (int k = 0; k < n; k++) if (!suhmo_field(L, depth, fields[k])) return -2;
    int rc = L->ex(L->user, L, depth, fields, n, (suhmo_stream_t)st);
    if (!rc) for (int k = 0; k < n; k++) if (fields[k] == SUHMO_F_PHI) L->d[depth].phi_fresh = suhmo_halo_rows(v);
    return rc;
}
// strips: make sure `need` halo rows of phi hold the neighbours' current values.  Every kernel that
// changes phi lowers Depth::phi_fresh (a colour pass that also advances the halo rows redundantly loses
// one row, a K-sweep fused launch 2K), so an exchange happens only when the stencil about to run would
// reach stale rows -- LevelData::exchange of the reference (src/VCAMRNonLinearPoissonOp.cpp:47,124,304,405,692)
int suhmo_ensure_phi_halo(suhmo_level *L, int depth, int need, hipStream_t st)
{
    Depth &D = L->d[depth];
    if (!(L->ex && (D.v.ext[0] || D.v.ext[1]))) return 0;
    if (D.phi_fresh >= need) return 0;
    return exchange_fields(L, depth, {SUHMO_F_PHI}, st);
}

extern "C" int suhmo_level_apply_op(suhmo_level_t *L, int depth, int homogeneous, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, (hipStream_t)s); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 0>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, homogeneous);
    else hipLaunchKernelGGL((k_apply<false, 0>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, homogeneous);
    HIPCHK(hipGetLastError());
    return 0;
}

// applyOpI (inhomogeneous) and the residual of it in one pass: LPHI = L(phi), RES = rhs - L(phi)
int suhmo_apply_and_residual(suhmo_level *L, int depth, hipStream_t st)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_RES)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 3>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph, 0);
    else hipLaunchKernelGGL((k_apply<false, 3>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, st, D.v, D.fp, L->ph, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

int suhmo_apply_and_residual_rects(suhmo_level *L, int depth, const int4 *d_rects, int n, int maxw, int maxh, hipStream_t st)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::applyOpI");
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_RES)) return -2;
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (!n) return 0;
    dim3 grd((maxw + 63) / 64, (maxh + 3) / 4, n);
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply_rects<true>), grd, BLK2D, 0, st, D.v, D.fp, L->ph, d_rects);
    else hipLaunchKernelGGL((k_apply_rects<false>), grd, BLK2D, 0, st, D.v, D.fp, L->ph, d_rects);
    HIPCHK(hipGetLastError());
    return 0;
}

// FAS cycle, coarse depth after the restriction: rhs_c = res_c + L_c(R phi) and the copy of R phi the prolongation
// subtracts (on a rank strip with its exchanged halo rows), one pass instead of applyOp + axby + a device copy
int suhmo_fas_coarse_rhs(suhmo_level *L, int depth, hipStream_t st, int hcomp)
{
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LPHI) || !suhmo_field(L, depth, SUHMO_F_PHIOLD)) return -2;
    const int halo = (D.v.ext[0] || D.v.ext[1]) ? D.v.gy : 0;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 2>), grid2d(D.v.nx, D.v.ny + 2 * halo), BLK2D, 0, st, D.v, D.fp, L->ph, 0, halo, hcomp);
    else hipLaunchKernelGGL((k_apply<false, 2>), grid2d(D.v.nx, D.v.ny + 2 * halo), BLK2D, 0, st, D.v, D.fp, L->ph, 0, halo, hcomp);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int suhmo_level_residual(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::residualI");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    int rc = suhmo_ensure_phi_halo(L, depth, 1, (hipStream_t)s); if (rc) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL((k_apply<true, 1>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, 0);
    else hipLaunchKernelGGL((k_apply<false, 1>), grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// COMPUTENONLINEARTERMS / lambda as stand-alone kernels (parity of a2, a9)
__global__ void k_nonlinear(DV v, FP fp, suhmo_phys_t ph)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double nl, dnl;
    nl_terms(ph, fp.f[SUHMO_F_PHI][idx], fp.f[SUHMO_F_B][idx], fp.f[SUHMO_F_PI][idx], fp.f[SUHMO_F_ZB][idx], fp.f[SUHMO_F_MASK][idx], nl, dnl);
    fp.f[SUHMO_F_NL][idx] = nl; fp.f[SUHMO_F_DNL][idx] = dnl;
}
__global__ void k_lambda(DV v, FP fp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double aterm = fp.f[SUHMO_F_ACOEF][idx] * v.alpha;
    fp.f[SUHMO_F_LAMBDA][idx] = lambda_cell(v, aterm, fp.f[SUHMO_F_BX][idx + 1], fp.f[SUHMO_F_BX][idx],
                                            fp.f[SUHMO_F_BY][idx + v.P], fp.f[SUHMO_F_BY][idx]);
}
extern "C" int suhmo_level_nonlinear(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_NL) || !suhmo_field(L, depth, SUHMO_F_DNL)) return -2;
    hipLaunchKernelGGL(k_nonlinear, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_compute_lambda(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (!suhmo_field(L, depth, SUHMO_F_LAMBDA)) return -2;
    hipLaunchKernelGGL(k_lambda, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int suhmo_level_gsrb(suhmo_level_t *L, int depth, int sweeps, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::levelGSRB");
    ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(sweeps >= 0);
    HIPCHK(hipSetDevice(L->device));
    int rc = suhmo_launch_gsrb(L, depth, sweeps, 0, (hipStream_t)s);
    if (rc) return rc;
    // levelGSRB leaves the ghosts with the HOMOGENEOUS BC applied (:757-759)
    if (sweeps > 0) return suhmo_level_fill_ghosts(L, depth, SUHMO_F_PHI, 1, s);
    return 0;
}

// RESTRICTRESVCNL2D (src/VCAMRNonLinearPoissonOpF.ChF:516-558) fused with BC + NL: one thread
// per coarse cell; the four fine contributions are accumulated in the reference's loop order
// (2I,2J), (2I+1,2J), (2I,2J+1), (2I+1,2J+1) onto a zero-initialised coarse value.
template <bool HAS_ALPHA>
__global__ __launch_bounds__(256) void k_restrict_residual(DV v, FP fp, DV vc, double *__restrict__ resC, double *__restrict__ phiC, suhmo_phys_t ph)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vc.nx || J >= vc.ny) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    // the thread's 2 x 2 fine cells start at an even column: every row is read as 16-byte pairs (canvas column
    // SUHMO_XOFF + 2I is 16-byte aligned), the W / E neighbours of the pair as single values
    const int i0 = 2 * I, j0 = 2 * J, base = cidx(v, i0, j0);
    auto ld2 = [&](const double *__restrict__ p, int idx) { return *reinterpret_cast<const double2 *>(p + idx); };
    double2 pc[2], pS, pN;
    pc[0] = ld2(phi, base); pc[1] = ld2(phi, base + v.P);
    // south of row j0 / north of row j0 + 1 (physical BC evaluated from the adjacent interior value)
    if (j0 > 0 || v.ext[0]) pS = ld2(phi, base - v.P);
    else if (v.per[1]) pS = ld2(phi, base + (v.ny - 1) * v.P);
    else { pS.x = phiS(v, phi, base, j0, pc[0].x, false); pS.y = phiS(v, phi, base + 1, j0, pc[0].y, false); }
    if (j0 + 1 < v.ny - 1 || v.ext[1]) pN = ld2(phi, base + 2 * v.P);
    else if (v.per[1]) pN = ld2(phi, base + v.P - (v.ny - 1) * v.P);
    else { pN.x = phiN(v, phi, base + v.P, j0 + 1, pc[1].x, false); pN.y = phiN(v, phi, base + v.P + 1, j0 + 1, pc[1].y, false); }
    double acc = 0.0, accp = 0.0;      // accp: RESTRICTVCNL of phi (restrictR), same visiting order
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int j = j0 + b, idx = base + b * v.P;
        const double2 cc = pc[b];
        const double w0 = phiW(v, phi, idx, i0, cc.x, false), e1 = phiE(v, phi, idx + 1, i0 + 1, cc.y, false);
        const double2 sS = b == 0 ? pS : pc[0], nN = b == 0 ? pc[1] : pN;
        const double2 bx01 = ld2(fp.f[SUHMO_F_BX], idx); const double bx2 = fp.f[SUHMO_F_BX][idx + 2];
        const double2 byS = ld2(fp.f[SUHMO_F_BY], idx), byN = ld2(fp.f[SUHMO_F_BY], idx + v.P);
        const double2 B2 = ld2(fp.f[SUHMO_F_B], idx), Pi2 = ld2(fp.f[SUHMO_F_PI], idx), zb2 = ld2(fp.f[SUHMO_F_ZB], idx), mk2 = ld2(fp.f[SUHMO_F_MASK], idx);
        const double2 rhs2 = ld2(fp.f[SUHMO_F_RHS], idx);
        double2 a2 = make_double2(0.0, 0.0);
        if (HAS_ALPHA) a2 = ld2(fp.f[SUHMO_F_ACOEF], idx);
        (void)j;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const double c = a ? cc.y : cc.x, w = a ? cc.x : w0, e = a ? e1 : cc.y;
            const double n = a ? nN.y : nN.x, s_ = a ? sS.y : sS.x;
            const double bxW = a ? bx01.y : bx01.x, bxE = a ? bx2 : bx01.y;
            double nl, dnl;
            nl_terms(ph, c, a ? B2.y : B2.x, a ? Pi2.y : Pi2.x, a ? zb2.y : zb2.x, a ? mk2.y : mk2.x, nl, dnl);
            double aterm = HAS_ALPHA ? v.alpha * (a ? a2.y : a2.x) : v.alpha;
            double lofphi = lofphi_cell(v, aterm, c, e, w, n, s_, bxE, bxW, a ? byN.y : byN.x, a ? byS.y : byS.x, nl);
            acc = acc + ((a ? rhs2.y : rhs2.x) - lofphi) / 4.0;
            accp = accp + c / 4.0;
        }
    }
    resC[cidx(vc, I, J)] = acc;
    if (phiC) phiC[cidx(vc, I, J)] = accp;
}

static int restrict_residual_impl(suhmo_level *L, int depth, bool also_phi, hipStream_t st)
{
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    int rc = suhmo_ensure_phi_halo(L, depth, 1, st); if (rc) return rc;
    if (also_phi) phi_changed(L, depth + 1);
    double *phiC = also_phi ? C.fp.f[SUHMO_F_PHI] : nullptr;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL(k_restrict_residual<true>, grid2d(C.v.nx, C.v.ny), BLK2D, 0, st, D.v, D.fp, C.v, C.fp.f[SUHMO_F_RES], phiC, L->ph);
    else hipLaunchKernelGGL(k_restrict_residual<false>, grid2d(C.v.nx, C.v.ny), BLK2D, 0, st, D.v, D.fp, C.v, C.fp.f[SUHMO_F_RES], phiC, L->ph);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_restrict_residual(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("VCAMRNonLinearPoissonOp::restrictResidual");
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    return restrict_residual_impl(L, depth, false, (hipStream_t)s);
}
// restrictResidual + restrictR of the FAS cycle in one pass over the fine level
int suhmo_restrict_both(suhmo_level *L, int depth, hipStream_t st) { return restrict_residual_impl(L, depth, true, st); }

// RESTRICTVCNL (src/VCAMRNonLinearPoissonOpF.ChF:432-446)
__global__ void k_restrict_r(DV v, const double *__restrict__ f, DV vc, double *__restrict__ c)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y * blockDim.y + threadIdx.y;
    if (I >= vc.nx || J >= vc.ny) return;
    int idx = cidx(v, 2 * I, 2 * J);
    double acc = 0.0;
    acc = acc + f[idx] / 4.0;
    acc = acc + f[idx + 1] / 4.0;
    acc = acc + f[idx + v.P] / 4.0;
    acc = acc + f[idx + v.P + 1] / 4.0;
    c[cidx(vc, I, J)] = acc;
}
extern "C" int suhmo_level_restrict_r(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth + 1);
    hipLaunchKernelGGL(k_restrict_r, grid2d(C.v.nx, C.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_PHI]);
    HIPCHK(hipGetLastError());
    return 0;
}

// PROLONGNL (src/AMRNonLinearPoissonOpF.ChF:617-628), m = 2
__global__ void k_prolong(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    phi[idx] = phi[idx] + c[cidx(vc, i / 2, j / 2)];
}
extern "C" int suhmo_level_prolong_increment(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::prolongIncrement");
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth);
    hipLaunchKernelGGL(k_prolong, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_CORR]);
    HIPCHK(hipGetLastError());
    return 0;
}

// FAS correction of the cycle: CORR_c = 1*phi_c + (-1)*phi_c,old (LevelDataOps::axby), phi += P(CORR_c) (PROLONGNL).
// On rank strips both kernels also cover the halo rows that are valid on BOTH depths (fine: phi_fresh rows, coarse:
// phi_fresh rows of phi_c; PHIOLD was copied after the exchange), so the post-smoothing can start without an exchange.
__global__ void k_axby_rows(DV v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b, int jlo, int jhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i >= v.nx || j > jhi) return;
    int idx = cidx(v, i, j);
    dst[idx] = a * x[idx] + b * y[idx];
}
__global__ void k_prolong_rows(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c, int jlo, int jhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = jlo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i >= v.nx || j > jhi) return;
    int idx = cidx(v, i, j);
    phi[idx] = phi[idx] + c[cidx(vc, i / 2, j >> 1)];          // j >> 1: floor, halo rows have j < 0
}
int suhmo_prolong_with_halo(suhmo_level *L, int depth, hipStream_t st)
{
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    const bool ext = L->ex && (D.v.ext[0] || D.v.ext[1]);
    int R = 0;                                              // fine halo rows that stay valid through the prolongation
    if (ext) { R = D.phi_fresh < 2 * C.phi_fresh ? D.phi_fresh : 2 * C.phi_fresh; R &= ~1; }
    const int Rc = R / 2;
    const int jlo = D.v.ext[0] ? -R : 0, jhi = D.v.ny - 1 + (D.v.ext[1] ? R : 0);
    const int cjlo = C.v.ext[0] ? -Rc : 0, cjhi = C.v.ny - 1 + (C.v.ext[1] ? Rc : 0);
    double *corr = suhmo_field(L, depth + 1, SUHMO_F_CORR);
    if (!corr) return -2;
    hipLaunchKernelGGL(k_axby_rows, grid2d(C.v.nx, cjhi - cjlo + 1), BLK2D, 0, st, C.v, corr, C.fp.f[SUHMO_F_PHI], C.fp.f[SUHMO_F_PHIOLD], 1.0, -1.0, cjlo, cjhi);
    hipLaunchKernelGGL(k_prolong_rows, grid2d(D.v.nx, jhi - jlo + 1), BLK2D, 0, st, D.v, D.fp.f[SUHMO_F_PHI], C.v, corr, jlo, jhi);
    HIPCHK(hipGetLastError());
    D.phi_fresh = R;
    return 0;
}

// PROLONG_2_NL (src/AMRNonLinearPoissonOpF.ChF:660-705), coarse data read with its stored ghosts
__global__ void k_prolong2(DV v, double *__restrict__ phi, DV vc, const double *__restrict__ c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    const double den = 1.0 / 16.0, fx1 = 3.0 * den, fx2 = 9.0 * den, f0 = 1.0 * den;
    int ic = i / 2, jc = j / 2, o1 = 2 * (i % 2) - 1, o2 = 2 * (j % 2) - 1;
    int idx = cidx(v, i, j), cc = cidx(vc, ic, jc);
    double p = phi[idx];
    p = p + fx2 * c[cc] + f0 * c[cc + o1 + o2 * vc.P];
    p = p + fx1 * (c[cc + o1] + c[cc + o2 * vc.P]);
    phi[idx] = p;
}
extern "C" int suhmo_level_prolong_bilinear(suhmo_level_t *L, int depth, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth + 1 < L->ndepth);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth], &C = L->d[depth + 1];
    phi_changed(L, depth);
    hipLaunchKernelGGL(k_prolong2, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_PHI], C.v, C.fp.f[SUHMO_F_CORR]);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ bCoef update (WFlx_level)
// step 1: cell-centred gradient = EdgeToCell(NEWMACGRAD) (util/Gradient.cpp:96-127, :623;
// util/GradientF.ChF:57-70)
__device__ __forceinline__ void d_gradcc_at(const DV &v, const FP &fp, int hasMask, int i, int j)
{
    if (i >= v.nx || j >= v.ny) return;
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
    fp.f[SUHMO_F_GRADX][idx] = 0.5 * (gW + gE);
    fp.f[SUHMO_F_GRADY][idx] = 0.5 * (gS + gN);
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
// step 3: COMPUTERE on the ghosted box (src/AmrHydro.cpp:1495-1505, src/AmrHydroF.ChF:92-109)
__device__ __forceinline__ void d_re(const DV &v, const FP &fp, suhmo_phys_t ph)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    bool xo = (i < 0 || i >= v.nx), yo = (j < 0 || j >= v.ny);
    if (xo && yo) return;                                      // corner ghosts are never read
    int idx = cidx(v, i, j);
    double gx = fp.f[SUHMO_F_GRADX][idx], gy = fp.f[SUHMO_F_GRADY][idx], B = fp.f[SUHMO_F_B][idx];
    double sg = sqrt(gx * gx + gy * gy);
    double discr = 1.0 + 4.0 * ph.omega * (B * B * B * ph.grav * sg) / (12.0 * ph.nu * ph.nu);
    fp.f[SUHMO_F_RE][idx] = (-1.0 + sqrt(discr)) / (2.0 * ph.omega);
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
        rc = exchange_fields(L, depth, {SUHMO_F_GRADX, SUHMO_F_GRADY}, st); if (rc) return rc;
        int n = 2 * D.v.ny + 2 * D.v.nx;
        hipLaunchKernelGGL(k_grad_ghosts, dim3((n + 255) / 256), dim3(256), 0, st, D.v, D.fp.f[SUHMO_F_GRADX], D.fp.f[SUHMO_F_GRADY]);
        hipLaunchKernelGGL(k_re, grid2d(D.v.nx + 2, D.v.ny + 2), BLK2D, 0, st, D.v, D.fp, L->ph);
        hipLaunchKernelGGL(k_bcoef_faces, grid2d(D.v.nx + 1, D.v.ny + 1), BLK2D, 0, st, D.v, D.fp, L->ph);
    }
    HIPCHK(hipGetLastError());
    // strips: the fused relaxation recomputes halo rows, so it needs the coefficients there too (faces_deferred: the V-cycle sends them
    // with the coarse depths' faces, one message for all depths: suhmo_average_operator_all)
    if (depth == 0 && L->faces_deferred) return 0;
    rc = exchange_fields(L, depth, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc;
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
    rc = exchange_fields(L, depth, {SUHMO_F_GRADX, SUHMO_F_GRADY}, st); if (rc) return rc;    // lvlgradH.exchange() :1490
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
__global__ __launch_bounds__(256) void k_average_faces_x_all(DV vf, const double *__restrict__ bxf, AvgOut o, int nd)
{
    const int R = 1 << (nd - 1);
    int ih = blockIdx.x * blockDim.x + threadIdx.x;      // i = 2 * ih
    int jb = blockIdx.y * blockDim.y + threadIdx.y;
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
__global__ __launch_bounds__(256) void k_average_faces_y_all(DV vf, const double *__restrict__ byf, AvgOut o, int nd)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const int jb = 2 * AVG_YR * (blockIdx.y * (blockDim.x / 64) + (threadIdx.x >> 6));
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
    int rc = exchange_fields(L, depth, {SUHMO_F_BX, SUHMO_F_BY}, (hipStream_t)s); if (rc) return rc;
    return 0;
}

int suhmo_average_operator_all(suhmo_level *L, int nd, hipStream_t st)
{
    Depth &F = L->d[0];
    const bool d0 = L->faces_deferred != 0;          // the halo rows of the depth-0 faces are still to travel
    L->faces_deferred = 0;
    if (nd < 2) return d0 ? exchange_fields(L, 0, {SUHMO_F_BX, SUHMO_F_BY}, st) : 0;
    if (nd > 7 || F.v.nx % (1 << (nd - 1)) || F.v.ny % (1 << (nd - 1))) {     // generic fallback
        if (d0) { int rc = exchange_fields(L, 0, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc; }
        for (int k = 1; k < nd; k++) { int rc = suhmo_level_average_operator(L, k, (suhmo_stream_t)st); if (rc) return rc; }
        return suhmo_agg_gather_faces(L, nd, st);
    }
    AvgOut o;
    for (int d = 0; d < nd; d++) { o.bx[d] = L->d[d].fp.f[SUHMO_F_BX]; o.by[d] = L->d[d].fp.f[SUHMO_F_BY]; o.P[d] = L->d[d].v.P; o.gy[d] = L->d[d].v.gy; }
    const int R = 1 << (nd - 1);
    dim3 gx((F.v.nx / 2 + 1 + 63) / 64, (F.v.ny / R + 3) / 4);
    hipLaunchKernelGGL(k_average_faces_x_all, gx, dim3(64, 4), 0, st, F.v, F.fp.f[SUHMO_F_BX], o, nd);
    dim3 gy((F.v.nx + 63) / 64, ((F.v.ny / 2 + 1 + AVG_YR - 1) / AVG_YR + 3) / 4);
    hipLaunchKernelGGL(k_average_faces_y_all, gy, dim3(256), 0, st, F.v, F.fp.f[SUHMO_F_BY], o, nd);
    HIPCHK(hipGetLastError());
    // strips: the coarse face coefficients of all depths travel as one message group when the transport can batch
    if (L->ipc) { int rc = suhmo_ipc_batch(L, 1, st); if (rc) return rc; }
    else if (L->ex_begin && L->ex) { int rc = L->ex_begin(L->user); if (rc) return rc; }
    for (int k = d0 ? 0 : 1; k < nd && !(L->agg && k >= L->agg_depth); k++) {
        int rc = exchange_fields(L, k, {SUHMO_F_BX, SUHMO_F_BY}, st); if (rc) return rc;
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
        rc = exchange_fields(L, dep, {SUHMO_F_ACOEF, SUHMO_F_B, SUHMO_F_PI, SUHMO_F_ZB, SUHMO_F_MASK}, st); if (rc) return rc;
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

// ------------------------------------------------------------------ small operators
// DIVERGENCE (util/DivergenceF.ChF:38-54), called for dir 0 then dir 1
__global__ void k_divergence(DV v, const double *__restrict__ ux, const double *__restrict__ uy, double *__restrict__ div)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    double d = div[idx];
    d = d + v.fdx * (ux[idx + 1] - ux[idx]);
    d = d + v.fdy * (uy[idx + v.P] - uy[idx]);
    div[idx] = d;
}
extern "C" int suhmo_level_divergence(suhmo_level_t *L, int depth, int dst_field, suhmo_stream_t s)
{
    CHECK_DF(L, depth, dst_field); ARG(!is_face(dst_field));
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    double *dst = suhmo_field(L, depth, dst_field);
    hipLaunchKernelGGL(k_divergence, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, D.fp.f[SUHMO_F_BX], D.fp.f[SUHMO_F_BY], dst);
    HIPCHK(hipGetLastError());
    return 0;
}

// getFlux (src/VCAMRNonLinearPoissonOp.cpp:820-840): F = -b * ((phi_hi - phi_lo) * (beta*ref/dx))
__global__ void k_getflux(DV v, FP fp, int dir, double scale, double *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    int nxf = dir == 0 ? v.nx + 1 : v.nx, nyf = dir == 0 ? v.ny : v.ny + 1;
    if (i >= nxf || j >= nyf) return;
    const double *__restrict__ phi = fp.f[SUHMO_F_PHI];
    int idx = cidx(v, i, j);
    double phihi = phi[idx], philo = dir == 0 ? phi[idx - 1] : phi[idx - v.P];   // stored ghosts
    double gradphi = (phihi - philo) * scale;
    out[(size_t)j * nxf + i] = -fp.f[dir == 0 ? SUHMO_F_BX : SUHMO_F_BY][idx] * gradphi;
}
extern "C" int suhmo_level_get_flux(suhmo_level_t *L, int depth, int dir, int ref, double *flux_host, suhmo_stream_t s)
{
    ARG(L); ARG(depth >= 0 && depth < L->ndepth); ARG(dir == 0 || dir == 1); ARG(flux_host);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    int rc = suhmo_level_fill_ghosts(L, depth, SUHMO_F_PHI, 0, s); if (rc) return rc;
    int nxf = dir == 0 ? D.v.nx + 1 : D.v.nx, nyf = dir == 0 ? D.v.ny : D.v.ny + 1;
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, (size_t)nxf * nyf * 8));
    double scale = D.v.beta * ref / (dir == 0 ? D.v.dx : D.v.dy);
    hipLaunchKernelGGL(k_getflux, grid2d(nxf, nyf), BLK2D, 0, st, D.v, D.fp, dir, scale, tmp);
    HIPCHK(hipMemcpyAsync(flux_host, tmp, (size_t)nxf * nyf * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipFree(tmp));
    return 0;
}

// LevelDataOps::axby / setVal on valid cells
__device__ __forceinline__ void d_axby(const DV &v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    int idx = cidx(v, i, j);
    dst[idx] = a * x[idx] + b * y[idx];
}
__global__ void k_axby(DV v, double *__restrict__ dst, const double *__restrict__ x, const double *__restrict__ y, double a, double b)
{
    d_axby(v, dst, x, y, a, b);
}
// every box of a multi-box AMR level in one launch (blockIdx.z = box; suhmo_hier.hip)
__global__ void k_axby_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int fd, int fx, int fy, double a, double b)
{
    d_axby(vt[blockIdx.z], ft[blockIdx.z].f[fd], ft[blockIdx.z].f[fx], ft[blockIdx.z].f[fy], a, b);
}
__global__ void k_setval(DV v, double *__restrict__ dst, double val)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= v.nx || j >= v.ny) return;
    dst[cidx(v, i, j)] = val;
}
extern "C" int suhmo_level_axby(suhmo_level_t *L, int depth, int dst, int x, int y, double a, double b, suhmo_stream_t s)
{
    CHECK_DF(L, depth, dst); CHECK_DF(L, depth, x); CHECK_DF(L, depth, y);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    double *pd = suhmo_field(L, depth, dst), *px = suhmo_field(L, depth, x), *py = suhmo_field(L, depth, y);
    if (dst == SUHMO_F_PHI) phi_changed(L, depth);
    if (dst == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }      // (the reports about the ice mask end with any write to it)
    hipLaunchKernelGGL(k_axby, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, pd, px, py, a, b);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int suhmo_level_set_value(suhmo_level_t *L, int depth, int field, double val, suhmo_stream_t s)
{
    CHECK_DF(L, depth, field);
    HIPCHK(hipSetDevice(L->device));
    Depth &D = L->d[depth];
    if (field == SUHMO_F_PHI) phi_changed(L, depth);
    if (field == SUHMO_F_MASK) { L->coarse_mask_ok = 0; L->maskflag_epoch = 0; }
    hipLaunchKernelGGL(k_setval, grid2d(D.v.nx, D.v.ny), BLK2D, 0, (hipStream_t)s, D.v, suhmo_field(L, depth, field), val);
    HIPCHK(hipGetLastError());
    return 0;
}

// norms over valid cells.  ord 0: max |x| (exact, order independent).  ord 2: sqrt(sum x^2),
// two-stage deterministic reduction (fixed partial order; differs from the serial CPU sum
// by rounding only).
__global__ __launch_bounds__(256) void k_norm_partial(DV v, const double *__restrict__ x, int ord, double *__restrict__ partial)
{
    __shared__ double sm[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) {
            double val = x[cidx(v, i, j)];
            if (ord == 0) acc = fmax(acc, fabs(val)); else acc += val * val;
        }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] = ord == 0 ? fmax(sm[tid], sm[tid + s]) : sm[tid] + sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}
__global__ void k_norm_final(const double *__restrict__ partial, int n, int ord, double *__restrict__ out, HostSlot hs)
{
    __shared__ double sm[256];
    int tid = threadIdx.x;
    double acc = 0.0;
    for (int k = tid; k < n; k += 256) acc = ord == 0 ? fmax(acc, partial[k]) : acc + partial[k];
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] = ord == 0 ? fmax(sm[tid], sm[tid + s]) : sm[tid] + sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) { out[0] = sm[0]; suhmo_publish(hs, sm[0]); }
}
// The 8-byte result of a reduction.  Synchronising the stream costs ~17 us of idle GPU per read-back on this platform (the copy
// kernel, the interrupt, the wake-up); instead the reduction's last kernel stores the value and then, with a system-scope release,
// a sequence number into pinned coherent host memory, and the host spins on the number.  In-order stream: when the number is
// there, everything enqueued before is done.  A kernel that never publishes (a fault) ends in the stream synchronisation below,
// which reports it.
HostSlot suhmo_host_slot(suhmo_level *L)
{
    HostSlot h{nullptr, nullptr, 0};
    if (L->poll_readback) { h.val = L->hscratch_dev; h.flag = (unsigned long long *)(L->hscratch_dev + 8); h.seq = ++L->hseq; }
    return h;
}
int suhmo_readback(suhmo_level *L, hipStream_t st, double *out, double *out2)
{
    if (L->poll_readback) {
        HIPCHK(hipGetLastError());
        volatile unsigned long long *flag = (volatile unsigned long long *)(L->hscratch + 8);
        for (long spin = 0; spin < 400000000L; spin++) {
            if (__atomic_load_n((unsigned long long *)flag, __ATOMIC_ACQUIRE) == L->hseq) { *out = L->hscratch[0]; if (out2) *out2 = L->hscratch[1]; return 0; }
            if ((spin & 0xffff) == 0xffff && hipStreamQuery(st) == hipSuccess) break;     // finished without us seeing the store: read below
        }
        HIPCHK(hipStreamSynchronize(st));
        if (__atomic_load_n((unsigned long long *)flag, __ATOMIC_ACQUIRE) == L->hseq) { *out = L->hscratch[0]; if (out2) *out2 = L->hscratch[1]; return 0; }
        // not published (mapping not coherent on this system): fall back for good
        L->poll_readback = 0;
    }
    HIPCHK(hipMemcpyAsync(L->hscratch, L->scratch, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out = L->hscratch[0];
    if (out2) *out2 = L->hscratch[1];
    return 0;
}
static inline bool on_strip(const suhmo_level *L) { const DV &v = L->d[0].v; return v.rk[0] || v.rk[1]; }
__global__ void k_publish2(const double *__restrict__ v, HostSlot hs)
{
    if (hs.val) hs.val[1] = v[1];
    suhmo_publish(hs, v[0]);
}
HostSlot suhmo_reduce_slot(suhmo_level *L)
{
    if (on_strip(L) && L->ard) return HostSlot{nullptr, nullptr, 0};       // published after the device all-reduce
    return suhmo_host_slot(L);
}
int suhmo_reduce_finish(suhmo_level *L, hipStream_t st, int n, int op, double *out, double *out2)
{
    ARG(n == 1 || n == 2);
    double v[2] = {0.0, 0.0};
    int rc;
    if (on_strip(L) && L->ard) {
        if ((rc = L->ard(L->user, L->scratch, n, op, (suhmo_stream_t)st))) return rc;
        hipLaunchKernelGGL(k_publish2, dim3(1), dim3(1), 0, st, L->scratch, suhmo_host_slot(L));
        HIPCHK(hipGetLastError());
        if ((rc = suhmo_readback(L, st, &v[0], &v[1]))) return rc;
    } else {
        if ((rc = suhmo_readback(L, st, &v[0], &v[1]))) return rc;
        if (on_strip(L) && (L->ar || L->ar2)) {
            if (L->ar2) { if ((rc = L->ar2(L->user, v, n, op))) return rc; }
            else if (op == 0) { for (int k = 0; k < n; k++) if ((rc = L->ar(L->user, &v[k]))) return rc; }
            else { suhmo_set_error("a SUM over the ranks of a strip needs suhmo_level_set_reduce_hook or the native transport (the hook of suhmo_level_set_hooks reduces MAX only)"); return -5; }
        }
    }
    *out = v[0];
    if (out2) *out2 = v[1];
    return 0;
}
__global__ __launch_bounds__(256) void k_dot_partial(DV v, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partial)
{
    __shared__ double sm[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) { int idx = cidx(v, i, j); acc += x[idx] * y[idx]; }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = sm[tid] + sm[tid + s]; __syncthreads(); }
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}
extern "C" int suhmo_level_dot(suhmo_level_t *L, int depth, int x, int y, double *out, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::dotProduct");
    CHECK_DF(L, depth, x); CHECK_DF(L, depth, y); ARG(out);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    const double *px = suhmo_field(L, depth, x), *py = suhmo_field(L, depth, y);
    if (!px || !py) { suhmo_set_error("field allocation failed"); return -2; }
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    hipLaunchKernelGGL(k_dot_partial, grd, BLK2D, 0, st, D.v, px, py, L->scratch + 2);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, (int)(grd.x * grd.y), 2, L->scratch, suhmo_reduce_slot(L));
    return suhmo_reduce_finish(L, st, 1, 1, out);
}
extern "C" int suhmo_level_norm(suhmo_level_t *L, int depth, int field, int ord, double *out, suhmo_stream_t s)
{
    SUHMO_TIME("AMRNonLinearPoissonOp::norm");
    CHECK_DF(L, depth, field); ARG(out); ARG(ord == 0 || ord == 2);
    HIPCHK(hipSetDevice(L->device));
    hipStream_t st = (hipStream_t)s;
    Depth &D = L->d[depth];
    dim3 grd(std::min((D.v.nx + 63) / 64, 32), std::min((D.v.ny + 3) / 4, 128));
    int np = grd.x * grd.y;
    hipLaunchKernelGGL(k_norm_partial, grd, BLK2D, 0, st, D.v, suhmo_field(L, depth, field), ord, L->scratch + 2);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, np, ord, L->scratch, suhmo_reduce_slot(L));
    double r = 0.0;
    // the reference's norm() reduces over the ranks (src/AMRNonLinearPoissonOp.cpp:1222-1264): MAX for the max norm, SUM of the squares for l2
    { int rc = suhmo_reduce_finish(L, st, 1, ord == 0 ? 0 : 1, &r); if (rc) return rc; }
    if (ord == 2) r = sqrt(r);
    *out = r;
    return 0;
}

// residualI of depth 0 and the max norm of the result in two launches instead of three (levels of up to scratch-many workgroups)
int suhmo_level_residual_and_norm(suhmo_level *L, double *out, hipStream_t st)
{
    Depth &D = L->d[0];
    const dim3 grd = grid2d(D.v.nx, D.v.ny);
    const size_t np = (size_t)grd.x * grd.y;
    int rc;
    if (np + 4 >= L->scratch_elems) {
        if ((rc = suhmo_level_residual(L, 0, (suhmo_stream_t)st))) return rc;
        return suhmo_level_norm(L, 0, SUHMO_F_RES, 0, out, (suhmo_stream_t)st);
    }
    if ((rc = suhmo_ensure_phi_halo(L, 0, 1, st))) return rc;
    if (D.v.alpha != 0.0) hipLaunchKernelGGL(k_residual_norm<true>, grd, BLK2D, 0, st, D.v, D.fp, L->ph, L->scratch + 2);
    else hipLaunchKernelGGL(k_residual_norm<false>, grd, BLK2D, 0, st, D.v, D.fp, L->ph, L->scratch + 2);
    HIPCHK(hipGetLastError());
    return suhmo_level_norm_from_partials(L, (int)np, out, st);
}
// max norm of RES at depth 0 from the partial maxima the cycle's last launch left behind (suhmo_gsrb.hip, residual output): the second
// stage of suhmo_level_norm alone
int suhmo_level_norm_from_partials(suhmo_level *L, int np, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, L->scratch + 2, np, 0, L->scratch, suhmo_reduce_slot(L));
    double r = 0.0;
    int rc = suhmo_reduce_finish(L, st, 1, 0, &r); if (rc) return rc;
    *out = r;
    return 0;
}

// ------------------------------------------------------------------ every box of a multi-box AMR level in one launch
__global__ void k_copy_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int fd, int fs)
{
    const DV &v = vt[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    ft[blockIdx.z].f[fd][idx] = ft[blockIdx.z].f[fs][idx];
}
// A level of boxes entering / leaving its FAS problem in an AMR V-cycle (suhmo_hier.hip:vcycle_amr), one launch each instead of three / two:
//   enter: RHS0 <- RHS (copy, ghost ring included), RHS <- 1 RES + 1 LPHI (axby, valid cells), PHIOLD <- PHI (copy)
//   leave: RHS <- RHS0 (copy), CORR <- 1 PHI + (-1) PHIOLD (axby)               -- the expressions of k_copy_m / k_axby_m on the same operands
__global__ void k_fas_enter_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    const DV &v = vt[blockIdx.z];
    const FP &f = ft[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    f.f[SUHMO_F_RHS0][idx] = f.f[SUHMO_F_RHS][idx];
    f.f[SUHMO_F_PHIOLD][idx] = f.f[SUHMO_F_PHI][idx];
    if (i >= 0 && i < v.nx && j >= 0 && j < v.ny) f.f[SUHMO_F_RHS][idx] = 1.0 * f.f[SUHMO_F_RES][idx] + 1.0 * f.f[SUHMO_F_LPHI][idx];
}
__global__ void k_fas_leave_m(const DV *__restrict__ vt, const FP *__restrict__ ft)
{
    const DV &v = vt[blockIdx.z];
    const FP &f = ft[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    f.f[SUHMO_F_RHS][idx] = f.f[SUHMO_F_RHS0][idx];
    if (i >= 0 && i < v.nx && j >= 0 && j < v.ny) f.f[SUHMO_F_CORR][idx] = 1.0 * f.f[SUHMO_F_PHI][idx] + -1.0 * f.f[SUHMO_F_PHIOLD][idx];
}
// fields of the boxes of one hierarchy <- fields of the same boxes of another (the implicit gap-height operator's copy of a level)
struct CopyPairs { int n, fd[4], fs[4]; };
__global__ void k_copy_between_m(const DV *__restrict__ vt, const FP *__restrict__ fdst, const FP *__restrict__ fsrc, CopyPairs cp)
{
    const DV &v = vt[blockIdx.z];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
    if (i > v.nx || j > v.ny) return;
    int idx = cidx(v, i, j);
    for (int q = 0; q < cp.n; q++) fdst[blockIdx.z].f[cp.fd[q]][idx] = fsrc[blockIdx.z].f[cp.fs[q]][idx];
}
__global__ __launch_bounds__(256) void k_norm_max_partial_m(const DV *__restrict__ vt, const FP *__restrict__ ft, int field, double *__restrict__ partial)
{
    __shared__ double sm[256];
    const DV &v = vt[blockIdx.z];
    const double *__restrict__ x = ft[blockIdx.z].f[field];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int j = blockIdx.y * blockDim.y + threadIdx.y; j < v.ny; j += gridDim.y * blockDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.nx; i += gridDim.x * blockDim.x) acc = fmax(acc, fabs(x[cidx(v, i, j)]));
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] = fmax(sm[tid], sm[tid + s]); __syncthreads(); }
    if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sm[0];
}
static inline dim3 grid_m(const suhmo_multi &m, int ex = 0, int ey = 0) { return dim3((m.maxnx + ex + 63) / 64, (m.maxny + ey + 3) / 4, m.nbox); }
int suhmo_multi_fill_ghosts(const suhmo_multi &m, int field, int homog, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_fill_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp, field, homog);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_apply(const suhmo_multi &m, const suhmo_phys_t &ph, bool has_alpha, int mode, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    if (has_alpha) { if (mode == 0) hipLaunchKernelGGL((k_apply_m<true, 0>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
                     else if (mode == 1) hipLaunchKernelGGL((k_apply_m<true, 1>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
                     else hipLaunchKernelGGL((k_apply_m<true, 3>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0); }
    else { if (mode == 0) hipLaunchKernelGGL((k_apply_m<false, 0>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
           else if (mode == 1) hipLaunchKernelGGL((k_apply_m<false, 1>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0);
           else hipLaunchKernelGGL((k_apply_m<false, 3>), grid_m(m), BLK2D, 0, st, m.dv, m.fp, ph, 0); }
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_grad_cc(const suhmo_multi &m, int hasMask, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_gradcc_m, grid_m(m), BLK2D, 0, st, m.dv, m.fp, hasMask);
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_grad_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp);
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
int suhmo_multi_coef_ghosts(const suhmo_multi &m, int field, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    int n = 2 * m.maxny + 2 * m.maxnx;
    hipLaunchKernelGGL(k_coef_ghosts_m, dim3((n + 255) / 256, 1, m.nbox), dim3(256), 0, st, m.dv, m.fp, field);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_axby(const suhmo_multi &m, int fd, int fx, int fy, double a, double b, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_axby_m, grid_m(m), BLK2D, 0, st, m.dv, m.fp, fd, fx, fy, a, b);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_copy(const suhmo_multi &m, int fd, int fs, hipStream_t st)
{
    if (m.nbox <= 0) return 0;                       // a rank that owns no box of the level
    hipLaunchKernelGGL(k_copy_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp, fd, fs);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_fas_enter(const suhmo_multi &m, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    hipLaunchKernelGGL(k_fas_enter_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_fas_leave(const suhmo_multi &m, hipStream_t st)
{
    if (m.nbox <= 0) return 0;
    hipLaunchKernelGGL(k_fas_leave_m, grid_m(m, 2, 2), BLK2D, 0, st, m.dv, m.fp);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_copy_between(const suhmo_multi &dst, const suhmo_multi &src, const int *fd, const int *fs, int n, hipStream_t st)
{
    if (n < 1 || n > 4 || dst.nbox != src.nbox) { suhmo_set_error("internal: copy between hierarchies"); return -4; }
    if (src.nbox <= 0) return 0;
    CopyPairs cp;
    cp.n = n;
    for (int q = 0; q < n; q++) { cp.fd[q] = fd[q]; cp.fs[q] = fs[q]; }
    hipLaunchKernelGGL(k_copy_between_m, grid_m(src, 2, 2), BLK2D, 0, st, src.dv, dst.fp, src.fp, cp);
    HIPCHK(hipGetLastError());
    return 0;
}
int suhmo_multi_norm_max(const suhmo_multi &m, suhmo_level *slot, int field, double *out, hipStream_t st)
{
    if (m.nbox <= 0) { *out = 0.0; return 0; }
    dim3 grd(std::min((m.maxnx + 63) / 64, 4), std::min((m.maxny + 3) / 4, 16), m.nbox);
    hipLaunchKernelGGL(k_norm_max_partial_m, grd, BLK2D, 0, st, m.dv, m.fp, field, m.red);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, st, m.red, (int)(grd.x * grd.y * grd.z), 0, slot->scratch, suhmo_host_slot(slot));
    HIPCHK(hipGetLastError());
    return suhmo_readback(slot, st, out);
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
