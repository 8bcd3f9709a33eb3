// suhmo_rccl.hip -- native strip-halo transport: RCCL point-to-point over xGMI, enqueued on the
// same HIP stream as the kernels (no host round trip per exchange).
//
// Replaces, for a level cut into row strips (one process per GPU), what the reference does with
// LevelData::exchange over MPI (src/VCAMRNonLinearPoissonOp.cpp:47,124,304,405,692,751) and the
// MPI_Allreduce inside norm() (src/AMRNonLinearPoissonOp.cpp:1222-1264): per exchange
//   1 pack kernel (all fields, both sides) -> ncclGroup{send hi, send lo, recv lo, recv hi} -> 1 unpack kernel.
// librccl is NOT a link dependency: it is dlopen'ed (the copy PyTorch already loaded, when the host is
// Python), so a single-GPU user never needs it.  The communicator is created from an id the host
// distributes (torch.distributed broadcast in suhmo_amd/multigpu.py; MPI_Bcast in a Chombo build).
#include "suhmo_common.h"
#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <vector>

namespace {
struct Fns {
    void *dl = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
} g;

constexpr int MAXF = 8;           // fields per message
struct PackList { double *p[MAXF]; int pack_lo[MAXF], pack_hi[MAXF], unpack_lo[MAXF], unpack_hi[MAXF]; int n; };
struct Pending { int depth; PackList pl; size_t cnt; int rows; };
// zero-copy message of one field: whole canvas rows (pitch P, padding columns included) are contiguous, so the edge rows
// are sent from, and the halo rows received into, the canvas itself -- no pack / unpack kernels, no staging buffers
struct Direct { double *send_lo, *send_hi, *recv_lo, *recv_hi; size_t cnt; };
struct Strip {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, lo = -1, hi = -1;
    hipStream_t st = nullptr;     // stream of the norm all-reduce (the hook carries none)
    double *buf[SUHMO_MAXDEPTH][4] = {};   // send lo, send hi, recv lo, recv hi
    double *dscalar = nullptr;
    long exchanges = 0;
    bool batching = false;        // between ex_begin and ex_end: packs run at once, the transfers of all queued messages
    std::vector<Pending> queue;   // form ONE ncclGroup (one kernel), the unpacks follow
    bool direct = false;          // env SUHMO_RCCL_DIRECT=1: zero-copy rows instead of pack / one message per neighbour / unpack.
                                  // Same speed when a GPU is its own neighbour (3.01 vs 2.99 ms per 4096^2 V-cycle); it issues
                                  // one transfer per field where the staged path issues one per neighbour, so staged stays default
    std::vector<Direct> dqueue;
};

#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { \
    suhmo_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, g.GetErrorString ? g.GetErrorString(r_) : "rccl error"); return -7; } } while (0)

// rows travel (nx + 1) wide (x-face rows whole); z = 2 * field + side
__global__ void k_pack_multi(DV v, PackList pl, int rows, double *__restrict__ blo, double *__restrict__ bhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    int q = blockIdx.z >> 1, side = blockIdx.z & 1;
    double *b = side ? bhi : blo;
    if (i > v.nx || r >= rows || !b) return;
    int j = (side ? pl.pack_hi[q] : pl.pack_lo[q]) + r;
    b[((size_t)q * rows + r) * (v.nx + 1) + i] = pl.p[q][cidx(v, i, j)];
}
__global__ void k_unpack_multi(DV v, PackList pl, int rows, const double *__restrict__ blo, const double *__restrict__ bhi)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
    int q = blockIdx.z >> 1, side = blockIdx.z & 1;
    const double *b = side ? bhi : blo;
    if (i > v.nx || r >= rows || !b) return;
    int j = (side ? pl.unpack_hi[q] : pl.unpack_lo[q]) + r;
    pl.p[q][cidx(v, i, j)] = b[((size_t)q * rows + r) * (v.nx + 1) + i];
}

// send / receive every queued message in one group, then unpack them
// one ncclGroup for a list of zero-copy messages; order matters when lo == hi (2 ranks, periodic; or a rank that is its own
// neighbour): per field to-hi before to-lo, from-lo before from-hi, fields in the same sequence on both sides
int send_direct(Strip *S, const std::vector<Direct> &ops, hipStream_t st)
{
    if (ops.empty()) return 0;
    NCCLCHK(g.GroupStart());
    for (const Direct &d : ops) {
        if (S->hi >= 0) NCCLCHK(g.Send(d.send_hi, d.cnt, ncclFloat64, S->hi, S->comm, st));
        if (S->lo >= 0) NCCLCHK(g.Send(d.send_lo, d.cnt, ncclFloat64, S->lo, S->comm, st));
    }
    for (const Direct &d : ops) {
        if (S->lo >= 0) NCCLCHK(g.Recv(d.recv_lo, d.cnt, ncclFloat64, S->lo, S->comm, st));
        if (S->hi >= 0) NCCLCHK(g.Recv(d.recv_hi, d.cnt, ncclFloat64, S->hi, S->comm, st));
    }
    NCCLCHK(g.GroupEnd());
    return 0;
}
int flush(Strip *S, suhmo_level_t *L, hipStream_t st)
{
    if (S->direct) { int rc = send_direct(S, S->dqueue, st); S->dqueue.clear(); return rc; }
    if (S->queue.empty()) return 0;
    NCCLCHK(g.GroupStart());
    for (const Pending &q : S->queue) {
        double **B = S->buf[q.depth];
        if (S->hi >= 0) NCCLCHK(g.Send(B[1], q.cnt, ncclFloat64, S->hi, S->comm, st));
        if (S->lo >= 0) NCCLCHK(g.Send(B[0], q.cnt, ncclFloat64, S->lo, S->comm, st));
    }
    for (const Pending &q : S->queue) {
        double **B = S->buf[q.depth];
        if (S->lo >= 0) NCCLCHK(g.Recv(B[2], q.cnt, ncclFloat64, S->lo, S->comm, st));
        if (S->hi >= 0) NCCLCHK(g.Recv(B[3], q.cnt, ncclFloat64, S->hi, S->comm, st));
    }
    NCCLCHK(g.GroupEnd());
    for (const Pending &q : S->queue) {
        const DV &v = L->d[q.depth].v;
        double **B = S->buf[q.depth];
        dim3 blk(64, 4), grd((v.nx + 1 + 63) / 64, (q.rows + 3) / 4, 2 * q.pl.n);
        hipLaunchKernelGGL(k_unpack_multi, grd, blk, 0, st, v, q.pl, q.rows, S->lo >= 0 ? B[2] : nullptr, S->hi >= 0 ? B[3] : nullptr);
    }
    HIPCHK(hipGetLastError());
    S->queue.clear();
    return 0;
}
int begin_hook(void *user) { ((Strip *)user)->batching = true; return 0; }
int end_hook(void *user, suhmo_level_t *L, suhmo_stream_t s)
{
    Strip *S = (Strip *)user;
    S->batching = false;
    return flush(S, L, (hipStream_t)s);
}

int exchange_hook(void *user, suhmo_level_t *L, int depth, const int *fields, int nfields, suhmo_stream_t s)
{
    Strip *S = (Strip *)user;
    hipStream_t st = (hipStream_t)s;
    const DV &v = L->d[depth].v;
    const int rows = v.gy < v.ny ? v.gy : v.ny;
    if (S->direct) {
        std::vector<Direct> ops;
        for (int q = 0; q < nfields; q++) {
            const int f = fields[q];
            double *p = suhmo_field(L, depth, f);
            if (!p) { suhmo_set_error("field allocation failed"); return -2; }
            // owned rows next to each side / ghost rows of each side.  y-faces: face row 0 of a strip IS face row ny of the
            // lower neighbour (both own it): the rows the lower neighbour lacks start at face row 1 and land from ny + 1
            auto row = [&](int j) { return p + (size_t)(j + v.gy) * v.P; };
            Direct d;
            d.send_lo = row(f == SUHMO_F_BY ? 1 : 0); d.send_hi = row(v.ny - rows);
            d.recv_lo = row(-rows); d.recv_hi = row(f == SUHMO_F_BY ? v.ny + 1 : v.ny);
            d.cnt = (size_t)rows * v.P;
            ops.push_back(d);
        }
        S->exchanges++;
        if (S->batching) { S->dqueue.insert(S->dqueue.end(), ops.begin(), ops.end()); return 0; }
        return send_direct(S, ops, st);
    }
    const size_t n = (size_t)rows * (v.nx + 1);
    double **B = S->buf[depth];
    for (int f0 = 0; f0 < nfields; f0 += MAXF) {
        PackList pl;
        pl.n = nfields - f0 < MAXF ? nfields - f0 : MAXF;
        for (int q = 0; q < pl.n; q++) {
            int f = fields[f0 + q];
            pl.p[q] = suhmo_field(L, depth, f);
            if (!pl.p[q]) { suhmo_set_error("field allocation failed"); return -2; }
            // owned rows next to each side / ghost rows of each side, ascending j.  y-faces: face row 0 of
            // a strip IS face row ny of the lower neighbour (both own it): the rows the lower neighbour
            // lacks start at face row 1 and land from face row ny + 1 (as suhmo_level_pack/unpack_rows)
            pl.pack_lo[q] = f == SUHMO_F_BY ? 1 : 0;
            pl.pack_hi[q] = v.ny - rows;
            pl.unpack_lo[q] = -rows;
            pl.unpack_hi[q] = f == SUHMO_F_BY ? v.ny + 1 : v.ny;
        }
        dim3 blk(64, 4), grd((v.nx + 1 + 63) / 64, (rows + 3) / 4, 2 * pl.n);
        const size_t cnt = n * pl.n;
        bool queue_it = false;
        if (S->batching) {
            // one message per depth in a batch (the staging buffers are per depth).  A second message of a depth must not be
            // packed over the queued one: flush what is queued BEFORE packing, then send this one on its own
            bool clash = nfields > MAXF;
            for (const Pending &q : S->queue) clash = clash || q.depth == depth;
            if (clash) { int rc = flush(S, L, st); if (rc) return rc; }
            else queue_it = true;
        }
        hipLaunchKernelGGL(k_pack_multi, grd, blk, 0, st, v, pl, rows, S->lo >= 0 ? B[0] : nullptr, S->hi >= 0 ? B[1] : nullptr);
        HIPCHK(hipGetLastError());
        if (queue_it) { S->queue.push_back(Pending{depth, pl, cnt, rows}); S->exchanges++; continue; }
        // order matters when lo == hi (2 ranks, periodic; or a rank that is its own neighbour):
        // to-hi before to-lo, from-lo before from-hi
        NCCLCHK(g.GroupStart());
        if (S->hi >= 0) NCCLCHK(g.Send(B[1], cnt, ncclFloat64, S->hi, S->comm, st));
        if (S->lo >= 0) NCCLCHK(g.Send(B[0], cnt, ncclFloat64, S->lo, S->comm, st));
        if (S->lo >= 0) NCCLCHK(g.Recv(B[2], cnt, ncclFloat64, S->lo, S->comm, st));
        if (S->hi >= 0) NCCLCHK(g.Recv(B[3], cnt, ncclFloat64, S->hi, S->comm, st));
        NCCLCHK(g.GroupEnd());
        hipLaunchKernelGGL(k_unpack_multi, grd, blk, 0, st, v, pl, rows, S->lo >= 0 ? B[2] : nullptr, S->hi >= 0 ? B[3] : nullptr);
        HIPCHK(hipGetLastError());
        S->exchanges++;
    }
    return 0;
}

int allreduce_hook(void *user, double *value)
{
    Strip *S = (Strip *)user;
    HIPCHK(hipMemcpyAsync(S->dscalar, value, sizeof(double), hipMemcpyHostToDevice, S->st));
    NCCLCHK(g.AllReduce(S->dscalar, S->dscalar, 1, ncclFloat64, ncclMax, S->comm, S->st));
    HIPCHK(hipMemcpyAsync(value, S->dscalar, sizeof(double), hipMemcpyDeviceToHost, S->st));
    HIPCHK(hipStreamSynchronize(S->st));
    return 0;
}
// the same on device values, on the kernels' stream: n values in place, op 0 MAX / 1 SUM (suhmo_reduce_finish reads them back
// through the pinned slot: no copy, no stream synchronisation)
int reduce_dev_hook(void *user, double *dev_values, int n, int op, suhmo_stream_t s)
{
    Strip *S = (Strip *)user;
    NCCLCHK(g.AllReduce(dev_values, dev_values, (size_t)n, ncclFloat64, op ? ncclSum : ncclMax, S->comm, (hipStream_t)s));
    return 0;
}
}  // namespace

// all-gather of the coarse cells a hierarchy's level 1 reads of a level 0 cut into strips (suhmo_hier.hip), on the kernels' stream
int suhmo_rccl_allgather_hook(void *user, const double *send, long count, double *recv, suhmo_stream_t s)
{
    Strip *S = (Strip *)user;
    NCCLCHK(g.AllGather(send, recv, (size_t)count, ncclFloat64, S->comm, (hipStream_t)s));
    S->exchanges++;
    return 0;
}

// RCCL must sit on the SAME HIP runtime instance as this library (streams and buffers cross the call): a
// process can hold two (ROCm's and the copy PyTorch ships, whichever was mapped first serves us).  With no
// explicit path, take the librccl that lives next to the libamdhip64 this library is bound to.
extern "C" int suhmo_rccl_load(const char *path)
{
    if (g.dl) return 0;
    void *dl = nullptr;
    std::string tried;
    if (path && path[0]) { dl = dlopen(path, RTLD_NOW | RTLD_GLOBAL); tried = path; }
    else {
        Dl_info info;
        std::string dir;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            dir = info.dli_fname;
            size_t k = dir.rfind('/');
            dir = k == std::string::npos ? std::string() : dir.substr(0, k + 1);
        }
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !dl; pass++)
            for (const char *nm : names) {
                std::string cand = (pass == 0 ? dir : std::string()) + nm;
                if (pass == 0 && dir.empty()) continue;
                dl = dlopen(cand.c_str(), RTLD_NOW | RTLD_GLOBAL);
                tried += cand + " ";
                if (dl) break;
            }
    }
    if (!dl) { suhmo_set_error("dlopen(%s) failed: %s", tried.c_str(), dlerror()); return -7; }
#define SYM(field, name) do { g.field = (decltype(g.field))dlsym(dl, name); \
    if (!g.field) { suhmo_set_error("librccl lacks %s", name); return -7; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
    g.CommCount = (decltype(g.CommCount))dlsym(dl, "ncclCommCount");      // (optional: reporting only)
#undef SYM
    g.dl = dl;
    return 0;
}

extern "C" int suhmo_rccl_unique_id(void *id128)
{
    ARG(id128);
    if (!g.dl) { suhmo_set_error("suhmo_rccl_load first"); return -7; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    NCCLCHK(g.GetUniqueId((ncclUniqueId *)id128));
    return 0;
}

extern "C" int suhmo_level_detach_rccl(suhmo_level_t *L)
{
    ARG(L);
    Strip *S = (Strip *)L->rccl;
    if (!S) return 0;
    (void)hipSetDevice(L->device);
    (void)hipDeviceSynchronize();
    for (int d = 0; d < SUHMO_MAXDEPTH; d++) for (int k = 0; k < 4; k++) if (S->buf[d][k]) (void)hipFree(S->buf[d][k]);
    if (S->dscalar) (void)hipFree(S->dscalar);
    if (S->comm && g.CommDestroy) (void)g.CommDestroy(S->comm);
    if (L->ag_user == S) { L->ag = nullptr; L->ag_user = nullptr; suhmo_agg_release(L); }
    if (L->user == S) { L->ex = nullptr; L->ar = nullptr; L->ard = nullptr; L->user = nullptr; L->ex_begin = nullptr; L->ex_end = nullptr; }
    delete S;
    L->rccl = nullptr;
    return 0;
}

// collective over the `world` ranks that own the strips of this level (rank r owns rows [j0, j0 + ny),
// strips ordered by rank); periodic_y: the first and last strip are neighbours
extern "C" int suhmo_level_attach_rccl(suhmo_level_t *L, const void *id128, int rank, int world, int periodic_y, suhmo_stream_t s)
{
    ARG(L && id128); ARG(world >= 1 && rank >= 0 && rank < world);
    if (!g.dl) { suhmo_set_error("suhmo_rccl_load first"); return -7; }
    if (L->rccl) { suhmo_set_error("level already attached"); return -1; }
    HIPCHK(hipSetDevice(L->device));
    Strip *S = new Strip;
    S->rank = rank; S->world = world; S->st = (hipStream_t)s;
    S->lo = rank > 0 ? rank - 1 : (periodic_y ? world - 1 : -1);
    S->hi = rank < world - 1 ? rank + 1 : (periodic_y ? 0 : -1);
    L->rccl = S;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g.CommInitRank(&S->comm, world, id, rank);
    if (r != ncclSuccess) { suhmo_set_error("ncclCommInitRank -> %s", g.GetErrorString(r)); S->comm = nullptr; suhmo_level_detach_rccl(L); return -7; }
    if (const char *e = getenv("SUHMO_RCCL_DIRECT")) S->direct = atoi(e) != 0;
    for (int d = 0; d < L->ndepth && !S->direct; d++) {
        const DV &v = L->d[d].v;
        int rows = v.gy < v.ny ? v.gy : v.ny;
        size_t cap = (size_t)MAXF * rows * (v.nx + 1) * sizeof(double);
        for (int k = 0; k < 4; k++)
            if (hipMalloc(&S->buf[d][k], cap) != hipSuccess) { suhmo_set_error("halo buffer allocation failed"); suhmo_level_detach_rccl(L); return -2; }
    }
    if (hipMalloc(&S->dscalar, sizeof(double)) != hipSuccess) { suhmo_level_detach_rccl(L); return -2; }
    // Every rank derives the number, order and size of its halo messages from its own strip (rows, halo depth, MG depths,
    // kernel selection): neighbours that disagree would post different ncclSend / ncclRecv sequences and hang or corrupt.
    // Agree now: MAX all-reduce of (x, -x) of every deciding quantity; max(x) != -max(-x) = a rank differs.
    {
        const DV &v0 = L->d[0].v;
        const double desc[] = {(double)v0.nx, (double)v0.ny, (double)v0.gy, (double)L->ndepth, (double)L->gsrb_variant, (double)L->fused_min_cells,
                               (double)L->tile_max_cells, (double)L->gsrb_tile, (double)L->tile_t, (double)L->tile_s, (double)L->fused_nt,
                               (double)L->fused_hc, (double)L->fused_restrict, (double)L->tile_strips, (double)L->tile_chunks,
                               (double)L->fas_rhs_in_relax, (double)L->strips_rhs_local, (double)L->bcoef_fused, (double)v0.nxg, (double)v0.nyg, (double)L->tile_restrict,
                               (double)L->overlap_halo, (double)L->agg_min_cells, (double)L->fas_rhs_fused};
        const int K = (int)(sizeof(desc) / sizeof(desc[0]));
        static const char *names[] = {"nx", "ny (rows per strip: the level must be cut into EQUAL strips)", "halo_rows", "multigrid depths", "SUHMO_GSRB_VARIANT",
                                      "SUHMO_FUSED_MIN_CELLS", "SUHMO_TILE_MAX_CELLS", "SUHMO_GSRB_TILE", "SUHMO_TILE_T", "SUHMO_TILE_S", "SUHMO_FUSED_NT",
                                      "SUHMO_FUSED_HC", "SUHMO_FUSED_RESTRICT", "SUHMO_TILE_STRIPS", "SUHMO_TILE_CHUNKS", "SUHMO_FAS_RHS_IN_RELAX",
                                      "SUHMO_STRIPS_RHS_LOCAL", "SUHMO_BCOEF_FUSED", "nx_global", "ny_global", "SUHMO_TILE_RESTRICT", "SUHMO_OVERLAP_HALO",
                                          "SUHMO_AGG_MIN_CELLS", "SUHMO_FAS_RHS_FUSED"};
        double h[2 * 32], *dbuf = nullptr;
        for (int k = 0; k < K; k++) { h[2 * k] = desc[k]; h[2 * k + 1] = -desc[k]; }
        bool ok = hipMalloc(&dbuf, 2 * K * sizeof(double)) == hipSuccess
                  && hipMemcpyAsync(dbuf, h, 2 * K * sizeof(double), hipMemcpyHostToDevice, S->st) == hipSuccess
                  && g.AllReduce(dbuf, dbuf, 2 * K, ncclFloat64, ncclMax, S->comm, S->st) == ncclSuccess
                  && hipMemcpyAsync(h, dbuf, 2 * K * sizeof(double), hipMemcpyDeviceToHost, S->st) == hipSuccess
                  && hipStreamSynchronize(S->st) == hipSuccess;
        if (dbuf) (void)hipFree(dbuf);
        if (!ok) { suhmo_set_error("attach: the consistency all-reduce failed"); suhmo_level_detach_rccl(L); return -7; }
        for (int k = 0; k < K; k++)
            if (h[2 * k] != -h[2 * k + 1]) {
                suhmo_set_error("attach: the ranks of this communicator disagree on %s (here %g, over the ranks %g .. %g)", names[k], desc[k], -h[2 * k + 1], h[2 * k]);
                suhmo_level_detach_rccl(L);
                return -8;
            }
    }
    L->ex = exchange_hook; L->ar = allreduce_hook; L->ard = reduce_dev_hook; L->user = S; L->ex_begin = begin_hook; L->ex_end = end_hook;
    // coarse depths below agg_min_cells cells per strip: agglomerated (suhmo_agg.hip) -- when the communicator spans the whole level
    // (a strip that is its own periodic neighbour, the one-GPU test harness, declares a level twice its size: not then)
    if (world * L->d[0].v.ny == L->d[0].v.nyg) { L->ag = suhmo_rccl_allgather_hook; L->ag_user = S; return suhmo_agg_setup(L); }
    return 0;
}

// the number of ranks the level's communicator reports (ncclCommCount): what a run over N GPUs prints so that N can be verified; -1: not attached
extern "C" int suhmo_level_rccl_comm_count(const suhmo_level_t *L)
{
    if (!L || !L->rccl || !g.CommCount) return -1;
    int n = -1;
    if (g.CommCount(((Strip *)L->rccl)->comm, &n) != ncclSuccess) return -1;
    return n;
}
extern "C" long suhmo_level_rccl_exchanges(const suhmo_level_t *L)
{
    return (L && L->rccl) ? ((Strip *)L->rccl)->exchanges : -1;
}
