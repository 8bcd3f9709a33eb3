// suhmo_ipc.hip -- peer-direct strip-halo transport: the pack kernel of a rank stores its edge rows STRAIGHT INTO THE NEIGHBOUR'S
// receive slots (device memory of the neighbouring GPU mapped with hipIpcOpenMemHandle, peer stores over xGMI) and publishes a sequence
// number there; the neighbour's unpack kernel polls that number in its own memory, copies the slot into its halo rows and acknowledges.
// Replaces, for the halo rows, what the reference does with MPI point-to-point inside LevelData::exchange
// (src/VCAMRNonLinearPoissonOp.cpp:692, 912-913) and what suhmo_rccl.hip does with pack -> ncclGroup{Send, Recv} -> unpack: no
// communication kernel, none of RCCL's own memsets and copies, two launches per exchange instead of three and a dozen small operations.
// Reductions and all-gathers stay with whatever transport the level is attached to (RCCL natively).
//
// Per rank ONE arena (fine-grained device memory, one IPC handle): per multigrid depth two receive slots (double buffering) for each
// side, and a block of flag words:
//   arrive[side]   written by the neighbour on that side: number of its last message that is complete in my slot
//   ack[side]      written by the neighbour on that side: number of MY last message it has copied out of ITS slot (frees the slot)
// All polling is local; everything remote is a store.  Message n of a depth uses slot n & 1 and may be packed once the neighbour has
// acknowledged message n - 2.  Every wait is bounded (about 3 s): a neighbour that never answers raises an error word the next
// exchange reports, instead of hanging the queue.  Kernels that wait are at most 128 workgroups, so the neighbour's kernels find room
// even when two ranks share one GPU (the test harness).  One process per GPU is the deployment; ranks that are THREADS of one process work as
// long as every rank's stream has a hardware queue of its own (HIP multiplexes streams onto a few: with more than two thread ranks a waiting
// kernel can sit in front of the kernel it waits for -- the test suite keeps thread ranks at two and runs more ranks as processes).
#include "suhmo_common.h"
#include <unistd.h>

namespace {
constexpr int MAXF = 8;
struct IpcFlags { unsigned long long arrive[2], ack[2]; unsigned int count[2]; unsigned int pad[2]; };
constexpr int MAXSEG = 24;
struct Seg { double *p; int w, P, rows, pack_lo, pack_hi, unpack_lo, unpack_hi, row0; long off; };   // p: canvas address of (i = 0, j = 0); row0: first row of the list
struct SegList { Seg e[MAXSEG]; int n, nrow, nch; };                                                 // nrow: rows of all segments; nch: 1024-double chunks of the widest row
struct IpcBlob { hipIpcMemHandle_t handle; long long pid; unsigned long long ptr; unsigned long long bytes; };    // 64 + 24 bytes <= 128
struct IpcStrip {
    int rank = 0, world = 1, lo = -1, hi = -1, ndepth = 0;
    char *arena = nullptr; size_t bytes = 0;
    char *remote[2] = {nullptr, nullptr}; bool mapped[2] = {false, false};     // the arenas of the lo / hi neighbour
    // channels: one per depth, and the batch channel (index ndepth) for the exchanges between ex_begin and ex_end
    size_t slot[SUHMO_MAXDEPTH + 1][2][2] = {}, slot_cap[SUHMO_MAXDEPTH + 1] = {}, flags[SUHMO_MAXDEPTH + 1] = {};   // offsets: [channel][side][slot & 1]; slot_cap in doubles
    unsigned long long seq[SUHMO_MAXDEPTH + 1] = {};
    bool batching = false; SegList batch;
    unsigned long long *herr = nullptr, *herr_dev = nullptr;                   // pinned: a kernel whose wait ran out says so here
    unsigned int *counters = nullptr;                                          // last-workgroup elections (ordinary device memory: two words per channel)
    long exchanges = 0;
    int max_blocks = 192;                                                      // workgroups per pack / unpack launch at most (env SUHMO_IPC_BLOCKS: A/B runs)
};

// err[0]: a wait ran out; err[1..4]: which one (1 / 2 acknowledgement from lo / hi, 3 / 4 arrival from lo / hi), the number waited for, the number
// seen, the depth
__device__ __forceinline__ void wait_ge(const unsigned long long *p, unsigned long long v, unsigned long long *err, int what, int depth)
{
    const long long t0 = wall_clock64();
    if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= v) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;       // a wait has run out before: what is queued behind it drains at once
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 300000000LL) {                                           // ~3 s at 100 MHz
            if (!__hip_atomic_exchange(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                err[1] = (unsigned long long)what; err[2] = v; err[3] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); err[4] = (unsigned long long)depth;
            }
            return;
        }
    }
}
// A message = a list of segments (one field of one depth each: `rows` canvas rows of w = nx + 1 doubles -- x-face rows whole -- next to the lo
// and hi side), laid out one after the other in a slot.  A single exchange is the list of its fields, on the channel of its depth; the
// exchanges between ex_begin and ex_end (the face coefficients of all depths) are ONE list on the batch channel: one pack and one unpack launch.
// the segment table from the kernel arguments into LDS (statically indexed copies: a dynamically indexed by-value argument lands in scratch)
__device__ __forceinline__ void seg_table(const SegList &sl, Seg *segs)
{
#pragma unroll
    for (int k = 0; k < MAXSEG; k++) if ((int)threadIdx.x == k && k < sl.n) segs[k] = sl.e[k];
    __syncthreads();
}
__device__ __forceinline__ const Seg &seg_of(const Seg *segs, int n, int row, int &r)
{
    int k = 0;
    while (k + 1 < n && row >= segs[k + 1].row0) k++;
    r = row - segs[k].row0;
    return segs[k];
}
// halo data crosses GPUs (or, in the test harness, the XCDs of one): stores that go through to memory, loads that do not stop at a cache
__device__ __forceinline__ void st_sys(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_sys(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// workgroup = (side, row, chunk of 1024 doubles): four loads in flight per thread, then the stores -- into the neighbour's slot when packing
__global__ __launch_bounds__(256) void k_ipc_pack(SegList sl, unsigned long long seq, double *__restrict__ to_lo, double *__restrict__ to_hi,
                                                  IpcFlags *mine, IpcFlags *flo, IpcFlags *fhi, unsigned long long *err, int chan, unsigned int *count)
{
    __shared__ int last;
    __shared__ Seg segs[MAXSEG];
    seg_table(sl, segs);
    if (threadIdx.x == 0 && seq > 2) {                       // the slot is free once the neighbour has copied message seq - 2 out of it
        if (to_lo) wait_ge(&mine->ack[0], seq - 2, err, 1, chan);
        if (to_hi) wait_ge(&mine->ack[1], seq - 2, err, 2, chan);
    }
    __syncthreads();
    const int per_side = sl.nrow * sl.nch;
    for (int t = blockIdx.x; t < 2 * per_side; t += gridDim.x) {
        const int side = t / per_side, u = t - side * per_side, row = u / sl.nch, c = u - row * sl.nch;
        double *b = side ? to_hi : to_lo;
        if (!b) continue;
        int r;
        const Seg &q = seg_of(segs, sl.n, row, r);
        const double *__restrict__ src = q.p + (long)((side ? q.pack_hi : q.pack_lo) + r) * q.P;
        double *dst = b + q.off + (long)r * q.w;
        const int i0 = c * 1024 + threadIdx.x;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        if (i0 < q.w) v0 = src[i0];
        if (i0 + 256 < q.w) v1 = src[i0 + 256];
        if (i0 + 512 < q.w) v2 = src[i0 + 512];
        if (i0 + 768 < q.w) v3 = src[i0 + 768];
        if (i0 < q.w) st_sys(dst + i0, v0);
        if (i0 + 256 < q.w) st_sys(dst + i0 + 256, v1);
        if (i0 + 512 < q.w) st_sys(dst + i0 + 512, v2);
        if (i0 + 768 < q.w) st_sys(dst + i0 + 768, v3);
    }
    __syncthreads();                                         // (the workgroup's stores are ordered before lane 0's fence by the barrier; the data
    if (threadIdx.x == 0) {                                  //  stores themselves go through to memory: st_sys)
        __threadfence_system();
        last = atomicAdd(count, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        *count = 0;
        __threadfence_system();
        if (to_lo) __hip_atomic_store(&flo->arrive[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);     // I am my lo neighbour's hi side
        if (to_hi) __hip_atomic_store(&fhi->arrive[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void k_ipc_unpack(SegList sl, unsigned long long seq, const double *__restrict__ from_lo, const double *__restrict__ from_hi,
                                                    IpcFlags *mine, IpcFlags *flo, IpcFlags *fhi, unsigned long long *err, int chan, unsigned int *count)
{
    __shared__ int last;
    __shared__ Seg segs[MAXSEG];
    seg_table(sl, segs);
    if (threadIdx.x == 0) {
        if (from_lo) wait_ge(&mine->arrive[0], seq, err, 3, chan);
        if (from_hi) wait_ge(&mine->arrive[1], seq, err, 4, chan);
    }
    __syncthreads();
    const int per_side = sl.nrow * sl.nch;
    for (int t = blockIdx.x; t < 2 * per_side; t += gridDim.x) {
        const int side = t / per_side, u = t - side * per_side, row = u / sl.nch, c = u - row * sl.nch;
        const double *b = side ? from_hi : from_lo;
        if (!b) continue;
        int r;
        const Seg &q = seg_of(segs, sl.n, row, r);
        double *__restrict__ dst = q.p + (long)((side ? q.unpack_hi : q.unpack_lo) + r) * q.P;
        const double *src = b + q.off + (long)r * q.w;
        const int i0 = c * 1024 + threadIdx.x;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        if (i0 < q.w) v0 = ld_sys(src + i0);
        if (i0 + 256 < q.w) v1 = ld_sys(src + i0 + 256);
        if (i0 + 512 < q.w) v2 = ld_sys(src + i0 + 512);
        if (i0 + 768 < q.w) v3 = ld_sys(src + i0 + 768);
        if (i0 < q.w) dst[i0] = v0;
        if (i0 + 256 < q.w) dst[i0 + 256] = v1;
        if (i0 + 512 < q.w) dst[i0 + 512] = v2;
        if (i0 + 768 < q.w) dst[i0 + 768] = v3;
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(count, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        *count = 0;
        if (from_lo) __hip_atomic_store(&flo->ack[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (from_hi) __hip_atomic_store(&fhi->ack[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one message: the segments of `sl` on channel `chan` (a depth, or the batch channel = ndepth)
int ipc_send(IpcStrip *S, SegList &sl, int chan, hipStream_t st)
{
    long off = 0;
    int row0 = 0, wmax = 1;
    for (int k = 0; k < sl.n; k++) { sl.e[k].off = off; sl.e[k].row0 = row0; off += (long)sl.e[k].rows * sl.e[k].w; row0 += sl.e[k].rows; wmax = std::max(wmax, sl.e[k].w); }
    sl.nrow = row0; sl.nch = (wmax + 1023) / 1024;
    if ((size_t)off > S->slot_cap[chan]) { suhmo_set_error("ipc transport: a message larger than the slots the arena was laid out with"); return -7; }
    const unsigned long long seq = ++S->seq[chan];
    const int q = (int)(seq & 1);
    IpcFlags *mine = (IpcFlags *)(S->arena + S->flags[chan]);
    IpcFlags *flo = S->lo >= 0 ? (IpcFlags *)(S->remote[0] + S->flags[chan]) : nullptr, *fhi = S->hi >= 0 ? (IpcFlags *)(S->remote[1] + S->flags[chan]) : nullptr;
    double *to_lo = S->lo >= 0 ? (double *)(S->remote[0] + S->slot[chan][1][q]) : nullptr;      // my rows next to my lo side are the lo neighbour's hi halo
    double *to_hi = S->hi >= 0 ? (double *)(S->remote[1] + S->slot[chan][0][q]) : nullptr;
    const double *from_lo = S->lo >= 0 ? (const double *)(S->arena + S->slot[chan][0][q]) : nullptr;
    const double *from_hi = S->hi >= 0 ? (const double *)(S->arena + S->slot[chan][1][q]) : nullptr;
    // a bounded number of workgroups: every one ends with a fence and a vote (on a counter in ordinary device memory: votes on the fine-grained
    // arena cost ten times as much), and a neighbour that shares the GPU (test harness) must find room while these wait
    const int nblk = std::max(1, std::min(S->max_blocks, 2 * sl.nrow * sl.nch));
    hipLaunchKernelGGL(k_ipc_pack, dim3(nblk), dim3(256), 0, st, sl, seq, to_lo, to_hi, mine, flo, fhi, S->herr_dev, chan, S->counters + 2 * chan);
    hipLaunchKernelGGL(k_ipc_unpack, dim3(nblk), dim3(256), 0, st, sl, seq, from_lo, from_hi, mine, flo, fhi, S->herr_dev, chan, S->counters + 2 * chan + 1);
    HIPCHK(hipGetLastError());
    S->exchanges++;
    return 0;
}
int ipc_check(IpcStrip *S)
{
    if (!*S->herr) return 0;
    static const char *what[] = {"?", "the acknowledgement of the lower neighbour", "the acknowledgement of the upper neighbour", "the message of the lower neighbour", "the message of the upper neighbour"};
    suhmo_set_error("ipc transport (rank %d of %d): %s did not come within 3 s: waited for number %llu on channel %llu, saw %llu (a neighbour rank stopped, or the ranks disagree on "
                    "the sequence of exchanges); this rank has sent %llu messages on that channel", S->rank, S->world, what[S->herr[1] <= 4 ? S->herr[1] : 0], S->herr[2], S->herr[4], S->herr[3],
                    S->herr[4] <= SUHMO_MAXDEPTH ? S->seq[S->herr[4]] : 0ull);
    return -7;
}
int ipc_exchange_hook(void *user, suhmo_level_t *L, int depth, const int *fields, int nfields, suhmo_stream_t s)
{
    (void)user;
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S) { suhmo_set_error("ipc transport: the level is not attached"); return -7; }
    int rc = ipc_check(S); if (rc) return rc;
    if (depth >= S->ndepth) { suhmo_set_error("ipc transport: depth %d beyond the arena's %d", depth, S->ndepth); return -7; }
    const DV &v = L->d[depth].v;
    const int rows = v.gy < v.ny ? v.gy : v.ny;
    SegList one;
    SegList &sl = S->batching ? S->batch : one;
    if (!S->batching) sl.n = 0;
    for (int q = 0; q < nfields; q++) {
        const int f = fields[q];
        double *p = suhmo_field(L, depth, f);
        if (!p) { suhmo_set_error("field allocation failed"); return -2; }
        if (sl.n == MAXSEG) {                                // (a list is full: send what there is; both neighbours fill theirs alike)
            if ((rc = ipc_send(S, sl, S->batching ? S->ndepth : depth, (hipStream_t)s))) return rc;
            sl.n = 0;
        }
        Seg &e = sl.e[sl.n++];
        e.p = p + cidx(v, 0, 0); e.w = v.nx + 1; e.P = v.P; e.rows = rows;
        e.pack_lo = f == SUHMO_F_BY ? 1 : 0;                 // (as suhmo_rccl.hip: face row 0 of a strip IS face row ny of the lower neighbour)
        e.pack_hi = v.ny - rows;
        e.unpack_lo = -rows;
        e.unpack_hi = f == SUHMO_F_BY ? v.ny + 1 : v.ny;
    }
    if (S->batching) return 0;
    return ipc_send(S, sl, depth, (hipStream_t)s);
}
}  // namespace

// the exchanges between open and close travel as ONE message (suhmo_average_operator_all: the face coefficients of every depth)
int suhmo_ipc_batch(suhmo_level *L, int open, hipStream_t st)
{
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S) return 0;
    if (open) { S->batching = true; S->batch.n = 0; return 0; }
    S->batching = false;
    if (S->batch.n == 0) return 0;
    int rc = ipc_send(S, S->batch, S->ndepth, st);
    S->batch.n = 0;
    return rc;
}

static void ipc_release(suhmo_level *L)
{
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S || !L->ipc_owner) { L->ipc = nullptr; return; }
    (void)hipSetDevice(L->device);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 2; k++) if (S->mapped[k] && S->remote[k] && !(k == 1 && S->remote[1] == S->remote[0] && S->mapped[0])) (void)hipIpcCloseMemHandle(S->remote[k]);
    if (S->arena) (void)hipFree(S->arena);
    if (S->counters) (void)hipFree(S->counters);
    if (S->herr) (void)hipHostFree(S->herr);
    delete S;
    L->ipc = nullptr; L->ipc_owner = 0;
}
void suhmo_ipc_release(suhmo_level *L) { ipc_release(L); }

// Step 1 (every rank): lay out and allocate this rank's arena; `blob` (128 bytes) is what its two neighbours need to reach it.
extern "C" int suhmo_level_ipc_export(suhmo_level_t *L, void *blob128)
{
    ARG(L && blob128);
    HIPCHK(hipSetDevice(L->device));
    if (L->ipc) { suhmo_set_error("ipc transport: already exported / attached"); return -1; }
    IpcStrip *S = new IpcStrip;
    S->ndepth = L->ndepth;
    if (const char *e = getenv("SUHMO_IPC_BLOCKS")) S->max_blocks = std::max(1, atoi(e));
    size_t off = 0;
    for (int d = 0; d < L->ndepth; d++) {
        const DV &v = L->d[d].v;
        const int rows = v.gy < v.ny ? v.gy : v.ny;
        S->slot_cap[d] = (size_t)MAXF * rows * (v.nx + 1);
        for (int side = 0; side < 2; side++) for (int sl = 0; sl < 2; sl++) { S->slot[d][side][sl] = off; off += (S->slot_cap[d] * sizeof(double) + 255) & ~(size_t)255; }
    }
    {   // the batch channel: up to four fields of every depth in one message
        size_t cap = 0;
        for (int d = 0; d < L->ndepth; d++) cap += S->slot_cap[d] / MAXF * 4;
        S->slot_cap[L->ndepth] = cap;
        for (int side = 0; side < 2; side++) for (int sl = 0; sl < 2; sl++) { S->slot[L->ndepth][side][sl] = off; off += (cap * sizeof(double) + 255) & ~(size_t)255; }
    }
    for (int d = 0; d <= L->ndepth; d++) { S->flags[d] = off; off += 256; }
    S->bytes = off;
    // fine-grained: stores of a peer and the flag words behind them must be visible to a running kernel of the owner
    if (hipExtMallocWithFlags((void **)&S->arena, S->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void **)&S->arena, S->bytes) != hipSuccess) { delete S; suhmo_set_error("ipc transport: arena allocation failed"); return -2; }
    }
    HIPCHK(hipMemset(S->arena, 0, S->bytes));
    HIPCHK(hipMalloc((void **)&S->counters, 2 * (SUHMO_MAXDEPTH + 1) * sizeof(unsigned int)));
    HIPCHK(hipMemset(S->counters, 0, 2 * (SUHMO_MAXDEPTH + 1) * sizeof(unsigned int)));
    HIPCHK(hipHostMalloc((void **)&S->herr, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(S->herr, 0, 64);
    HIPCHK(hipHostGetDevicePointer((void **)&S->herr_dev, S->herr, 0));
    HIPCHK(hipDeviceSynchronize());
    IpcBlob b;
    memset(&b, 0, sizeof(b));
    static_assert(sizeof(IpcBlob) <= 128, "blob");
    if (hipIpcGetMemHandle(&b.handle, S->arena) != hipSuccess) { (void)hipGetLastError(); memset(&b.handle, 0, sizeof(b.handle)); }   // (threads of one process need none)
    b.pid = (long long)getpid(); b.ptr = (unsigned long long)(uintptr_t)S->arena; b.bytes = S->bytes;
    memset(blob128, 0, 128);
    memcpy(blob128, &b, sizeof(b));
    L->ipc = S; L->ipc_owner = 1;
    return 0;
}
// Step 2 (every rank, after the blobs have travelled): map the neighbours' arenas and route the level's halo exchanges through them.
// blob_lo / blob_hi: the blobs of rank - 1 / rank + 1 (the periodic neighbours at the ends; NULL where there is none; the rank's own blob
// when it is its own neighbour).  Reductions and all-gathers keep the hooks the level has.
extern "C" int suhmo_level_attach_ipc(suhmo_level_t *L, int rank, int world, int periodic_y, const void *blob_lo, const void *blob_hi)
{
    ARG(L && world >= 1 && rank >= 0 && rank < world);
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S) { suhmo_set_error("ipc transport: suhmo_level_ipc_export first"); return -1; }
    HIPCHK(hipSetDevice(L->device));
    S->rank = rank; S->world = world;
    S->lo = rank > 0 ? rank - 1 : (periodic_y ? world - 1 : -1);
    S->hi = rank < world - 1 ? rank + 1 : (periodic_y ? 0 : -1);
    const void *blobs[2] = {blob_lo, blob_hi};
    for (int k = 0; k < 2; k++) {
        if ((k == 0 ? S->lo : S->hi) < 0) continue;
        if (!blobs[k]) { suhmo_set_error("ipc transport: the blob of the %s neighbour is missing", k ? "upper" : "lower"); return -1; }
        IpcBlob b;
        memcpy(&b, blobs[k], sizeof(b));
        if (b.bytes != S->bytes) { suhmo_set_error("ipc transport: the neighbour's arena is laid out differently (%llu bytes against %zu): unequal strips or options", b.bytes, S->bytes); return -8; }
        if (b.pid == (long long)getpid()) { S->remote[k] = (char *)(uintptr_t)b.ptr; continue; }          // a thread of this process (or this rank itself)
        if (k == 1 && S->mapped[0] && blob_lo && !memcmp(blob_lo, blob_hi, sizeof(IpcBlob))) { S->remote[1] = S->remote[0]; S->mapped[1] = true; continue; }   // two ranks, periodic
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); suhmo_set_error("ipc transport: hipIpcOpenMemHandle -> %s", hipGetErrorString(e)); return -7; }
        S->remote[k] = (char *)p; S->mapped[k] = true;
    }
    L->ex = ipc_exchange_hook; L->ex_begin = nullptr; L->ex_end = nullptr;
    return 0;
}
extern "C" long suhmo_level_ipc_exchanges(const suhmo_level_t *L) { return (L && L->ipc) ? ((IpcStrip *)L->ipc)->exchanges : -1; }
