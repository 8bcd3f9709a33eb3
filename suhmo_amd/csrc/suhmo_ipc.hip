// suhmo_ipc.hip -- peer-direct strip-halo transport: the exchange kernel of a rank stores its edge rows STRAIGHT INTO THE NEIGHBOUR'S
// receive slots (device memory of the neighbouring GPU mapped with hipIpcOpenMemHandle, peer stores over xGMI) and publishes a sequence
// number there; the same launch then polls the neighbour's number in its own memory, copies its slots into its halo rows and acknowledges.
// Replaces, for the halo rows, what the reference does with MPI point-to-point inside LevelData::exchange
// (src/VCAMRNonLinearPoissonOp.cpp:692, 912-913) and what suhmo_rccl.hip does with pack -> ncclGroup{Send, Recv} -> unpack: no
// communication kernel, none of RCCL's own memsets and copies, ONE launch per exchange instead of three and a dozen small operations.
// Reductions and all-gathers stay with whatever transport the level is attached to (RCCL natively).
//
// Per rank ONE arena (fine-grained device memory, one IPC handle): per multigrid depth two receive slots (double buffering) for each
// side, and a block of flag words, one PER WORKGROUP of the exchange launch (k_ipc_exchange, below):
//   arrive[side][b]   written by workgroup b of the neighbour on that side: number of its last message whose pieces b are complete in my slot
//   ack[side][b]      written by workgroup b of the neighbour on that side: number of MY last message whose pieces b it has copied out of ITS
//                     slot (frees them)
// All polling is local; everything remote is a store.  Message n of a depth uses slot n & 1 and may be stored once the neighbour has
// acknowledged message n - 2.  Every wait is bounded (about 3 s): a neighbour that never answers raises an error word the next
// exchange reports, instead of hanging the queue.  Launches that wait are at most 512 workgroups of 256 threads, so the neighbour's kernels
// find room even when two ranks share one GPU (the test harness).  One process per GPU is the deployment; ranks that are THREADS of one process work as
// long as every rank's stream has a hardware queue of its own (HIP multiplexes streams onto a few: with more than two thread ranks a waiting
// kernel can sit in front of the kernel it waits for -- the test suite runs ranks that share a GPU as processes).
#include "suhmo_common.h"
#include <unistd.h>

namespace {
constexpr int MAXF = 8;
constexpr int GMAX = 512;                                   // workgroups of an exchange launch at most: one arrive and one ack word per workgroup and side
struct IpcFlags { unsigned long long arrive[2][GMAX], ack[2][GMAX]; };
constexpr int MAXSEG = 24;
struct Seg { double *p; int w, P, rows, pack_lo, pack_hi, unpack_lo, unpack_hi, piece0, nch; long off; };   // p: canvas address of (i = 0, j = 0); piece0: first piece of the list; nch: pieces per row
struct SegList { Seg e[MAXSEG]; int n, npiece; };                                                    // npiece: pieces (a row x a chunk of CHUNK doubles) of all segments, per side
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
    int nblk[SUHMO_MAXDEPTH + 1][2] = {};                                      // workgroups of the last message on each slot of a channel
    long exchanges = 0;
    // what suhmo_level_attach_ipc replaced
    suhmo_exchange_fn prev_ex = nullptr; int (*prev_begin)(void *) = nullptr; int (*prev_end)(void *, suhmo_level *, suhmo_stream_t) = nullptr; bool hooked = false;
    int max_blocks = GMAX;                                                     // workgroups per exchange launch at most (env SUHMO_IPC_BLOCKS: A/B runs)
};

// err[0]: a wait ran out; err[1..4]: which one (1 / 2 acknowledgement from lo / hi, 3 / 4 arrival from lo / hi), the number waited for, the number
// seen, the depth
__device__ __forceinline__ void wait_ge(const unsigned long long *p, unsigned long long v, unsigned long long *err, int what, int depth)
{
    const long long t0 = wall_clock64();
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= v) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;       // a wait has run out before: what is queued behind it drains at once
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
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
// exchanges between ex_begin and ex_end (the face coefficients of all depths) are ONE list on the batch channel: one launch.
// the segment table from the kernel arguments into LDS (statically indexed copies: a dynamically indexed by-value argument lands in scratch)
__device__ __forceinline__ void seg_table(const SegList &sl, Seg *segs)
{
#pragma unroll
    for (int k = 0; k < MAXSEG; k++) if ((int)threadIdx.x == k && k < sl.n) segs[k] = sl.e[k];
    __syncthreads();
}
__device__ __forceinline__ const Seg &seg_of(const Seg *segs, int n, int u, int &r, int &c)
{
    int k = 0;
    while (k + 1 < n && u >= segs[k + 1].piece0) k++;
    const int v = u - segs[k].piece0;
    r = v / segs[k].nch; c = v - r * segs[k].nch;
    return segs[k];
}
// halo data crosses GPUs (or, in the test harness, the XCDs of one): stores that go through to memory, loads that do not stop at a cache
__device__ __forceinline__ void st_sys(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_sys(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// ONE launch per message and NO grid-wide step in it.  Workgroup b owns the pieces u = b, b + G, ... of the message (piece = a row of a
// segment x a chunk of CHUNK doubles) on BOTH sides, and so does workgroup b of either neighbour (the messages of two neighbours are laid
// out alike and launched with the same G): it stores its pieces of this rank's edge rows into the neighbours' slots, waits for its own
// stores, and publishes the message number in the neighbours' arrive[.][b]; then it polls ITS OWN arrive[.][b], copies the same pieces
// of its own slots into the halo rows and acknowledges in the neighbours' ack[.][b] (which is what frees those pieces of the slot two
// messages later).  A workgroup's wait depends on the neighbour's workgroup b having stored, and that one waits for nothing but an
// acknowledgement given two messages earlier: no cycle, whatever the order workgroups are scheduled in.  The serial chain of a message is
// load, store, flag, poll, load, store -- six memory latencies -- instead of a vote of all workgroups in each direction.
constexpr int CHUNK = 1280, PER = CHUNK / 256;
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__global__ __launch_bounds__(256) void k_ipc_exchange(SegList sl, unsigned long long seq, double *__restrict__ to_lo, double *__restrict__ to_hi,
                                                      const double *__restrict__ from_lo, const double *__restrict__ from_hi,
                                                      IpcFlags *mine, IpcFlags *flo, IpcFlags *fhi, unsigned long long *err, int chan, int gprev)
{
    __shared__ Seg segs[MAXSEG];
    seg_table(sl, segs);
    const int per_side = sl.npiece, b = blockIdx.x;
    bool first = true;
    for (int u = b; u < per_side; u += gridDim.x) {
        int r, c;
        const Seg &q = seg_of(segs, sl.n, u, r, c);
        const int i0 = c * CHUNK + threadIdx.x;
        double v[2][PER];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (!(side ? to_hi : to_lo)) continue;
            const double *__restrict__ src = q.p + (long)((side ? q.pack_hi : q.pack_lo) + r) * q.P;
#pragma unroll
            for (int k = 0; k < PER; k++) v[side][k] = i0 + 256 * k < q.w ? src[i0 + 256 * k] : 0.0;
        }
        if (first) {                                         // the slot is free once the neighbour has copied message seq - 2 out of it: every one of the gprev
            if (seq > 2)                                     // workgroups of THAT message (its fields, and so its pieces, may have been others)
                for (int t = threadIdx.x; t < gprev; t += 256) {         // (both words on their way before either is looked at; normally both are there)
                    const unsigned long long a0 = to_lo ? __hip_atomic_load(&mine->ack[0][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
                    const unsigned long long a1 = to_hi ? __hip_atomic_load(&mine->ack[1][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
                    if (a0 < seq - 2) wait_ge(&mine->ack[0][t], seq - 2, err, 1, chan);
                    if (a1 < seq - 2) wait_ge(&mine->ack[1][t], seq - 2, err, 2, chan);
                }
            __syncthreads();
            first = false;
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            double *slot = side ? to_hi : to_lo;
            if (!slot) continue;
            slot += q.off + (long)r * q.w;
#pragma unroll
            for (int k = 0; k < PER; k++) if (i0 + 256 * k < q.w) st_sys(slot + i0 + 256 * k, v[side][k]);
        }
    }
    wait_own_stores();                                       // every wave: its stores (write-through, st_sys) have been acknowledged by the memory they went to ...
    __syncthreads();                                         // ... before lane 0 says so.  No acquire / release operations anywhere in this kernel: each one writes back or
                                                             // invalidates a whole L2, and 512 workgroups doing that cost 100 us; what crosses ranks bypasses the caches instead
    if (threadIdx.x == 0) {
        if (to_lo) __hip_atomic_store(&flo->arrive[1][b], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // I am my lo neighbour's hi side
        if (to_hi) __hip_atomic_store(&fhi->arrive[0][b], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (from_lo) wait_ge(&mine->arrive[0][b], seq, err, 3, chan);
        if (from_hi) wait_ge(&mine->arrive[1][b], seq, err, 4, chan);
    }
    __syncthreads();
    for (int u = b; u < per_side; u += gridDim.x) {
        int r, c;
        const Seg &q = seg_of(segs, sl.n, u, r, c);
        const int i0 = c * CHUNK + threadIdx.x;
        double v[2][PER];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const double *slot = side ? from_hi : from_lo;
            if (!slot) continue;
            slot += q.off + (long)r * q.w;
#pragma unroll
            for (int k = 0; k < PER; k++) v[side][k] = i0 + 256 * k < q.w ? ld_sys(slot + i0 + 256 * k) : 0.0;
        }
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (!(side ? from_hi : from_lo)) continue;
            double *__restrict__ dst = q.p + (long)((side ? q.unpack_hi : q.unpack_lo) + r) * q.P;
#pragma unroll
            for (int k = 0; k < PER; k++) if (i0 + 256 * k < q.w) dst[i0 + 256 * k] = v[side][k];
        }
    }
    __syncthreads();                                         // (every load of the slots has returned: its value went into a store)
    if (threadIdx.x == 0) {
        if (from_lo) __hip_atomic_store(&flo->ack[1][b], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (from_hi) __hip_atomic_store(&fhi->ack[0][b], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one message: the segments of `sl` on channel `chan` (a depth, or the batch channel = ndepth)
int ipc_send(IpcStrip *S, SegList &sl, int chan, hipStream_t st)
{
    long off = 0;
    int piece0 = 0;
    for (int k = 0; k < sl.n; k++) {
        Seg &e = sl.e[k];
        e.off = off; e.piece0 = piece0; e.nch = (e.w + CHUNK - 1) / CHUNK;
        off += (long)e.rows * e.w; piece0 += e.rows * e.nch;
    }
    sl.npiece = piece0;
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
    const int nblk = std::max(1, std::min(S->max_blocks, sl.npiece));
    const int gprev = S->nblk[chan][q];                      // workgroups of message seq - 2 (same slot)
    S->nblk[chan][q] = nblk;
    hipLaunchKernelGGL(k_ipc_exchange, dim3(nblk), dim3(256), 0, st, sl, seq, to_lo, to_hi, from_lo, from_hi, mine, flo, fhi, S->herr_dev, chan, gprev);
    HIPCHK(hipGetLastError());
    S->exchanges++;
    return 0;
}
int ipc_check(IpcStrip *S)
{
    if (!*S->herr) return 0;
    static const char *what[] = {"?", "the acknowledgement of the lower neighbour", "the acknowledgement of the upper neighbour",
        "the message of the lower neighbour", "the message of the upper neighbour"};
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
    if (const char *e = getenv("SUHMO_IPC_BLOCKS")) S->max_blocks = std::max(1, std::min(GMAX, atoi(e)));
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
    for (int d = 0; d <= L->ndepth; d++) { S->flags[d] = off; off += (sizeof(IpcFlags) + 255) & ~(size_t)255; }
    S->bytes = off;
    // fine-grained: stores of a peer and the flag words behind them must be visible to a running kernel of the owner
    // (and nothing else will do: the exchange kernel orders its stores and flags without acquire / release operations, which holds for memory no
    //  cache keeps a private copy of)
    if (hipExtMallocWithFlags((void **)&S->arena, S->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        delete S;
        suhmo_set_error("ipc transport: no fine-grained device memory for the arena (hipExtMallocWithFlags)");
        return -2;
    }
    HIPCHK(hipMemset(S->arena, 0, S->bytes));
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
    if (!S->hooked) { S->prev_ex = L->ex; S->prev_begin = L->ex_begin; S->prev_end = L->ex_end; S->hooked = true; }
    L->ex = ipc_exchange_hook; L->ex_begin = nullptr; L->ex_end = nullptr;
    return 0;
}
// back to the transport the level had before suhmo_level_attach_ipc (a host that probes the peer-direct path and finds it wanting on some rank:
// suhmo_amd.multigpu.attach); the arena is unmapped and freed.  Not while a gap-height operator or another handle shares the attachment.
extern "C" int suhmo_level_detach_ipc(suhmo_level_t *L)
{
    ARG(L);
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S) return 0;
    if (S->hooked) { L->ex = S->prev_ex; L->ex_begin = S->prev_begin; L->ex_end = S->prev_end; }
    ipc_release(L);
    return 0;
}
extern "C" long suhmo_level_ipc_exchanges(const suhmo_level_t *L) { return (L && L->ipc) ? ((IpcStrip *)L->ipc)->exchanges : -1; }
