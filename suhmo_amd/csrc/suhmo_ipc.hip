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
struct PackList { double *p[MAXF]; int pack_lo[MAXF], pack_hi[MAXF], unpack_lo[MAXF], unpack_hi[MAXF]; int n; };
struct IpcBlob { hipIpcMemHandle_t handle; long long pid; unsigned long long ptr; unsigned long long bytes; };    // 64 + 24 bytes <= 128
struct IpcStrip {
    int rank = 0, world = 1, lo = -1, hi = -1, ndepth = 0;
    char *arena = nullptr; size_t bytes = 0;
    char *remote[2] = {nullptr, nullptr}; bool mapped[2] = {false, false};     // the arenas of the lo / hi neighbour
    size_t slot[SUHMO_MAXDEPTH][2][2] = {}, slot_cap[SUHMO_MAXDEPTH] = {}, flags[SUHMO_MAXDEPTH] = {};   // offsets: [depth][side][slot & 1]
    unsigned long long seq[SUHMO_MAXDEPTH] = {};
    unsigned long long *herr = nullptr, *herr_dev = nullptr;                   // pinned: a kernel whose wait ran out says so here
    long exchanges = 0;
};

__device__ __forceinline__ void wait_ge(const unsigned long long *p, unsigned long long v, unsigned long long *err)
{
    const long long t0 = wall_clock64();
    if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= v) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;       // a wait has run out before: what is queued behind it drains at once
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 300000000LL) { __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }   // ~3 s at 100 MHz
    }
}
// message `seq` of a depth: rows travel (nx + 1) wide (x-face rows whole), field after field; to the hi neighbour's lo slot and to the lo
// neighbour's hi slot (NULL: no neighbour on that side)
__global__ __launch_bounds__(256) void k_ipc_pack(DV v, PackList pl, int rows, unsigned long long seq, double *__restrict__ to_lo, double *__restrict__ to_hi,
                                                  IpcFlags *mine, IpcFlags *flo, IpcFlags *fhi, unsigned long long *err)
{
    __shared__ int last;
    if (threadIdx.x == 0 && seq > 2) {                       // the slot is free once the neighbour has copied message seq - 2 out of it
        if (to_lo) wait_ge(&mine->ack[0], seq - 2, err);
        if (to_hi) wait_ge(&mine->ack[1], seq - 2, err);
    }
    __syncthreads();
    const int w = v.nx + 1;
    const long per = (long)rows * w, total = per * pl.n * 2;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int side = (int)(t / (per * pl.n));
        long u = t - (long)side * per * pl.n;
        const int q = (int)(u / per); u -= (long)q * per;
        const int r = (int)(u / w), i = (int)(u - (long)r * w);
        double *b = side ? to_hi : to_lo;
        if (!b) continue;
        const int j = (side ? pl.pack_hi[q] : pl.pack_lo[q]) + r;
        b[(long)q * per + u] = pl.p[q][cidx(v, i, j)];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&mine->count[0], 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        mine->count[0] = 0;
        __threadfence_system();
        if (to_lo) __hip_atomic_store(&flo->arrive[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);     // I am my lo neighbour's hi side
        if (to_hi) __hip_atomic_store(&fhi->arrive[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void k_ipc_unpack(DV v, PackList pl, int rows, unsigned long long seq, const double *__restrict__ from_lo, const double *__restrict__ from_hi,
                                                    IpcFlags *mine, IpcFlags *flo, IpcFlags *fhi, unsigned long long *err)
{
    __shared__ int last;
    if (threadIdx.x == 0) {
        if (from_lo) wait_ge(&mine->arrive[0], seq, err);
        if (from_hi) wait_ge(&mine->arrive[1], seq, err);
    }
    __syncthreads();
    const int w = v.nx + 1;
    const long per = (long)rows * w, total = per * pl.n * 2;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int side = (int)(t / (per * pl.n));
        long u = t - (long)side * per * pl.n;
        const int q = (int)(u / per); u -= (long)q * per;
        const int r = (int)(u / w), i = (int)(u - (long)r * w);
        const double *b = side ? from_hi : from_lo;
        if (!b) continue;
        const int j = (side ? pl.unpack_hi[q] : pl.unpack_lo[q]) + r;
        pl.p[q][cidx(v, i, j)] = __builtin_nontemporal_load(&b[(long)q * per + u]);
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&mine->count[1], 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        mine->count[1] = 0;
        if (from_lo) __hip_atomic_store(&flo->ack[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (from_hi) __hip_atomic_store(&fhi->ack[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int ipc_exchange_hook(void *user, suhmo_level_t *L, int depth, const int *fields, int nfields, suhmo_stream_t s)
{
    (void)user;
    IpcStrip *S = (IpcStrip *)L->ipc;
    if (!S) { suhmo_set_error("ipc transport: the level is not attached"); return -7; }
    if (*S->herr) { suhmo_set_error("ipc transport: a halo message did not arrive within 3 s (a neighbour rank stopped?)"); return -7; }
    hipStream_t st = (hipStream_t)s;
    const DV &v = L->d[depth].v;
    const int rows = v.gy < v.ny ? v.gy : v.ny;
    const size_t n = (size_t)rows * (v.nx + 1);
    if (depth >= S->ndepth || n * (size_t)(nfields < MAXF ? nfields : MAXF) > S->slot_cap[depth]) { suhmo_set_error("ipc transport: a message larger than the slots the arena was laid out with"); return -7; }
    for (int f0 = 0; f0 < nfields; f0 += MAXF) {
        PackList pl;
        pl.n = nfields - f0 < MAXF ? nfields - f0 : MAXF;
        for (int q = 0; q < pl.n; q++) {
            const int f = fields[f0 + q];
            pl.p[q] = suhmo_field(L, depth, f);
            if (!pl.p[q]) { suhmo_set_error("field allocation failed"); return -2; }
            pl.pack_lo[q] = f == SUHMO_F_BY ? 1 : 0;         // (as suhmo_rccl.hip: face row 0 of a strip IS face row ny of the lower neighbour)
            pl.pack_hi[q] = v.ny - rows;
            pl.unpack_lo[q] = -rows;
            pl.unpack_hi[q] = f == SUHMO_F_BY ? v.ny + 1 : v.ny;
        }
        const unsigned long long seq = ++S->seq[depth];
        const int sl = (int)(seq & 1);
        IpcFlags *mine = (IpcFlags *)(S->arena + S->flags[depth]);
        IpcFlags *flo = S->lo >= 0 ? (IpcFlags *)(S->remote[0] + S->flags[depth]) : nullptr, *fhi = S->hi >= 0 ? (IpcFlags *)(S->remote[1] + S->flags[depth]) : nullptr;
        double *to_lo = S->lo >= 0 ? (double *)(S->remote[0] + S->slot[depth][1][sl]) : nullptr;      // my rows next to my lo side are the lo neighbour's hi halo
        double *to_hi = S->hi >= 0 ? (double *)(S->remote[1] + S->slot[depth][0][sl]) : nullptr;
        const double *from_lo = S->lo >= 0 ? (const double *)(S->arena + S->slot[depth][0][sl]) : nullptr;
        const double *from_hi = S->hi >= 0 ? (const double *)(S->arena + S->slot[depth][1][sl]) : nullptr;
        const long total = 2L * (long)n * pl.n;
        const int nblk = (int)std::max(1L, std::min(128L, (total + 1023) / 1024));
        hipLaunchKernelGGL(k_ipc_pack, dim3(nblk), dim3(256), 0, st, v, pl, rows, seq, to_lo, to_hi, mine, flo, fhi, S->herr_dev);
        hipLaunchKernelGGL(k_ipc_unpack, dim3(nblk), dim3(256), 0, st, v, pl, rows, seq, from_lo, from_hi, mine, flo, fhi, S->herr_dev);
        HIPCHK(hipGetLastError());
        S->exchanges++;
    }
    return 0;
}
}  // namespace

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
    size_t off = 0;
    for (int d = 0; d < L->ndepth; d++) {
        const DV &v = L->d[d].v;
        const int rows = v.gy < v.ny ? v.gy : v.ny;
        S->slot_cap[d] = (size_t)MAXF * rows * (v.nx + 1);
        for (int side = 0; side < 2; side++) for (int sl = 0; sl < 2; sl++) { S->slot[d][side][sl] = off; off += (S->slot_cap[d] * sizeof(double) + 255) & ~(size_t)255; }
    }
    for (int d = 0; d < L->ndepth; d++) { S->flags[d] = off; off += 256; }
    S->bytes = off;
    // fine-grained: stores of a peer and the flag words behind them must be visible to a running kernel of the owner
    if (hipExtMallocWithFlags((void **)&S->arena, S->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void **)&S->arena, S->bytes) != hipSuccess) { delete S; suhmo_set_error("ipc transport: arena allocation failed"); return -2; }
    }
    HIPCHK(hipMemset(S->arena, 0, S->bytes));
    HIPCHK(hipHostMalloc((void **)&S->herr, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *S->herr = 0;
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
