// ipc_probe.hip -- the peer-direct halo protocol of suhmo_ipc.hip in isolation, as two PROCESSES sharing one GPU (or one each):
// every process allocates an arena (receive slots + flag words), exports it with hipIpcGetMemHandle, opens the neighbour's, and then,
// per round: a pack kernel stores a message STRAIGHT INTO THE NEIGHBOUR'S receive slot and publishes its sequence number there (last
// block, system-scope release); an unpack kernel waits (a few polling blocks, bounded) for the neighbour's number in its OWN arena, copies
// the slot out and acknowledges into the neighbour's arena, which is what lets the neighbour reuse the slot two rounds later.
// No RCCL, no host round trip per message.  Prints the time per exchange and whether every word arrived.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/ipc_probe.hip -o /tmp/ipc_probe && timeout -k 5 120 /tmp/ipc_probe [doubles per message] [rounds] [finegrained 0/1]
// The parent forks BEFORE any HIP call and never touches the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); _exit(3); } } while (0)
static int g_rank = -1;

struct Flags { unsigned long long arrive[2], ack[2], error; unsigned int count[2], pad[2]; };   // arrive / ack: written by the NEIGHBOUR, polled here

__device__ __forceinline__ bool wait_ge(const unsigned long long *p, unsigned long long v, unsigned long long *err)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 300000000LL) { __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }   // ~3 s at 100 MHz
    }
    return true;
}
// message seq (1-based) of n doubles: src -> the neighbour's slot; waits for the neighbour's acknowledgement of message seq - 2 first
__global__ void k_pack(const double *__restrict__ src, double *__restrict__ remote_slot, long n, unsigned long long seq, Flags *mine, Flags *theirs)
{
    __shared__ int last;
    if (threadIdx.x == 0 && seq > 2) wait_ge(&mine->ack[0], seq - 2, &mine->error);
    __syncthreads();
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) remote_slot[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&mine->count[0], 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        mine->count[0] = 0;
        __threadfence_system();
        __hip_atomic_store(&theirs->arrive[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_unpack(double *__restrict__ dst, const double *__restrict__ my_slot, long n, unsigned long long seq, Flags *mine, Flags *theirs)
{
    __shared__ int last;
    if (threadIdx.x == 0) wait_ge(&mine->arrive[0], seq, &mine->error);
    __syncthreads();
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = __builtin_nontemporal_load(&my_slot[i]);
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&mine->count[1], 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        mine->count[1] = 0;
        __hip_atomic_store(&theirs->ack[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_fill(double *p, long n, double base) { for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = base + (double)i; }
__global__ void k_check(const double *p, long n, double base, unsigned long long *bad)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) if (p[i] != base + (double)i) atomicAdd(bad, 1ull);
}

static int child(int rank, int rfd, int wfd, long n, int rounds, int fine)
{
    g_rank = rank;
    int ndev = 0;
    CHK(hipGetDeviceCount(&ndev));
    CHK(hipSetDevice(rank % ndev));
    const size_t slot = (size_t)n * sizeof(double), arena_bytes = 2 * slot + 4096;
    char *arena = nullptr;
    if (fine) CHK(hipExtMallocWithFlags((void **)&arena, arena_bytes, hipDeviceMallocFinegrained)); else CHK(hipMalloc((void **)&arena, arena_bytes));
    CHK(hipMemset(arena, 0, arena_bytes));
    CHK(hipDeviceSynchronize());
    hipIpcMemHandle_t mine, theirs;
    CHK(hipIpcGetMemHandle(&mine, arena));
    if (write(wfd, &mine, sizeof(mine)) != (ssize_t)sizeof(mine) || read(rfd, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) { fprintf(stderr, "pipe\n"); return 4; }
    char *remote = nullptr;
    CHK(hipIpcOpenMemHandle((void **)&remote, theirs, hipIpcMemLazyEnablePeerAccess));
    Flags *fm = (Flags *)(arena + 2 * slot), *ft = (Flags *)(remote + 2 * slot);
    double *src = nullptr, *dst = nullptr;
    unsigned long long *bad = nullptr;
    CHK(hipMalloc(&src, slot)); CHK(hipMalloc(&dst, slot)); CHK(hipMalloc(&bad, 8)); CHK(hipMemset(bad, 0, 8));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    const int nblk = (int)((n + 255) / 256 < 128 ? (n + 255) / 256 : 128);
    char tok = 1;                                            // both arenas are open before anybody stores into one
    if (write(wfd, &tok, 1) != 1 || read(rfd, &tok, 1) != 1) return 4;
    double ms = 0.0;
    for (int phase = 0; phase < 2; phase++) {                // 0: warm-up + check of every word, 1: timed
        CHK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        const int base_seq = phase * rounds;
        for (int r = 1; r <= rounds; r++) {
            const unsigned long long seq = base_seq + r;
            const double mark = 1000.0 * rank + seq * 1.0e6;
            if (phase == 0) hipLaunchKernelGGL(k_fill, dim3(nblk), dim3(256), 0, st, src, n, mark);
            hipLaunchKernelGGL(k_pack, dim3(nblk), dim3(256), 0, st, src, (double *)(remote + (seq & 1) * slot), n, seq, fm, ft);
            hipLaunchKernelGGL(k_unpack, dim3(nblk), dim3(256), 0, st, dst, (const double *)(arena + (seq & 1) * slot), n, seq, fm, ft);
            if (phase == 0) hipLaunchKernelGGL(k_check, dim3(nblk), dim3(256), 0, st, dst, n, 1000.0 * (1 - rank) + seq * 1.0e6, bad);
        }
        CHK(hipGetLastError());
        CHK(hipStreamSynchronize(st));
        if (phase == 1) ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    unsigned long long hb = 0, err = 0;
    CHK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&err, &fm->error, 8, hipMemcpyDeviceToHost));
    printf("rank %d: %ld doubles per message, %d rounds, %s arena: %.2f us per exchange (pack + unpack), wrong words %llu, timeouts %llu\n",
           rank, n, rounds, fine ? "fine-grained" : "plain hipMalloc", 1e3 * ms / rounds, hb, err);
    fflush(stdout);
    if (write(wfd, &tok, 1) != 1 || read(rfd, &tok, 1) != 1) return 4;      // nobody unmaps while the other may still store
    CHK(hipIpcCloseMemHandle(remote));
    CHK(hipFree(arena)); CHK(hipFree(src)); CHK(hipFree(dst)); CHK(hipFree(bad));
    return (hb || err) ? 5 : 0;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 98328;         // 24 rows x 4097 doubles: one field of a 4096-wide strip
    const int rounds = argc > 2 ? atoi(argv[2]) : 200, fine = argc > 3 ? atoi(argv[3]) : 1;
    int a[2], b[2];
    if (pipe(a) || pipe(b)) return 1;
    pid_t p0 = fork();
    if (p0 == 0) _exit(child(0, a[0], b[1], n, rounds, fine));
    pid_t p1 = fork();
    if (p1 == 0) _exit(child(1, b[0], a[1], n, rounds, fine));
    int s0 = 0, s1 = 0;
    waitpid(p0, &s0, 0); waitpid(p1, &s1, 0);
    printf("exit codes %d %d\n", WEXITSTATUS(s0), WEXITSTATUS(s1));
    return (WEXITSTATUS(s0) || WEXITSTATUS(s1)) ? 1 : 0;
}
