// stream_probe.hip -- how does the HBM rate of a streaming kernel depend on the NUMBER of concurrent address streams per wave?
// Context: k_gsrb_fused<2> moves 1.32 GB per launch at 4.66 TB/s in twelve streams (phi, rhs, B, Pi, zb, bx x2, by x2, mask ... per
// wave), a plain copy reaches 6.3 TB/s.  The probe reads N arrays with 16-byte loads per lane and writes one:
//   layout 0  N separate arrays (what the level canvas is)
//   layout 1  the N arrays interleaved in segments of 128 doubles (1 KB = what one wave loads per instruction): the N loads of a wave
//             hit N consecutive kilobytes -- "fewer streams" without changing a single coalesced access
// mode 0: grid-stride over the cells (many waves, each touching every stream once per iteration);
// mode 1: one wave per workgroup marching down the rows of a column strip of 128 cells (the access pattern of the fused kernel).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o gpurun_out/stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N, int LAYOUT>
__global__ __launch_bounds__(256) void k_gridstride(const double *__restrict__ in, double *__restrict__ out, long npairs, long arr_stride)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npairs; p += (long)gridDim.x * blockDim.x) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < N; k++) {
            long idx = 2 * p;
            long a = LAYOUT == 0 ? k * arr_stride + idx : ((idx >> 7) * N + k) * 128 + (idx & 127);
            double2 v = *reinterpret_cast<const double2 *>(in + a);
            acc.x += v.x; acc.y += v.y;
        }
        *reinterpret_cast<double2 *>(out + 2 * p) = acc;
    }
}
// one wave per workgroup: strip of 128 columns, rows [r0, r1): next row's loads are issued before this row's sum is stored
template <int N, int LAYOUT>
__global__ __launch_bounds__(64) void k_march(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int rows_per_chunk, long arr_stride)
{
    const int nstrips = nx / 128;
    const int strip = blockIdx.x % nstrips, chunk = blockIdx.x / nstrips;
    const int r0 = chunk * rows_per_chunk, r1 = min(ny, r0 + rows_per_chunk);
    const int i = strip * 128 + 2 * threadIdx.x;
    double2 cur[N];
    auto load = [&](int r, double2 *dst) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            long idx = (long)r * nx + i;
            long a = LAYOUT == 0 ? k * arr_stride + idx : ((idx >> 7) * N + k) * 128 + (idx & 127);
            dst[k] = *reinterpret_cast<const double2 *>(in + a);
        }
    };
    if (r0 < r1) load(r0, cur);
    for (int r = r0; r < r1; r++) {
        double2 nxt[N];
        if (r + 1 < r1) load(r + 1, nxt);
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < N; k++) { acc.x += cur[k].x; acc.y += cur[k].y; }
        *reinterpret_cast<double2 *>(out + (long)r * nx + i) = acc;
#pragma unroll
        for (int k = 0; k < N; k++) cur[k] = nxt[k];
    }
}

template <int N, int LAYOUT>
int run(const double *in, double *out, int nx, int ny, int mode, int waves)
{
    const long cells = (long)nx * ny;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL((k_gridstride<N, LAYOUT>), dim3(waves / 4), dim3(256), 0, 0, in, out, cells / 2, cells);
        else {
            const int nstrips = nx / 128;
            int nch = waves / nstrips; if (nch < 1) nch = 1;
            const int rpc = (ny + nch - 1) / nch;
            nch = (ny + rpc - 1) / rpc;
            hipLaunchKernelGGL((k_march<N, LAYOUT>), dim3(nstrips * nch), dim3(64), 0, 0, in, out, nx, ny, rpc, cells);
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (N + 1) * 8.0 * cells;
    printf("mode %d  layout %d  N = %2d read streams + 1 write  waves %5d : %.3f ms  %.2f TB/s\n", mode, LAYOUT, N, waves, best, bytes / best / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv)
{
    const int nx = 4096, ny = 4096, NMAX = 10;
    const long cells = (long)nx * ny;
    double *in, *out;
    CK(hipMalloc(&in, NMAX * cells * sizeof(double)));
    CK(hipMalloc(&out, cells * sizeof(double)));
    CK(hipMemset(in, 0, NMAX * cells * sizeof(double)));
    CK(hipMemset(out, 0, cells * sizeof(double)));
    for (int mode = 0; mode < 2; mode++)
        for (int waves : {2048, 4096, 8192}) {
#define RUN(N) if (run<N, 0>(in, out, nx, ny, mode, waves) || run<N, 1>(in, out, nx, ny, mode, waves)) return 1;
            RUN(1) RUN(2) RUN(4) RUN(6) RUN(8) RUN(10)
        }
    printf("done\n");
    return 0;
}
