// Microbenchmark: what a 2048-point pass of a 2^31-point complex128 transform could reach with 128-byte rows.
//
// The library's 2048-point tiles are 2048 x 4 complex128 (64-byte rows, 128 KiB: what fits LDS) and move
// 2.7-2.9 TB/s; a 2048 x 8 tile has full 128-byte rows but is 256 KiB -- it would live in the registers of a
// 1024-thread workgroup (16 points per thread) and exchange through LDS in two halves, with no room for a
// second register set, so its load, butterfly and store phases cannot overlap.  This program times exactly that
// access shape as a bare copy (load the whole tile, barrier, store the whole tile; one workgroup per CU because
// the LDS request is what the real kernel would need), beside the 2048 x 4 shape the library runs and the
// 1024 x 8 shape of its other passes for calibration.  An upper bound: the butterflies and exchanges would come
// on top, un-overlapped.
//   shape A (first pass):   G x C tile at row stride L = N / G in, one contiguous run of G * C elements out
//   shape M (middle pass):  G x C tile at row stride L in, G x C tile at row stride NS out
//
//   hipcc -O3 --offload-arch=gfx950 tools/csrc/wide_tile_shape.hip -o build/wide_tile_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LOG2G, int C, int BLOCK>
__global__ void __launch_bounds__(BLOCK) shape_copy(const double2* __restrict__ in, double2* __restrict__ out, long L, long NS, int mode,
                                                    int remap, int nt)
{
    extern __shared__ unsigned char smem[];
    constexpr int G = 1 << LOG2G;
    constexpr int PER = G * C / BLOCK;
    constexpr int RSTEP = BLOCK / C;
    const int c = threadIdx.x % C, r0 = threadIdx.x / C;
    const long tiles = L / C;
    const long per_xcd = tiles / 8, wg_per_xcd = gridDim.x / 8;
    for (long it = 0;; ++it) {
        long w;
        if (remap) {
            const long local = it * wg_per_xcd + (blockIdx.x >> 3);
            if (local >= per_xcd) break;
            w = (long)(blockIdx.x & 7) * per_xcd + local;
        } else {
            w = it * gridDim.x + blockIdx.x;
            if (w >= tiles) break;
        }
        const long j0 = w * C;
        const double2* src = in + j0 + c + (long)r0 * L;
        double2 v[PER];
        if (nt & 1) {
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const double2* p = src + (long)(RSTEP * e) * L;
                v[e].x = __builtin_nontemporal_load(&p->x); v[e].y = __builtin_nontemporal_load(&p->y);
            }
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) v[e] = src[(long)(RSTEP * e) * L];
        }
        if (smem[threadIdx.x] == 77) v[0].x += 1.0;       // keeps the LDS allocation alive
        __syncthreads();
        if (mode == 0) {
            double2* dst = out + j0 * G + threadIdx.x;
#pragma unroll
            for (int e = 0; e < PER; ++e) dst[(long)BLOCK * e] = v[e];
        } else {
            // tile never straddles a block of NS super-butterflies
            double2* dst = out + (j0 / NS) * NS * G + (j0 % NS) + c + (long)r0 * NS;
#pragma unroll
            for (int e = 0; e < PER; ++e) dst[(long)(RSTEP * e) * NS] = v[e];
        }
    }
}

template <int LOG2G, int C, int BLOCK> void run(const double2* in, double2* out, int log2n, int mode, long NS, int remap, int nt, size_t lds)
{
    const long n = 1L << log2n, L = n >> LOG2G;
    auto kern = shape_copy<LOG2G, C, BLOCK>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, 0, in, out, L, NS, mode, remap, nt);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, 0, in, out, L, NS, mode, remap, nt);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("2^%d complex128  tile %4d x %d (%3d-byte rows, %3d KiB)  wg=%4d  shape=%c NS=%-8ld %s  xcd_remap=%d  %8.2f ms per pass  %6.0f GB/s (read+write)\n",
           log2n, 1 << LOG2G, C, C * 16, (1 << LOG2G) * C * 16 / 1024, BLOCK, mode ? 'M' : 'A', mode ? NS : 1L, nt ? "streaming loads" : "plain loads    ",
           remap, best, 2.0 * n * 16 / (best * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 31;
    const long n = 1L << log2n;
    double2 *in, *out;
    CHECK(hipMalloc(&in, n * 16)); CHECK(hipMalloc(&out, n * 16));
    CHECK(hipMemset(in, 1, n * 16));
    const size_t lds = 136 * 1024;
    for (int nt = 0; nt < 2; ++nt) {
        // the shape the library runs: 2048 x 4, 512 threads, XCD-aware order
        run<11, 4, 512>(in, out, log2n, 0, 1, 1, nt, lds);
        run<11, 4, 512>(in, out, log2n, 1, 2048, 1, nt, lds);
        // the wide tile: 2048 x 8 in the registers of 1024 threads
        run<11, 8, 1024>(in, out, log2n, 0, 1, 0, nt, lds);
        run<11, 8, 1024>(in, out, log2n, 1, 2048, 0, nt, lds);
        run<11, 8, 1024>(in, out, log2n, 1, 1L << 20, 0, nt, lds);
        // calibration: the 1024 x 8 tiles of the other two passes
        run<10, 8, 512>(in, out, log2n, 0, 1, 0, nt, lds);
        run<10, 8, 512>(in, out, log2n, 1, 2048, 0, nt, lds);
        run<10, 8, 512>(in, out, log2n, 1, 1L << 21, 0, nt, lds);
    }
    return 0;
}
