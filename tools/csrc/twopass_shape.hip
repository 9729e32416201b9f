// Microbenchmark: the memory-access shape a TWO-pass 2^24-point complex64 transform (4096 x 4096)
// would have, as bare copies with no arithmetic -- an upper bound on what such passes could reach.
//
// A pass of radix G = 4096 owns tiles of G rows x C contiguous elements at row stride L = N / G =
// 4096 elements; a tile must fit LDS (or registers), so C = 4 (128 KiB) is the widest that does.
//   shape A (first pass):  read G x C strided tile, write one contiguous run of G*C elements
//   shape B (last pass):   read G x C strided tile, write G x C strided tile
// One 1024-thread workgroup per CU (136 KiB of LDS requested, as the real kernel would), persistent
// grid.  `remap` = 1 places the tiles that share 128-byte lines on the same XCD (workgroup w runs
// on XCD w % 8), so that the lines' other sectors are L2 hits instead of repeated HBM fetches.
// C = 8 (64-byte runs; 256 KiB per tile, which fits neither LDS nor a sane register budget) is timed
// for comparison.
//
//   hipcc -O3 --offload-arch=gfx950 tools/csrc/twopass_shape.hip -o build/twopass_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int C, bool NT, int BLOCK = 1024>
__global__ void __launch_bounds__(BLOCK) shape_copy(const float2* __restrict__ in, float2* __restrict__ out, int mode, int remap)
{
    constexpr int G = 4096;
    constexpr long L = 4096;                     // row stride in elements
    constexpr int PER = G * C / BLOCK;            // elements per thread
    constexpr int RSTEP = BLOCK / C;
    constexpr int TILES = (int)(L / C);
    const int c = threadIdx.x % C, r0 = threadIdx.x / C;
    const int nwg = gridDim.x;
    for (int it = 0;; ++it) {
        int w;
        if (remap) {
            // XCD x = blockIdx % 8 takes the contiguous tile range [x*TILES/8, (x+1)*TILES/8)
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = TILES / 8, wg_per_xcd = nwg / 8;
            const int local = it * wg_per_xcd + slot;
            if (local >= per_xcd) break;
            w = xcd * per_xcd + local;
        } else {
            w = it * nwg + blockIdx.x;
            if (w >= TILES) break;
        }
        const long j0 = (long)w * C;
        const float2* src = in + j0 + c;
        float2 v[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const float2* p = src + (long)(r0 + RSTEP * e) * L;
            if (NT) { auto t = __builtin_nontemporal_load(reinterpret_cast<const double*>(p)); v[e] = *reinterpret_cast<float2*>(&t); }
            else v[e] = *p;
        }
        if (mode == 0) {
            float2* dst = out + j0 * G;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                if (NT) __builtin_nontemporal_store(*reinterpret_cast<double*>(&v[e]), reinterpret_cast<double*>(dst + threadIdx.x + BLOCK * e));
                else dst[threadIdx.x + BLOCK * e] = v[e];
            }
        } else {
            float2* dst = out + j0 + c;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                float2* p = dst + (long)(r0 + RSTEP * e) * L;
                if (NT) __builtin_nontemporal_store(*reinterpret_cast<double*>(&v[e]), reinterpret_cast<double*>(p));
                else *p = v[e];
            }
        }
    }
}

// ROT buffer pairs are used in turn: with one pair (256 MiB in all) repeated launches find their input
// in the 256 MiB Infinity Cache, which a real plan -- three buffers, each pass reading what the
// previous one wrote -- does not; four pairs (1 GiB) take the cache out of the picture.
static int g_rot = 1;
template <int C, bool NT, int BLOCK = 1024> void run(const float2* in0, float2* out0, int mode, int remap, size_t lds)
{
    const int grid = 256 * (1024 / BLOCK);
    const long n = 1L << 24;
    auto kern = shape_copy<C, NT, BLOCK>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long n_ = 1L << 24;
#define LAUNCH(i) hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, 0, in0 + ((i) % g_rot) * n_, out0 + ((i) % g_rot) * n_, mode, remap)
    for (int i = 0; i < 20; ++i) LAUNCH(i);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 7; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) LAUNCH(i);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 20 < best) best = ms / 20;
    }
    printf("buffers=%d  run=%4d B  wg=%4d  shape=%c  %s  xcd_remap=%d  %7.1f us per pass  %6.0f GB/s (read+write)\n", g_rot, C * 8, BLOCK, mode ? 'B' : 'A',
           NT ? "nontemporal" : "plain      ", remap, best * 1e3, 2.0 * n * 8 / (best * 1e-3) / 1e9);
}

int main()
{
    const long n = 1L << 24;
    float2 *in, *out;
    CHECK(hipMalloc(&in, 4 * n * 8)); CHECK(hipMalloc(&out, 4 * n * 8));
    CHECK(hipMemset(in, 1, 4 * n * 8));
    const size_t lds = 136 * 1024;
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((shape_copy<8, false>), dim3(256), dim3(1024), 0, 0, in, out, 0, 0);
    CHECK(hipDeviceSynchronize());
    printf("# 2^24 complex64 points (128 MiB in, 128 MiB out); the three-pass plan spends 146-153 us in all,\n"
           "# so two passes win only if A + B < ~150 us, i.e. each well above 3.6 TB/s\n");
    for (g_rot = 1; g_rot <= 4; g_rot *= 4)
    for (int mode = 0; mode < 2; ++mode)
        for (int remap = 0; remap < 2; ++remap) {
            run<4, false>(in, out, mode, remap, lds);
            run<4, true>(in, out, mode, remap, lds);
            run<8, false>(in, out, mode, remap, lds);
            run<8, true>(in, out, mode, remap, lds);
            run<2, false, 512>(in, out, mode, remap, lds / 2);      // 16-byte runs, 64 KiB tiles: two workgroups per CU
            run<2, false, 256>(in, out, mode, remap, lds / 2);
        }
    return 0;
}
