// Microbenchmark: what does HBM deliver for "G rows x S bytes at row stride L" tile reads
// (the access shape of one global FFT pass) as a function of the run length S?
// Each workgroup copies tiles: reads G runs of S bytes (stride = rowbytes), writes them
// either contiguously (mode 0) or with the same strided shape (mode 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// data viewed as float2 elements; tile = G rows x C elements; L elements per row-stride
template <int G, int C>
__global__ void __launch_bounds__(256) tile_copy(const float2* __restrict__ in, float2* __restrict__ out,
                                                 long L, long tiles, int mode)
{
    constexpr int PER = G * C / 256;                 // elements per thread
    const int c = threadIdx.x % C, r0 = threadIdx.x / C;
    constexpr int RSTEP = 256 / C;
    for (long w = blockIdx.x; w < tiles; w += gridDim.x) {
        const long tiles_per_band = L / C;
        const long band = w / tiles_per_band, j0 = (w % tiles_per_band) * C;
        const float2* src = in + band * (long)G * L + j0 + c;
        float2 v[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] = src[(long)(r0 + RSTEP * e) * L];
        if (mode == 0) {
            float2* dst = out + w * (long)(G * C);
#pragma unroll
            for (int e = 0; e < PER; ++e) dst[threadIdx.x + 256 * e] = v[e];
        } else {
            float2* dst = out + band * (long)G * L + j0 + c;
#pragma unroll
            for (int e = 0; e < PER; ++e) dst[(long)(r0 + RSTEP * e) * L] = v[e];
        }
    }
}

template <int G, int C> void run(const float2* in, float2* out, long n, long L, int mode)
{
    const long tiles = n / ((long)G * C);
    if ((long)G * L > n || L % C != 0 || n % ((long)G * L) != 0) {   // a band of G rows must fit the buffer
        printf("G=%4d run=%5d B  L=%8ld  mode=%d  skipped (band larger than the buffer)\n", G, C * 8, L, mode);
        return;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((tile_copy<G, C>), dim3(grid), dim3(256), 0, 0, in, out, L, tiles, mode);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((tile_copy<G, C>), dim3(grid), dim3(256), 0, 0, in, out, L, tiles, mode);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    printf("G=%4d run=%5d B  L=%8ld  mode=%d  %8.1f us  %7.0f GB/s (read+write)\n", G, C * 8, L, mode, best * 1e3,
           2.0 * n * 8 / (best * 1e-3) / 1e9);
}

int main()
{
    const long n = 1L << 26;                          // 512 MiB of float2
    float2 *in, *out;
    CHECK(hipMalloc(&in, n * 8)); CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMemset(in, 1, n * 8));
    // warm the clocks
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((tile_copy<16, 256>), dim3(2048), dim3(256), 0, 0, in, out, 256L, n / 4096, 0);
    CHECK(hipDeviceSynchronize());
    for (int mode = 0; mode < 2; ++mode) {
        for (long L : {1L << 12, 1L << 16, 1L << 18}) {
            run<256, 16>(in, out, n, L, mode);
            run<128, 32>(in, out, n, L, mode);
            run<64, 64>(in, out, n, L, mode);
            run<32, 128>(in, out, n, L, mode);
            run<16, 256>(in, out, n, L, mode);
        }
    }
    return 0;
}
