// Memory-side ceiling of the config-2 access mix (read 8 B/point, write 4 B/point) as a function of
// the bytes each lane moves per instruction: 8-B loads + 4-B stores (what the 1024-point kernel issues)
// against 16-B loads and/or 16-B stores.  One "signal" = 1024 points, a wave owns one signal per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LB, int SB>   // load bytes per lane (8 or 16), store bytes per lane (4 or 16)
__global__ void __launch_bounds__(256) k(const float* in, float* out, long batch)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long b = wave; b < batch; b += nw) {
        const float* src = in + b * 2048;
        float* dst = out + b * 1024;
        float s[16];
        if constexpr (LB == 8) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2*>(src) + lane + 64 * e);
                s[e] = v.x * v.x + v.y * v.y;                 // point lane + 64 e
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + lane + 64 * e);
                s[2 * e] = v.x * v.x + v.y * v.y;             // points 2(lane + 64 e), +1
                s[2 * e + 1] = v.z * v.z + v.w * v.w;
            }
        }
        if constexpr (SB == 4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) __builtin_nontemporal_store(s[e], dst + lane + 64 * e);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f4 v = {s[4 * e], s[4 * e + 1], s[4 * e + 2], s[4 * e + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst) + lane + 64 * e);
            }
        }
    }
}

template <int LB, int SB> int run(const float* in, float* out, long batch, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<LB, SB>), dim3(grid), dim3(256), 0, 0, in, out, batch);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<LB, SB>), dim3(grid), dim3(256), 0, 0, in, out, batch);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double us = best / 200 * 1e3, bytes = 12.0 * 1024 * batch;
    printf("load %2d B/lane, store %2d B/lane, grid %5d: %7.2f us  %6.0f GB/s\n", LB, SB, grid, us, bytes / us / 1e3);
    return 0;
}

int main()
{
    const long batch = 65536;
    float *in, *out;
    CK(hipMalloc(&in, batch * 2048 * sizeof(float)));
    CK(hipMalloc(&out, batch * 1024 * sizeof(float)));
    CK(hipMemset(in, 0, batch * 2048 * sizeof(float)));
    for (int grid : {512, 1024, 2048}) {
        if (run<8, 4>(in, out, batch, grid)) return 1;
        if (run<16, 4>(in, out, batch, grid)) return 1;
        if (run<8, 16>(in, out, batch, grid)) return 1;
        if (run<16, 16>(in, out, batch, grid)) return 1;
    }
    return 0;
}
