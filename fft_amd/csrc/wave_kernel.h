// Lane-distributed FFT for the shortest transforms: a signal of N <= 128 (complex64) or N <= 64
// (complex128) points lives across the lanes of one wavefront, 16 bytes per lane, and never
// touches LDS memory.
//
// Why a second kernel family: in the LDS-resident kernel a 16-point signal is 4 lanes x 4 points,
// so a wave-level load instruction fetches 32-byte pieces of sixteen different signals (one
// quarter of each 128-byte line per instruction) -- 3.2 TB/s at N = 16, 3.8 TB/s at N = 32.  Here
// every lane loads ONE 16-byte vector, lane-consecutive, so a wave instruction moves one fully
// contiguous KiB whatever N is; the transform is radix-2 decimation in frequency across lanes:
//
//   stage with half-size h (in lanes: m = h / K):   partner = lane ^ m
//     lane with bit m clear:  x <- x + x_partner
//     lane with bit m set:    x <- (x_partner - x) * W_{2h}^(i mod h)
//
// The partner exchange is a DPP quad permutation (m = 1, 2) or a ds_swizzle bit-mask pattern
// (m = 4, 8, 16), a ds_bpermute for m = 32: register-to-register through the cross-lane network,
// no LDS allocation, no barrier.  The last stage of a complex64 signal (h = 1) pairs the two
// elements a lane holds.
// Decimation in frequency leaves bin bitrev(i) at element i; the store scatters within the
// wave's own contiguous KiB, so every cache line it touches is written completely.
#pragma once
#include "lds_kernel.h"

#ifndef MFFT_WAVE_UNROLL
#define MFFT_WAVE_UNROLL 4     // 1-KiB steps a wave keeps in flight per loop trip
#endif

namespace mfft {

template <typename T> struct WaveArgs {
    const void* in;        // complex samples, or (REAL) real samples
    void* out;             // complex bins, or (POWER) the real power spectrum
    const cx<T>* tw;       // W_N^j = exp(-2*pi*i*j/N), j < N/2
    const T* win;          // WIN: window of N samples
    int64_t units;         // 16-byte units in the whole batch = batch * N / K
    T scale;
    int grid_over;         // host only: resident sets of workgroups to launch, 0 = the measured default
};

// value of lane (lane ^ MASK), MASK in {1, 2, 4, 8, 16, 32}
template <int MASK> __device__ __forceinline__ int xor_lane_i32(int v)
{
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 32)                                                            // across the two halves
        return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ 32) << 2), v);
    else return __builtin_amdgcn_ds_swizzle(v, (MASK << 10) | 0x1F);                          // bit-mask mode: xor MASK
}
template <int MASK> __device__ __forceinline__ float xor_lane(float v)
{
    return __int_as_float(xor_lane_i32<MASK>(__float_as_int(v)));
}
template <int MASK> __device__ __forceinline__ double xor_lane(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = xor_lane_i32<MASK>((int)(b & 0xffffffffll)), hi = xor_lane_i32<MASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// one cross-lane stage: M = partner distance in lanes
template <typename T, int K, int M>
__device__ __forceinline__ void wave_stage(cx<T> (&x)[K], const cx<T> (&w)[K], bool upper)
{
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const cx<T> o{xor_lane<M>(x[r].x), xor_lane<M>(x[r].y)};
        const cx<T> sum = x[r] + o;
        const cx<T> dif = cmul(o - x[r], w[r]);
        x[r] = upper ? sum : dif;
    }
}

// WIN: x[j] *= win[j] on load; POWER: out = |scale * X|^2 (both: the fused form of BASELINE config 2)
// REAL: the input holds real samples (K per lane, 8 bytes)
template <typename T, int LOG2N, int UNROLL, bool WIN, bool POWER, bool REAL = false>
__global__ void __launch_bounds__(256) fft_wave_kernel(const WaveArgs<T> a)
{
    constexpr int N = 1 << LOG2N;
    constexpr int K = 16 / (int)sizeof(cx<T>);       // elements per lane: 2 (complex64), 1 (complex128)
    constexpr int LOG2K = K == 2 ? 1 : 0;
    constexpr int LANES = N / K;                     // lanes per signal (2..64)
    constexpr int XS = LOG2N - LOG2K;                // cross-lane stages
    static_assert(LANES >= 1 && LANES <= 64, "signal must fit one wavefront");
    using C = cx<T>;
    using V4 = unsigned int __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x & 63;
    const int li = lane & (LANES - 1);               // lane within the signal
    // twiddles of the cross-lane stages, fetched once: stage s has half-size h = N >> (s+1),
    // element i = li*K + r uses W_{2h}^(i mod h) = W_N^((i mod h) << s)
    C w[XS > 0 ? XS : 1][K];
#pragma unroll
    for (int s = 0; s < XS; ++s) {
        const int h = N >> (s + 1);
#pragma unroll
        for (int r = 0; r < K; ++r) w[s][r] = a.tw[((li * K + r) & (h - 1)) << s];
    }
    T win[K];
#pragma unroll
    for (int r = 0; r < K; ++r) win[r] = WIN ? a.win[li * K + r] : (T)1;
    // bit-reversed position of each element, relative to the wave's block
    int pos[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const unsigned i = (unsigned)(li * K + r);
        const unsigned rev = LOG2N > 0 ? (__brev(i) >> (32 - LOG2N)) : 0u;
        pos[r] = (lane - li) * K + (int)rev;         // first element of this signal in the block + bin
    }

    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t blocks = (a.units + 63) >> 6;      // 64 units (1 KiB) per wave step
    for (int64_t b0 = wave * UNROLL; b0 < blocks; b0 += nwaves * UNROLL) {
        C x[UNROLL][K];
        // ---- load: one 16-byte vector per lane and step, all steps in flight together -----------
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t unit = ((b0 + u) << 6) + lane;
            V4 v = {0u, 0u, 0u, 0u};
            if constexpr (REAL) {
                using V2 = unsigned int __attribute__((ext_vector_type(2)));
                V2 r = {0u, 0u};
                if (unit < a.units) r = __builtin_nontemporal_load(reinterpret_cast<const V2*>(a.in) + unit);
                if constexpr (K == 2) v = V4{r.x, 0u, r.y, 0u};      // two floats -> (re, 0), (re, 0)
                else v = V4{r.x, r.y, 0u, 0u};                       // one double -> (re, 0)
            } else {
                if (unit < a.units) v = __builtin_nontemporal_load(reinterpret_cast<const V4*>(a.in) + unit);
            }
            if constexpr (K == 2) {
                x[u][0] = C{__uint_as_float(v.x), __uint_as_float(v.y)};
                x[u][1] = C{__uint_as_float(v.z), __uint_as_float(v.w)};
            } else {
                x[u][0] = C{(T)__longlong_as_double(((long long)v.y << 32) | v.x),
                            (T)__longlong_as_double(((long long)v.w << 32) | v.z)};
            }
            if constexpr (WIN) {
#pragma unroll
                for (int r = 0; r < K; ++r) { x[u][r].x *= win[r]; x[u][r].y *= win[r]; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            // ---- radix-2 DIF across lanes ---------------------------------------------------------
            if constexpr (XS >= 1 && LANES >= 64) wave_stage<T, K, 32>(x[u], w[XS - 6], !(li & 32));
            if constexpr (XS >= 1 && LANES >= 32) wave_stage<T, K, 16>(x[u], w[XS - 5], !(li & 16));
            if constexpr (XS >= 1 && LANES >= 16) wave_stage<T, K, 8>(x[u], w[XS - 4], !(li & 8));
            if constexpr (XS >= 1 && LANES >= 8) wave_stage<T, K, 4>(x[u], w[XS - 3], !(li & 4));
            if constexpr (XS >= 1 && LANES >= 4) wave_stage<T, K, 2>(x[u], w[XS - 2], !(li & 2));
            if constexpr (XS >= 1 && LANES >= 2) wave_stage<T, K, 1>(x[u], w[XS - 1], !(li & 1));
            if constexpr (K == 2) {                   // last stage pairs the lane's own two elements
                const C p = x[u][0], q = x[u][1];
                x[u][0] = p + q;
                x[u][1] = p - q;
            }
            // ---- store: bin bitrev(i) of the same signal, inside the wave's contiguous KiB ---------
            const int64_t base = (b0 + u) << 6;
            if (base + lane < a.units) {
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    const T re = x[u][r].x * a.scale, im = x[u][r].y * a.scale;
                    if constexpr (POWER) st_stream(static_cast<T*>(a.out) + base * K + pos[r], re * re + im * im);
                    else st_stream(static_cast<C*>(a.out) + base * K + pos[r], C{re, im});
                }
            }
        }
    }
}

template <typename T, int LOG2N, bool WIN, bool POWER, bool REAL = false>
int launch_wave_one(const WaveArgs<T>& a, hipStream_t s)
{
    constexpr int UNROLL = MFFT_WAVE_UNROLL;
    auto kern = fft_wave_kernel<T, LOG2N, UNROLL, WIN, POWER, REAL>;
    static thread_local int attr_dev = -1, per_cu = 1;
    int dev = 0;
    MFFT_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        per_cu = blocks_per_cu(reinterpret_cast<const void*>(kern), 256, 0);
        attr_dev = dev;
    }
    const int64_t steps = (a.units + 63) / 64;       // wave steps
    if (steps <= 0) return MFFT_OK;
    const int64_t groups = (steps + 4 * UNROLL - 1) / (4 * UNROLL);   // four waves per workgroup
    // Several resident sets of workgroups instead of one that strides over the batch: the dispatcher then
    // evens out the CUs' different speeds (measured on 2^26-byte-pair batches: 8..17 % up to 64 points;
    // at 128 complex64 points only the power-output forms gain, 7..8 %).
    constexpr int over = sizeof(T) == 4 ? (LOG2N <= 5 ? 16 : LOG2N == 6 ? 8 : POWER ? 8 : 1)
                                        : (LOG2N <= 5 ? 16 : POWER ? 4 : 8);
    hipLaunchKernelGGL(kern, dim3(pick_grid(per_cu, groups, a.grid_over > 0 ? a.grid_over : over)), dim3(256), 0, s, a);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

// complex64: 4..128 points, complex128: 2..64 points
template <typename T>
int launch_wave_any(int log2n, bool win, bool power, bool real_in, const WaveArgs<T>& a, hipStream_t s)
{
    constexpr int MAXLOG = sizeof(T) == 4 ? 7 : 6;
    constexpr int MINLOG = sizeof(T) == 4 ? 2 : 1;
    switch (log2n) {
#define MFFT_W(L) case L: if constexpr (L >= MINLOG && L <= MAXLOG) \
        return real_in ? launch_wave_one<T, L, false, false, true>(a, s) : win ? (power ? launch_wave_one<T, L, true, true>(a, s) : launch_wave_one<T, L, true, false>(a, s)) \
                   : (power ? launch_wave_one<T, L, false, true>(a, s) : launch_wave_one<T, L, false, false>(a, s)); else break;
        MFFT_W(1) MFFT_W(2) MFFT_W(3) MFFT_W(4) MFFT_W(5) MFFT_W(6) MFFT_W(7)
#undef MFFT_W
    }
    set_error("length not served by the lane-distributed kernel");
    return MFFT_E_UNSUPPORTED;
}

}  // namespace mfft
