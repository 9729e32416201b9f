// LDS-resident batched Stockham FFT for gfx950: one kernel transforms whole
// signals of N = 2^k points without touching HBM between passes.
//
// Work decomposition.  A signal of N points is owned by TX = N/P threads, each
// holding P points in registers.  A pass of radix R lets every thread run P/R
// butterflies on its registers; between passes the points are exchanged through
// LDS in Stockham (autosort) order, so no bit-reversal pass exists.  The read
// side of every exchange is the same for all passes: thread t takes positions
// t + TX*e (e < P), which is also the pattern of the global load that feeds the
// first pass and of the global store that drains the last -- consecutive lanes
// touch consecutive elements everywhere, in HBM and in LDS.
//
// When TX <= 64 a signal lives inside one wavefront: the exchange needs no
// s_barrier (LDS instructions of one wave execute in order), and the waves of a
// workgroup drift apart freely, which is what keeps HBM loads, butterflies and
// stores of different signals overlapped on a CU.  For TX > 64 the exchange is
// bracketed by workgroup barriers.
//
// Workgroups stride over the batch, so per-lane constants (window samples,
// twiddles) are fetched once into registers and reused for every signal a
// workgroup transforms.  How many workgroups a launch asks for -- one resident
// set, several, or one set plus a tail the dispatcher hands out -- is measured
// per length and variant (lds_grid_over, fft_lds_tail_kernel).
#pragma once
#include "common.h"
#include <type_traits>

// Tuning knobs (overridable with -D for A/B builds; the defaults are the measured best).
#ifndef MFFT_TW_SRC
#define MFFT_TW_SRC 1        // fast variants: twiddles from 0 = global (L1/L2), 1 = registers, 2 = LDS table
#endif
#ifndef MFFT_WIN_SRC
#define MFFT_WIN_SRC 0       // hann variant: window from 0 = registers, 2 = global (L1/L2)
#endif
#ifndef MFFT_WAVES_PER_EU
#define MFFT_WAVES_PER_EU 0  // 0 = let the compiler choose; k = cap VGPRs for k waves per SIMD
#endif
#ifndef MFFT_BLOCK_1024_HANN
#define MFFT_BLOCK_1024_HANN 64   // workgroup size of the 1024-point complex64 Hann + power kernels (one wave = one signal)
#endif
#ifndef MFFT_BLOCK_MIN
#define MFFT_BLOCK_MIN 256   // workgroup size when a signal needs fewer threads than this
#endif
#ifndef MFFT_BLOCK_2048
#define MFFT_BLOCK_2048 1    // 1 = 2048-point signals (two waves) run in 128-thread workgroups
#endif
#ifndef MFFT_NT
#define MFFT_NT 1            // 1 = non-temporal global loads/stores in the fast variants (+10 % on the 1024-point batch)
#endif
#ifndef MFFT_FAST_BUF
#define MFFT_FAST_BUF 0      // tuning aid: 1 = buffer-resource accesses (scalar offsets) in the plain / hann+power variants
#endif
#ifndef MFFT_WIN_BUF_FROM
#define MFFT_WIN_BUF_FROM 16384   // windowed fast variants read the window through a descriptor from this length on
#endif
#ifndef MFFT_P8192
#define MFFT_P8192 32        // points per thread of the 8192-point complex64 kernels (16 or 32)
#endif
#ifndef MFFT_PREFETCH
#define MFFT_PREFETCH -1     // fast variants load the next signal before transforming the current one: -1 = where measured faster
#endif                       // (prefetch_default), 0 = never, bit 0 = signals inside one wave, bit 1 = signals spanning several waves
#ifndef MFFT_DIAG_NOFFT
#define MFFT_DIAG_NOFFT 0    // diagnostic builds only: skip the passes (load -> epilogue -> store) to price the memory side
#endif
#ifndef MFFT_R1024
#define MFFT_R1024 16, 4, 16, 1
#endif
#ifndef MFFT_DMA
#define MFFT_DMA 1           // 1 = the 64- and 128-KiB signals fetch their successor by LDS-DMA behind the last exchange (dma_default)
#endif
#ifndef MFFT_VEC
#define MFFT_VEC -1          // packed-form butterflies (fft_core.h): -1 = where measured faster (vec_default), 0 = never, 1 = every complex64 kernel
#endif
// Exchanges address LDS as padded base + immediate: -1 = where measured faster (the LDS-DMA kernels and
// complex128 at 8192 points: +2 % plain, +6 % Hann + power; elsewhere +-1 %, complex64 4096 points
// windowed -7 %), 0 = the LDS-DMA kernels only, 1 = everywhere
#ifndef MFFT_AFFINE
#define MFFT_AFFINE -1
#endif
#ifndef MFFT_STATS_F32
#define MFFT_STATS_F32 1     // complex64 statistics: a thread sums its own values in single precision before the double-precision reduction
#endif
#ifndef MFFT_DIAG_SKIP
#define MFFT_DIAG_SKIP 0     // diagnostic builds only: bit 0 = no butterflies / twiddles, bit 1 = no LDS exchanges (results are wrong)
#endif
// 1 = the one-workgroup-per-CU LDS-DMA kernels (128-KiB signals) touch the next signal's lines at the start of a
// trip, so that the fetch behind the last exchange finds them in L2 (complex64 16384 points +6..12 %, complex128
// 8192 points +6..9 %, for ~7 % more fetched bytes)
#ifndef MFFT_DMA_WARM
#define MFFT_DMA_WARM 1
#endif
// 1: the centroid sum of the fast statistics variants from compile-time weights (see the epilogue)
#ifndef MFFT_STATS_CLOSED
#define MFFT_STATS_CLOSED 1
#endif
// 1: the LDS-DMA kernels combine the waves' statistics in LDS and store them; 0: atomics per wave
#ifndef MFFT_DMA_STATS_LDS
#define MFFT_DMA_STATS_LDS 1
#endif
#ifndef MFFT_DIAG_STAMPS
#define MFFT_DIAG_STAMPS 0   // diagnostic builds only: per-workgroup cycle counts of the LDS-DMA pipeline's phases (mfft_diag_read)
#endif

namespace mfft {

#if MFFT_DIAG_STAMPS
__device__ unsigned long long g_diag_stamps[16 * 2048];
#endif

enum { TW_GLOBAL = 0, TW_REG = 1, TW_LDS = 2, TW_LDSC = 3 };   // TW_LDSC: LDS table, large passes keep w^1 and w^4 only
enum { WIN_REG = 0, WIN_GLOBAL = 2, WIN_BUF = 3 };

// Waves per SIMD the register allocator must leave room for (1 = no constraint).  In its
// 16-point-per-thread form (MFFT_P8192 = 16) the 8192-point complex64 kernel (512 threads, 68 KiB of
// LDS) fits two workgroups per CU only at <= 128 VGPRs; holding the plain variants there is worth
// +9..14 % (3.7 -> 4.05..4.3 TB/s).  Every other size, the general variants (which then spill) and
// the default 32-point form measured slower when capped.
template <typename T, int N, int VAR> constexpr int min_waves_per_eu()
{
#if MFFT_WAVES_PER_EU > 0
    return MFFT_WAVES_PER_EU;
#else
    return (MFFT_P8192 == 16 && sizeof(T) == 4 && N == 8192 && (VAR == V_PLAIN || VAR == V_SEG)) ? 4 : 1;
#endif
}
#define MFFT_LAUNCH_BOUNDS(B, T, N, VAR) __launch_bounds__(B, (min_waves_per_eu<T, N, VAR>()))

// Streaming accesses: each input element is read once and each output written once.
template <typename T> struct nt_vec { using type = T __attribute__((ext_vector_type(2))); };
template <typename T> __device__ __forceinline__ cx<T> ld_stream(const cx<T>* p)
{
#if MFFT_NT
    using V = typename nt_vec<T>::type;
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    return cx<T>{v.x, v.y};
#else
    return *p;
#endif
}
template <typename T> __device__ __forceinline__ void st_stream(cx<T>* p, cx<T> v)
{
#if MFFT_NT
    using V = typename nt_vec<T>::type;
    V w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<V*>(p));
#else
    *p = v;
#endif
}
template <typename T> __device__ __forceinline__ T ld_stream_real(const T* p)
{
#if MFFT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T> __device__ __forceinline__ void st_stream(T* p, T v)
{
#if MFFT_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Buffer-resource accesses for the runtime prologue / epilogue: the hardware range check
// (byte offset = voffset + soffset against num_records) returns zero for loads beyond the
// extent and drops stores beyond it, which IS zero padding, truncation and the inactive-slot
// guard, with no index clamps, no predicates and one lane-dependent offset register.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kBufWord3 = 0x00020000u;      // raw buffer, 32-bit data format

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    // the base is wave-uniform by construction (callers pass one signal per wave or more);
    // say so, or the compiler wraps every access in a waterfall loop
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    void* q = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), (int)kBufWord3);
}
// the same for a base the compiler can see is uniform (kernel arguments and block indices only)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, (int)kBufWord3);
}
// AUX: cache-policy bits of the instruction (0 = default, 2 = streaming / non-temporal)
template <typename V, int AUX = 0> __device__ __forceinline__ V buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (sizeof(V) == 4) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
    else if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <typename V, int AUX = 0> __device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, V v)
{
    if constexpr (sizeof(V) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, AUX);
    else if constexpr (sizeof(V) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, AUX);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, AUX);
}

// ---- pieces of the LDS-DMA pipeline (dma_default) -------------------------------------------
// Workgroup barrier for LDS traffic alone: it waits for this wave's LDS operations and not for the
// vector-memory counter, so global stores and LDS-DMA loads stay in flight across it
// (__syncthreads() carries a fence that drains them).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// One wave-instruction of LDS-DMA: 64 lanes x 16 bytes, lane l from the wave-uniform address sbase
// plus its own byte offset voff, into the 1 KiB of LDS that starts at the wave-uniform byte address
// lds_dst.  No register destination; completion is counted on vmcnt like any load.  M0 holds the
// destination and belongs to the compiler, hence saved and restored inside the statement.
__device__ __forceinline__ void dma_1k(const void* sbase, uint32_t voff, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// Cache warm-up by LDS-DMA: each lane reads ONE dword (from the line at its own address) into a 256-byte
// LDS scratch nobody reads, which brings 64 lines into L2 for the price of one instruction.
__device__ __forceinline__ void dma_touch(const void* addr, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(addr), "s"(lds_dst) : "memory");
}
// wait until at most LEFT vector-memory operations of this wave are outstanding (they retire in issue order)
template <int LEFT> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LEFT) : "memory");
}

template <int R0, int R1, int R2, int R3, bool SPLIT16 = true> struct Radices {
    static constexpr int r[4] = {R0, R1, R2, R3};
    static constexpr bool split16 = SPLIT16;   // radix-16 passes keep six twiddle generators (see tw_rows)
    static constexpr int npass = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static constexpr int ns(int s) { return s == 0 ? 1 : (s == 1 ? R0 : (s == 2 ? R0 * R1 : R0 * R1 * R2)); }
    static constexpr int n = R0 * R1 * R2 * R3;
};

// Entries of the twiddle table of pass s (s >= 1): rows * NS, laid out [row][m].
// Rows of a pass' twiddle table.  A radix-R butterfly needs w^1 .. w^(R-1) with w = W^(m);
// a radix-16 pass stores only the six powers 1, 2, 3, 4, 8, 12 (rows 0..5) and forms
// w^(4a+b) = w^(4a) * w^(b) on the fly: 2.5x fewer table entries, registers and loads for
// nine extra complex multiplies per butterfly.  A radix-32 pass likewise stores 1, 2, 3 and
// 4, 8, .. 28: ten rows instead of thirty-one.
constexpr int tw_rows(int r, bool split16) { return (r >= 16 && split16) ? 2 + r / 4 : r - 1; }
// power of w held by row `row` of a split table
constexpr int tw_split_power(int row) { return row < 3 ? row + 1 : 4 * (row - 2); }
template <class RD> constexpr int tw_pass_elems(int s) { return tw_rows(RD::r[s], RD::split16) * RD::ns(s); }
template <class RD> constexpr int tw_pass_offset(int s)
{
    int off = 0;
    for (int q = 1; q < s; ++q) off += tw_pass_elems<RD>(q);
    return off;
}

// Compact LDS table (TW_LDSC): a split radix-16 pass of NS >= kTwCompactFrom keeps only w^1 and w^4
// per position and forms w^2 = (w^1)^2, w^3 = w^2 w^1, w^8 = (w^4)^2, w^12 = w^8 w^4 on the fly
// (at most two roundings more than a table entry).
constexpr int kTwCompactFrom = 512;
template <class RD> constexpr bool twc_compact(int s) { return RD::r[s] == 16 && RD::split16 && RD::ns(s) >= kTwCompactFrom; }
template <class RD> constexpr int twc_rows(int s) { return twc_compact<RD>(s) ? 2 : tw_rows(RD::r[s], RD::split16); }
template <class RD> constexpr int twc_pass_offset(int s)
{
    int off = 0;
    for (int q = 1; q < s; ++q) off += twc_rows<RD>(q) * RD::ns(q);
    return off;
}
template <class RD> constexpr int twc_total_elems()
{
    int c = 0;
    for (int s = 1; s < RD::npass; ++s) c += twc_rows<RD>(s) * RD::ns(s);
    return c;
}
template <typename T, class RD, int S, int BLOCK>
__device__ __forceinline__ void fill_twc(cx<T>* twl, const cx<T>* __restrict__ tw)
{
    if constexpr (S < RD::npass) {
        constexpr int NS = RD::ns(S), off = tw_pass_offset<RD>(S), offc = twc_pass_offset<RD>(S);
        for (int i = threadIdx.x; i < twc_rows<RD>(S) * NS; i += BLOCK) {
            const int r = i / NS, m = i % NS;
            twl[offc + i] = tw[off + (twc_compact<RD>(S) ? 3 * r : r) * NS + m];     // compact: rows 0 and 3 (w^1, w^4)
        }
    }
}

template <int TX, bool RAW = false> __device__ __forceinline__ void exchange_sync()
{
    if constexpr (TX <= 64) {
        // signal confined to one wave: only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if constexpr (RAW) {
        lds_barrier();
    } else {
        __syncthreads();
    }
}

template <typename T> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

// Register-resident twiddles: entries pass s occupies, and where they start.
template <class RD, int TX, int P> constexpr int twreg_pass_count(int s)
{
    return s == 0 ? 0 : ((RD::ns(s) <= TX) ? 1 : P / RD::r[s]) * tw_rows(RD::r[s], RD::split16);
}
template <class RD, int TX, int P> constexpr int twreg_pass_offset(int s)
{
    int off = 0;
    for (int q = 1; q < s; ++q) off += twreg_pass_count<RD, TX, P>(q);
    return off;
}

// One Stockham pass on the registers of one thread; leaves v[u + U*q] = output q
// of butterfly u.
template <typename T, class RD, int S, int P, int TX, int TWSRC, bool VEC = false>
__device__ __forceinline__ void pass_compute(cx<T> (&v)[P], int t, const cx<T>* __restrict__ tw,
                                             const cx<T>* twr, const cx<T>* twl)
{
    constexpr int R = RD::r[S], U = P / R, NS = RD::ns(S);
    constexpr int twr_pos = twreg_pass_offset<RD, TX, P>(S);
    if constexpr ((MFFT_DIAG_SKIP & 1) != 0) return;
    if constexpr (S > 0) {
        constexpr int off = tw_pass_offset<RD>(S);
        constexpr int ROWS = tw_rows(R, RD::split16);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = (t + TX * u) % NS;
            cx<T> row[ROWS];
            if constexpr (TWSRC == TW_LDSC && twc_compact<RD>(S)) {
                constexpr int offc = twc_pass_offset<RD>(S);
                static_assert(ROWS == 6, "compact tables serve split radix-16 passes");
                const cx<T> w1 = twl[offc + m], w4 = twl[offc + NS + m];
                row[0] = w1; row[1] = cmul<T, VEC>(w1, w1); row[2] = cmul<T, VEC>(row[1], w1);
                row[3] = w4; row[4] = cmul<T, VEC>(w4, w4); row[5] = cmul<T, VEC>(row[4], w4);
            } else
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if constexpr (TWSRC == TW_LDSC) {
                    row[r] = twl[twc_pass_offset<RD>(S) + r * NS + m];
                } else if constexpr (TWSRC == TW_REG) {
                    // register copy: one set per thread when m does not depend on u
                    row[r] = twr[twr_pos + ((NS <= TX) ? 0 : u * ROWS) + r];
                    if constexpr (ROWS != R - 1) {
                        // keep the nine derived powers inside the loop: hoisted, they would
                        // occupy 18 more registers for the whole kernel
                        asm volatile("" : "+v"(row[r].x), "+v"(row[r].y));
                    }
                } else if constexpr (TWSRC == TW_LDS) {
                    row[r] = twl[off + r * NS + m];
                } else {
                    row[r] = tw[off + r * NS + m];
                }
            }
            if constexpr (ROWS == R - 1) {
#pragma unroll
                for (int i = 1; i < R; ++i) v[u + U * i] = cmul<T, VEC>(v[u + U * i], row[i - 1]);
            } else {
                // rows: w^1, w^2, w^3, w^4, w^8, w^12 (, .. w^28 for radix 32)
#pragma unroll
                for (int i = 1; i < R; ++i) {
                    const int a4 = i >> 2, b = i & 3;
                    cx<T> w;
                    if (b == 0) w = row[2 + a4];                      // w^4, w^8, w^12
                    else if (a4 == 0) w = row[b - 1];                 // w^1, w^2, w^3
                    else w = cmul<T, VEC>(row[2 + a4], row[b - 1]);
                    v[u + U * i] = cmul<T, VEC>(v[u + U * i], w);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) dft_strided<T, R, U, VEC>(&v[u]);
}

// Scatter the pass outputs into LDS in Stockham order and gather the next
// pass' inputs (positions t + TX*e).
// Padded LDS position of base + c for a compile-time c.  Positions of one exchange differ from a
// per-thread base by multiples of NS (writes) or TX (reads); base mod 16 is below min(NS, 16) and
// the low four bits of such a multiple leave room for it, so no carry crosses bit 4 and
// pad(base + c) = pad(base) + pad(c): one address register per butterfly, immediate offsets for the rest.
template <int SHIFT> MFFT_HD int lds_pad_off(int c) { return lds_pad<SHIFT>(c); }

template <typename T, class RD, int S, int P, int TX, int PADSHIFT, bool RAW = false>
__device__ __forceinline__ void pass_exchange(cx<T> (&v)[P], int t, cx<T>* lds)
{
    constexpr int R = RD::r[S], U = P / R, NS = RD::ns(S);
    constexpr bool AFFINE = RAW || MFFT_AFFINE > 0 || (MFFT_AFFINE < 0 && sizeof(T) == 8 && TX * P == 8192);
    if constexpr ((MFFT_DIAG_SKIP & 2) != 0) return;
    if constexpr (TX > 64) {                  // everyone has finished reading the previous contents
        if constexpr (RAW) lds_barrier(); else __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = t + TX * u;
        const int base = (j / NS) * (NS * R) + (j % NS);
        if constexpr (AFFINE) {
            cx<T>* w = lds + lds_pad<PADSHIFT>(base);
#pragma unroll
            for (int q = 0; q < R; ++q) w[lds_pad_off<PADSHIFT>(q * NS)] = v[u + U * q];
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) lds[lds_pad<PADSHIFT>(base + q * NS)] = v[u + U * q];
        }
    }
    exchange_sync<TX, RAW>();
    if constexpr (AFFINE) {
        const cx<T>* r = lds + lds_pad<PADSHIFT>(t);
#pragma unroll
        for (int e = 0; e < P; ++e) v[e] = r[lds_pad_off<PADSHIFT>(TX * e)];
    } else {
#pragma unroll
        for (int e = 0; e < P; ++e) v[e] = lds[lds_pad<PADSHIFT>(t + TX * e)];
    }
}

// Copy the twiddles pass S needs for this thread into its register array.
template <typename T, class RD, int S, int P, int TX>
__device__ __forceinline__ void preload_twiddles(cx<T>* twr, int t, const cx<T>* __restrict__ tw)
{
    constexpr int R = RD::r[S], U = P / R, NS = RD::ns(S);
    constexpr int off = tw_pass_offset<RD>(S);
    constexpr int pos = twreg_pass_offset<RD, TX, P>(S);
    constexpr int sets = (NS <= TX) ? 1 : U;
#pragma unroll
    for (int u = 0; u < sets; ++u) {
        const int m = (t + TX * u) % NS;
#pragma unroll
        for (int r = 0; r < tw_rows(R, RD::split16); ++r) twr[pos + u * tw_rows(R, RD::split16) + r] = tw[off + r * NS + m];
    }
}

template <class RD, int TX, int P> constexpr int twreg_count()
{
    int c = 0;
    for (int s = 1; s < RD::npass; ++s) {
        const int R = RD::r[s], U = P / R, NS = RD::ns(s);
        c += ((NS <= TX) ? 1 : U) * tw_rows(R, RD::split16);
    }
    return c;
}

template <class RD> constexpr int tw_total_elems()
{
    int c = 0;
    for (int s = 1; s < RD::npass; ++s) c += tw_pass_elems<RD>(s);
    return c;
}

// Which source the fast variants take their twiddles / window from, and the LDS
// bytes the shared tables need behind the data slots.
template <typename T, class RD, int TX, int P, int VAR> constexpr int tw_source()
{
    if (RD::npass < 2) return TW_GLOBAL;
    if (MFFT_TW_SRC == TW_REG) {
        // measured exceptions to "registers where the set is small enough":
        // complex128 from 1024 points reads the (L1/L2-resident) table per signal: a register copy
        // is 4 VGPRs per twiddle there (plain +2..6 %, Hann + power at 2048 points +13 %);
        // the 4096-point complex64 Hann + power kernel, which also keeps 16 window samples per
        // lane, takes the table from LDS (4.7 -> 5.3 TB/s)
        if (sizeof(T) == 8 && RD::n >= 1024) return TW_GLOBAL;
        if (sizeof(T) == 4 && RD::n == 4096 && VAR == V_HANN_POWER && tw_total_elems<RD>() <= 2048) return TW_LDS;
        return twreg_count<RD, TX, P>() <= (RD::split16 ? 12 : 18) ? TW_REG : TW_GLOBAL;
    }
    if (MFFT_TW_SRC == TW_LDS) return tw_total_elems<RD>() <= 2048 ? TW_LDS : TW_GLOBAL;
    return TW_GLOBAL;
}
template <typename T, class RD, int TX, int P, int VAR> constexpr size_t table_lds_bytes()
{
    size_t b = 0;
    if (tw_source<T, RD, TX, P, VAR>() == TW_LDS) b += (size_t)tw_total_elems<RD>() * sizeof(cx<T>);
    return b;
}

// Where the next signal is fetched by LDS-DMA (below, "LDS-DMA pipeline"): signals of 64 and 128 KiB, i.e.
// two or one resident workgroups per CU, dense complex input, every fast variant (+8..30 %).  Measured and
// left out: 2048 / 4096 complex64 points (-3..-9 %: three or more workgroups per CU overlap by themselves)
// and 4096 complex128 points (+-0 plain, +3..5 % with power output).
template <typename T, int N, int VAR> constexpr bool dma_default()
{
    return MFFT_DMA != 0 && ((sizeof(T) == 4 && (N == 16384 || N == 8192)) || (sizeof(T) == 8 && N == 8192)) &&
           (VAR == V_PLAIN || VAR == V_SEG || VAR == V_WIN || VAR == V_POW || VAR == V_HANN_POWER ||
            VAR == V_HANN_POWER_STATS || VAR == V_POW_STATS);
}

// Where the packed-form butterflies measured faster (MI355X, complex64, 2^26 points per launch):
// 2048 and 4096 points +2 % plain, +5..6 % Hann + power (but the 4096-point windowed, complex-output
// variant -4 %: excluded); 8192 points power-only +9 % and the LDS-DMA kernels; every 16384-point
// variant +10..19 %.  Within +-1.5 % at 256..1024 points and in the other 8192-point forms, which
// keep the scalar form.
template <typename T, int N, int VAR, bool DMA> constexpr bool vec_default()
{
    if (sizeof(T) != 4) return false;
    if (MFFT_VEC >= 0) return MFFT_VEC != 0;
    if (N == 4096) return VAR != V_WIN;
    if (N == 8192) return DMA || VAR == V_POW || VAR == V_POW_STATS;
    return DMA || N == 2048 || N == 16384;
}

// Loading the next signal into registers ahead of the butterflies (MFFT_PREFETCH) is off everywhere.
// Signals inside one wave never gained (the waves of a workgroup drift apart and overlap loads with
// butterflies by themselves; 1024 points Hann + power: 139-141 against 134 us); from 8192 points the
// extra registers spill, and complex128 loses 1..17 % at every length.  complex64 2048 / 4096 points
// gained 2..14 % with the scalar-form butterflies; with the packed form (fewer vector instructions per
// trip, more room for the other workgroups of the CU to cover the loads) the register cost dominates
// (MI355X, 2^26 points per launch, prefetch on -> off, GB/s): 2048 points plain 5631 -> 5704, Hann + power
// 5165 -> 5363, power + statistics 4954 -> 5195, Hann + power + statistics 4600 -> 5066; 4096 points power
// 5584 -> 5680, power + statistics 5038 -> 5148, Hann + power + statistics 4430 -> 4950.
template <typename T, int N, int VAR> constexpr bool prefetch_default()
{
    return false;
}

// LDS-DMA pipeline (DMA = true; one signal per workgroup, dense complex input).  With one or two
// resident workgroups per CU little overlaps a signal's load, butterfly and store phases, and neither
// registers nor LDS hold a second signal.  Here the NEXT signal is fetched by LDS-DMA -- loads without
// a register destination -- into the exchange buffer as soon as the last exchange has been read out of
// it; the last pass and the stores follow, and the wait at the end of the trip leaves exactly the P
// stores outstanding, which drain while the next signal, already on chip, is being transformed.
// Inside the loop every barrier is an LDS-only barrier, every twiddle comes from LDS and the window
// from registers, because a compiler-counted global load or a __syncthreads() fence would wait for
// everything outstanding.  (Measured and not kept: requesting half of the next signal already when
// the butterflies begin, into LDS freed by exchanging through a half-size buffer -- twice the
// barriers; 16384 points plain 215.8 against 215.4 us, Hann 201 against 212, Hann + power 184 against 181.)
// The kernel's body over the groups g_first, g_first + g_step, ... below g_end (all three uniform).
template <typename T, int P, class RD, int PADSHIFT, int BLOCK, int VAR, bool DMA>
__device__ __forceinline__ void lds_body(const KArgs<T>& a, const int64_t g_first, const int64_t g_end, const unsigned g_step)
{
    using C = cx<T>;
    constexpr int N = RD::n;
    constexpr int TX = N / P;
    constexpr int SLOTS = BLOCK / TX;
    constexpr int NPASS = RD::npass;
    constexpr int LDS_ELEMS = ((lds_pad<PADSHIFT>(N - 1) + 1 + 1) / 2) * 2;   // keep slots 16-byte aligned
    constexpr bool FASTSTATS = (VAR == V_HANN_POWER_STATS || VAR == V_POW_STATS);
    constexpr bool POWER = (VAR == V_HANN_POWER || VAR == V_GEN_POWER || VAR == V_POW || FASTSTATS);
    constexpr bool FASTWIN = (VAR == V_HANN_POWER || VAR == V_WIN || VAR == V_HANN_POWER_STATS);   // window samples held per lane
    constexpr bool GENERAL = (VAR == V_GEN_CPLX || VAR == V_GEN_POWER);
    constexpr int TWN = twreg_count<RD, TX, P>();
    constexpr int TWSRC = DMA ? TW_LDSC : tw_source<T, RD, TX, P, VAR>();
    constexpr bool TWREG = TWSRC == TW_REG;
    constexpr bool VEC = vec_default<T, N, VAR, DMA>();
    static_assert(!DMA || (SLOTS == 1 && TX > 64 && !GENERAL && VAR != V_REAL && NPASS >= 2),
                  "the LDS-DMA pipeline serves dense complex signals that own their workgroup");
    // the 1024-thread workgroup of 16384 points has no registers to spare for sixteen window samples
    // (140 B/lane of scratch): it reads the L2-resident table through one descriptor and scalar
    // offsets instead, 2.4 -> 3.5 TB/s; with 16 points per thread at 8192 the register copy is 5-7 % faster
    // (likewise the 32-point-per-thread kernels: 32 more registers would halve their occupancy,
    // 8192 points Hann + power 3.2 against 4.25 TB/s)
    // (the LDS-DMA kernels need far fewer registers and must not load inside their loop: register copy)
    constexpr int WINSRC = DMA ? WIN_REG : (MFFT_WIN_SRC == WIN_REG && (N >= MFFT_WIN_BUF_FROM || P >= 32)) ? WIN_BUF : MFFT_WIN_SRC;
    static_assert(BLOCK % TX == 0 && SLOTS >= 1, "block must hold whole signals");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (the LDS-DMA kernels own their workgroup: a slot the compiler can see is zero keeps the signal's
    // base addresses provably uniform, i.e. in scalar registers and out of waterfall loops)
    const int slot = DMA ? 0 : threadIdx.x / TX;
    const int t = DMA ? (int)threadIdx.x : threadIdx.x % TX;
    C* lds = reinterpret_cast<C*>(smem_raw) + (size_t)slot * LDS_ELEMS;
    // LDS-DMA pipeline: [exchange buffer, which also receives the next signal as a lane-linear image] [twiddle tables]
    constexpr int DMA_TABLES_AT = LDS_ELEMS * (int)sizeof(C);
    constexpr int DMA_WARM_AT = DMA_TABLES_AT + twc_total_elems<RD>() * (int)sizeof(C);      // 256 bytes per wave, never read
    constexpr int DMA_STATS_AT = DMA_WARM_AT + (MFFT_DMA_WARM != 0 ? BLOCK * 4 : 0);         // 3 doubles per wave

    // ---- tables shared by the workgroup, staged in LDS behind the data slots --
    C* twl = DMA ? reinterpret_cast<C*>(smem_raw + DMA_TABLES_AT)
                 : reinterpret_cast<C*>(smem_raw) + (size_t)SLOTS * LDS_ELEMS;
    if constexpr (TWSRC == TW_LDS) {
        for (int i = threadIdx.x; i < tw_total_elems<RD>(); i += BLOCK) twl[i] = a.tw[i];
    }
    if constexpr (TWSRC == TW_LDSC) {
        fill_twc<T, RD, 1, BLOCK>(twl, a.tw);
        fill_twc<T, RD, 2, BLOCK>(twl, a.tw);
        fill_twc<T, RD, 3, BLOCK>(twl, a.tw);
        __syncthreads();
    }
    if constexpr (TWSRC == TW_LDS) __syncthreads();

    // ---- per-lane constants, fetched once ---------------------------------
    T wreg[(FASTWIN && WINSRC == WIN_REG) ? P : 1];
    if constexpr (FASTWIN && WINSRC == WIN_REG) {
#pragma unroll
        for (int e = 0; e < P; ++e) wreg[e] = a.win[t + TX * e];
    }
    C twr[TWREG ? TWN : 1];
    if constexpr (TWREG) {
        if constexpr (NPASS > 1) preload_twiddles<T, RD, 1, P, TX>(twr, t, a.tw);
        if constexpr (NPASS > 2) preload_twiddles<T, RD, 2, P, TX>(twr, t, a.tw);
        if constexpr (NPASS > 3) preload_twiddles<T, RD, 3, P, TX>(twr, t, a.tw);
    }

    // Load of one signal in the fast variants (pass-0 inputs: positions t + TX*e).  A slot beyond the
    // batch re-reads the last signal and skips its stores: no branch here.
    const auto load_fast = [&](C (&dst)[P], int64_t bsig) {
        const int64_t bl = bsig < a.batch ? bsig : a.batch - 1;
        if constexpr (VAR == V_SEG) {
            const C* __restrict__ src = reinterpret_cast<const C*>(a.in) + bl * a.in_stride;
            // Element j = t + TX*e sits at (j >> lseg) * seg_stride + (j & (seg - 1)).  Segments are powers
            // of two of at least one wave's 64 elements, so either a segment holds whole TX-element rows or
            // a row holds whole segments: no carry between t and TX*e, and the position splits into a
            // lane part (from t) and a part that depends on e and the kernel arguments only -- wave-uniform
            // (scalar registers), so one lane-dependent address serves all P accesses
            const C* __restrict__ lane = src + seg_pos(t, a.in_lseg, a.in_seg_stride);
#pragma unroll
            for (int e = 0; e < P; ++e) dst[e] = ld_stream(lane + seg_pos(TX * e, a.in_lseg, a.in_seg_stride));
        } else if constexpr (VAR == V_REAL) {
            const T* __restrict__ src = reinterpret_cast<const T*>(a.in) + bl * a.in_stride;
#pragma unroll
            for (int e = 0; e < P; ++e) dst[e] = C{ld_stream_real(src + t + TX * e), (T)0};
        } else {
            const C* __restrict__ src = reinterpret_cast<const C*>(a.in) + bl * a.in_stride;
            if constexpr (MFFT_FAST_BUF != 0 && TX >= 64) {
                const auto rs = make_rsrc(src, (uint32_t)N * sizeof(C));
#pragma unroll
                for (int e = 0; e < P; ++e) dst[e] = buf_ld<C, MFFT_NT ? 2 : 0>(rs, (uint32_t)t * sizeof(C), TX * e * sizeof(C));
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) dst[e] = ld_stream(src + t + TX * e);
            }
        }
    };
    // Software pipeline (fast variants): the next signal's loads are issued before the current one
    // is transformed, so the bus does not idle during the butterflies.
    constexpr bool PREFETCH = !GENERAL && (MFFT_PREFETCH < 0 ? prefetch_default<T, N, VAR>()
                                                             : (((MFFT_PREFETCH & 1) != 0 && TX <= 64) || ((MFFT_PREFETCH & 2) != 0 && TX > 64)));
    C vn[PREFETCH ? P : 1];
    if constexpr (PREFETCH) {
        if (g_first < g_end) load_fast(vn, g_first * SLOTS + slot);
    }
    // LDS-DMA of one signal into the (then idle) exchange buffer as a lane-linear image: element j at
    // byte j * sizeof(C).  A piece is one wave-instruction = 1 KiB = PE consecutive elements, which lie
    // inside one row of TX elements, hence inside one segment; wave w moves pieces w, w + NW, ...
    constexpr int PE = 1024 / (int)sizeof(C), NW = BLOCK / 64;
    constexpr int PPW = DMA ? (N / PE) / NW : 1;
    static_assert(!DMA || PPW * NW * PE == N, "whole pieces per wave");
    const uint32_t stage_lds = (uint32_t)reinterpret_cast<size_t>(smem_raw);       // LDS byte address of the exchange buffer
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const auto dma_signal = [&](int64_t bsig) {
        const C* __restrict__ src = reinterpret_cast<const C*>(a.in) + bsig * a.in_stride;      // wave-uniform
        const uint32_t lane_off = (threadIdx.x & 63u) * 16u;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wv + NW * i;
            const int j0 = piece * PE;
            int64_t off = j0;
            if constexpr (VAR == V_SEG) off = seg_pos(j0, a.in_lseg, a.in_seg_stride);     // a piece lies inside one segment (seg >= PE)
            dma_1k(src + off, lane_off, stage_lds + (uint32_t)piece * 1024u);
        }
    };
#if MFFT_DIAG_STAMPS
    unsigned long long stamp_acc[16] = {};
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    [[maybe_unused]] const unsigned long long diag_t0 = stamp_prev;
#define MFFT_STAMP(k) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now__ - stamp_prev; stamp_prev = now__; } while (0)
#else
#define MFFT_STAMP(k) do { } while (0)
#endif
    // lines of the wave's own pieces: touch k, lane l -> piece wv + NW * (8k + (l >> 3)), line l & 7 of it.
    // One workgroup per CU only (128 KB signals): with two, the other workgroup covers the latency.
    constexpr bool WARM = DMA && MFFT_DMA_WARM != 0 && N * (int)sizeof(C) >= 128 * 1024 && PPW % 8 == 0;
    const auto dma_warm = [&](int64_t bsig) {
        const C* __restrict__ src = reinterpret_cast<const C*>(a.in) + bsig * a.in_stride;
        const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
        for (int k = 0; k < PPW / 8; ++k) {
            const int j0 = (wv + NW * (8 * k + (int)(lane >> 3))) * PE;       // first element of the lane's piece
            int64_t off = j0;
            if constexpr (VAR == V_SEG) off = seg_pos(j0, a.in_lseg, a.in_seg_stride);
            dma_touch(reinterpret_cast<const char*>(src + off) + (lane & 7u) * 128u, stage_lds + (uint32_t)DMA_WARM_AT + (uint32_t)wv * 256u);
        }
    };
    if constexpr (DMA) {
        if (g_first < g_end) dma_signal(g_first);
        wait_vm<0>();
        lds_barrier();
    }
    for (int64_t g = g_first; g < g_end; g += g_step) {
        const int64_t b = g * SLOTS + slot;
        const bool active = b < a.batch;
        C v[P];
        // LDS-DMA pipeline: every address of a trip is derived from tl inside the trip.  Hoisted out of
        // the loop, the scatter addresses of the exchanges do not fit the 128 registers of a 1024-thread
        // workgroup, and a reload from scratch is a counted vector-memory load: it would wait for the
        // stores and the DMA in flight.
        int tl = t;
        if constexpr (DMA) asm volatile("" : "+v"(tl));

        // ---- load ----------------------------------------------------------------
        if constexpr (DMA) {
            const C* stage = reinterpret_cast<const C*>(smem_raw);
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = stage[tl + TX * e];
            if constexpr (WARM) {
                if (g + g_step < g_end) dma_warm(g + g_step);
            }
            if constexpr (FASTWIN) {
#pragma unroll
                for (int e = 0; e < P; ++e) { v[e].x *= wreg[e]; v[e].y *= wreg[e]; }
            }
            MFFT_STAMP(0);
        } else if constexpr (!GENERAL) {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = vn[e];
                // the group after this one (re-reads this one at the end of the batch: no branch)
                const int64_t gn = g + g_step < g_end ? g + g_step : g;
                load_fast(vn, gn * SLOTS + slot);
            } else {
                load_fast(v, b);
            }
            if constexpr (FASTWIN) {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    T w;
                    if constexpr (WINSRC == WIN_REG) w = wreg[e];
                    else if constexpr (WINSRC == WIN_BUF) w = buf_ld<T>(make_rsrc(a.win, (uint32_t)N * sizeof(T)), (uint32_t)t * sizeof(T), TX * e * sizeof(T));
                    else w = a.win[t + TX * e];
                    v[e].x *= w;
                    v[e].y *= w;
                }
            }
        } else if constexpr (TX >= 64) {
            // Runtime prologue, signals owned by whole waves: range-checked buffer loads.
            // (The general variants take dense signals only; segmented layouts that V_SEG does
            // not serve are staged through a dense buffer by the plan.)
            const int64_t bl = active ? b : a.batch - 1;
            const uint32_t tt = (uint32_t)t;
            if (a.real_in) {
                const auto rs = make_rsrc(reinterpret_cast<const T*>(a.in) + bl * a.in_stride, (uint32_t)a.n_in * sizeof(T));
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = C{buf_ld<T>(rs, tt * sizeof(T), TX * e * sizeof(T)), 0};
            } else {
                const auto rs = make_rsrc(reinterpret_cast<const C*>(a.in) + bl * a.in_stride, (uint32_t)a.n_in * sizeof(C));
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = buf_ld<C>(rs, tt * sizeof(C), TX * e * sizeof(C));
            }
            if (a.conj_in) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e].y = -v[e].y;
            }
            if (a.win) {
                const auto rw = make_rsrc(a.win, (uint32_t)a.n_in * sizeof(T));
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    const T w = buf_ld<T>(rw, tt * sizeof(T), TX * e * sizeof(T));
                    v[e].x *= w; v[e].y *= w;
                }
            }
            if (a.pre) {
                const auto rp = make_rsrc(a.pre, (uint32_t)a.n_in * sizeof(C));
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = cmul(v[e], buf_ld<C>(rp, tt * sizeof(C), TX * e * sizeof(C)));
            }
        } else {
            // Runtime prologue, several signals per wave.  Every load is issued unconditionally at
            // a clamped index and the options are uniform branches around whole loops, so the loads
            // of one signal stay in flight together (a per-element branch would serialise them).
            const int64_t bl = active ? b : a.batch - 1;
            const int last = a.n_in - 1;
            int jj[P];
#pragma unroll
            for (int e = 0; e < P; ++e) { const int j = t + TX * e; jj[e] = j < last ? j : last; }
            // the indices do not depend on the signal; computed outside the batch loop, the element
            // addresses of x, window and premultiplier stay live in ~100 registers, which at 512
            // points leaves one resident workgroup per CU: keep them per-signal work there
            // (+12..33 %; the shorter lengths measured 2-5 % slower with it)
            if constexpr (N == 512) {
#pragma unroll
                for (int e = 0; e < P; ++e) asm volatile("" : "+v"(jj[e]));
            }
            if (a.real_in) {
                const T* __restrict__ src = reinterpret_cast<const T*>(a.in) + bl * a.in_stride;
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = C{src[jj[e]], 0};
            } else {
                const C* __restrict__ src = reinterpret_cast<const C*>(a.in) + bl * a.in_stride;
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = src[jj[e]];
            }
            if (a.conj_in) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e].y = -v[e].y;
            }
            if (a.win) {
#pragma unroll
                for (int e = 0; e < P; ++e) { const T w = a.win[jj[e]]; v[e].x *= w; v[e].y *= w; }
            }
            if (a.pre) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = cmul(v[e], a.pre[jj[e]]);
            }
            if (a.n_in < N) {          // zero padding: samples beyond n_in
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    if (t + TX * e > last) v[e] = C{0, 0};
                }
            }
        }

        // ---- passes ---------------------------------------------------------
        // the exchange buffer is idle from the last exchange on: the workgroup's next signal goes there
        const auto dma_next = [&]() {
            lds_barrier();                                  // every wave has read its exchange results
            if (g + g_step < g_end) dma_signal(g + g_step);
        };
#if !MFFT_DIAG_NOFFT
        const auto exchange = [&](auto pass) {
            constexpr int S = decltype(pass)::value;
            pass_exchange<T, RD, S, P, TX, PADSHIFT, DMA>(v, tl, lds);
            if constexpr (DMA && S == NPASS - 2) dma_next();
        };
        pass_compute<T, RD, 0, P, TX, TWSRC, VEC>(v, tl, a.tw, twr, twl);
        MFFT_STAMP(1);
        if constexpr (NPASS > 1) {
            exchange(std::integral_constant<int, 0>{});
            MFFT_STAMP(2);
            pass_compute<T, RD, 1, P, TX, TWSRC, VEC>(v, tl, a.tw, twr, twl);
            MFFT_STAMP(3);
        }
        if constexpr (NPASS > 2) {
            exchange(std::integral_constant<int, 1>{});
            MFFT_STAMP(4);
            pass_compute<T, RD, 2, P, TX, TWSRC, VEC>(v, tl, a.tw, twr, twl);
            MFFT_STAMP(5);
        }
        if constexpr (NPASS > 3) {
            exchange(std::integral_constant<int, 2>{});
            MFFT_STAMP(6);
            pass_compute<T, RD, 3, P, TX, TWSRC, VEC>(v, tl, a.tw, twr, twl);
            MFFT_STAMP(7);
        }
#else
        if constexpr (DMA) dma_next();
#endif

        // ---- store: output k = t + TX*e sits in v[e] --------------------------
        if constexpr (POWER) {
            T* __restrict__ dst = reinterpret_cast<T*>(a.out) + b * a.out_stride;
            T s[P];
            if constexpr (GENERAL) {
                if (a.post) {                  // |.|^2 ignores conj_out
                    if constexpr (TX >= 64) {
                        const auto rp = make_rsrc(a.post, (uint32_t)a.n_out * sizeof(C));
#pragma unroll
                        for (int e = 0; e < P; ++e)
                            v[e] = cmul(v[e], buf_ld<C>(rp, (uint32_t)t * sizeof(C), TX * e * sizeof(C)));
                    } else {
#pragma unroll
                        for (int e = 0; e < P; ++e) {
                            const int k = t + TX * e;
                            v[e] = cmul(v[e], a.post[k < a.n_out ? k : a.n_out - 1]);
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const T re = v[e].x * a.scale, im = v[e].y * a.scale;
                s[e] = re * re + im * im;
            }
            if constexpr (GENERAL && TX >= 64) {
                // an inactive slot gets an empty extent: every store is dropped by the range check
                const auto ro = make_rsrc(dst, active ? (uint32_t)a.n_out * sizeof(T) : 0u);
#pragma unroll
                for (int e = 0; e < P; ++e) buf_st<T>(ro, (uint32_t)t * sizeof(T), TX * e * sizeof(T), s[e]);
            } else if (active) {
                if constexpr (GENERAL) {
#pragma unroll
                    for (int e = 0; e < P; ++e) {
                        if (t + TX * e < a.n_out) dst[t + TX * e] = s[e];
                    }
                } else if constexpr (DMA) {
                    const auto ro = make_rsrc_uniform(dst, (uint32_t)N * sizeof(T));
#pragma unroll
                    for (int e = 0; e < P; ++e) buf_st<T, MFFT_NT ? 2 : 0>(ro, (uint32_t)tl * (uint32_t)sizeof(T), TX * e * sizeof(T), s[e]);
                } else if constexpr (MFFT_FAST_BUF != 0 && TX >= 64) {
                    const auto ro = make_rsrc(dst, (uint32_t)N * sizeof(T));
#pragma unroll
                    for (int e = 0; e < P; ++e) buf_st<T, MFFT_NT ? 2 : 0>(ro, (uint32_t)t * sizeof(T), TX * e * sizeof(T), s[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) st_stream(dst + t + TX * e, s[e]);
                }
            }
            if constexpr (VAR == V_GEN_POWER || FASTSTATS) {
                if (a.stats != nullptr) {
                    double mx = 0.0, sm = 0.0, sk = 0.0;
                    if constexpr (FASTSTATS && MFFT_STATS_CLOSED != 0) {
                        // all N outputs are kept: output k = t + TX*e has frequency index k for e < P/2 and
                        // k - N = t + TX*(e - P) beyond, so sum f*s = t * sum s + TX * sum e'*s with
                        // compile-time weights e' (three instructions per point instead of eight).
                        // complex64 accumulates a thread's own P values in single precision (the spectrum
                        // itself is single precision: P more roundings of 6e-8 change nothing).
                        using A = std::conditional_t<sizeof(T) == 4 && MFFT_STATS_F32 != 0, T, double>;
                        A mxa = 0, sma = 0, ska = 0;
#pragma unroll
                        for (int e = 0; e < P; ++e) {
                            const A sv = (A)s[e];
                            mxa = sv > mxa ? sv : mxa;
                            sma += sv;
                            ska += (A)(e < P / 2 ? e : e - P) * sv;
                        }
                        if (active) {
                            mx = (double)mxa; sm = (double)sma;
                            sk = (double)t * (double)sma + (double)TX * (double)ska;
                        }
                    } else if constexpr (sizeof(T) == 4 && MFFT_STATS_F32 != 0) {
                        // a thread's own P values are accumulated in single precision (the spectrum itself is
                        // single precision: P more roundings of 6e-8 change nothing), lanes and waves in double
                        T mxf = 0, smf = 0, skf = 0;
#pragma unroll
                        for (int e = 0; e < P; ++e) {
                            const int k = t + TX * e;
                            const T f = (k < (a.n_out + 1) / 2) ? (T)k : (T)(k - a.n_out);
                            const T sv = (active && k < a.n_out) ? s[e] : (T)0;
                            mxf = sv > mxf ? sv : mxf;
                            smf += sv;
                            skf += f * sv;
                        }
                        mx = mxf; sm = smf; sk = skf;
                    } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) {
                        const int k = t + TX * e;
                        const double f = (k < (a.n_out + 1) / 2) ? (double)k : (double)(k - a.n_out);
                        const double sv = (active && k < a.n_out) ? (double)s[e] : 0.0;
                        mx = sv > mx ? sv : mx;
                        sm += sv;
                        sk += f * sv;
                    }
                    }
                    constexpr int W = TX < 64 ? TX : 64;
#pragma unroll
                    for (int o = W / 2; o > 0; o >>= 1) {
                        const double omx = __shfl_xor(mx, o, 64);
                        mx = omx > mx ? omx : mx;
                        sm += __shfl_xor(sm, o, 64);
                        sk += __shfl_xor(sk, o, 64);
                    }
                    if constexpr (DMA && MFFT_DMA_STATS_LDS != 0) {
                        // one signal per workgroup: the waves' partial results meet in LDS after the
                        // barrier that ends the trip, and wave 0 stores the three numbers.  Atomics here
                        // would be vector-memory operations issued behind the stores, so the counted wait
                        // would have to let three more stores drain, and they retire slowly (16 waves on
                        // three addresses); the sum also becomes independent of the waves' arrival order.
                        if ((threadIdx.x & 63u) == 0) {
                            double* red = reinterpret_cast<double*>(smem_raw + DMA_STATS_AT) + 3 * wv;
                            red[0] = mx; red[1] = sm; red[2] = sk;
                        }
                    } else if (active && (t % W) == 0) {
                        double* st = a.stats + 3 * b;
                        if constexpr (TX <= 64) {
                            // the whole signal was reduced inside one wave: plain stores
                            st[0] = mx; st[1] = sm; st[2] = sk;
                        } else {
                            atomicMax(reinterpret_cast<unsigned long long*>(st),
                                      (unsigned long long)__double_as_longlong(mx));
                            atomicAdd(st + 1, sm);
                            atomicAdd(st + 2, sk);
                        }
                    }
                }
            }
        } else {
            C* __restrict__ dst = reinterpret_cast<C*>(a.out) + b * a.out_stride;
#pragma unroll
            for (int e = 0; e < P; ++e) { v[e].x *= a.scale; v[e].y *= a.scale; }
            if constexpr (GENERAL) {
                if (a.conj_out) {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e].y = -v[e].y;
                }
                if (a.post) {
                    if constexpr (TX >= 64) {
                        const auto rp = make_rsrc(a.post, (uint32_t)a.n_out * sizeof(C));
#pragma unroll
                        for (int e = 0; e < P; ++e)
                            v[e] = cmul(v[e], buf_ld<C>(rp, (uint32_t)t * sizeof(C), TX * e * sizeof(C)));
                    } else {
#pragma unroll
                        for (int e = 0; e < P; ++e) {
                            const int k = t + TX * e;
                            v[e] = cmul(v[e], a.post[k < a.n_out ? k : a.n_out - 1]);
                        }
                    }
                }
            }
            if constexpr (GENERAL && TX >= 64) {
                const auto ro = make_rsrc(dst, active ? (uint32_t)a.n_out * sizeof(C) : 0u);
#pragma unroll
                for (int e = 0; e < P; ++e) buf_st<C>(ro, (uint32_t)t * sizeof(C), TX * e * sizeof(C), v[e]);
            } else if (active) {
                if constexpr (GENERAL) {
#pragma unroll
                    for (int e = 0; e < P; ++e) {
                        if (t + TX * e < a.n_out) dst[t + TX * e] = v[e];
                    }
                } else if constexpr (DMA) {
                    // buffer stores: base and row offset live in scalar registers, one lane-dependent
                    // offset register serves all P stores and no vector instruction is spent on addresses
                    const uint32_t voff = (uint32_t)tl * (uint32_t)sizeof(C);
                    if constexpr (VAR == V_SEG) {
                        // lane part of the position as the 32-bit buffer offset (launch_one checks that a row's
                        // segments span less than 4 GiB), the e part in the descriptor's base
                        const uint32_t vseg = (uint32_t)seg_pos(tl, a.out_lseg, a.out_seg_stride) * (uint32_t)sizeof(C);
                        const uint32_t span = ((uint32_t)seg_pos(TX - 1, a.out_lseg, a.out_seg_stride) + 1u) * (uint32_t)sizeof(C);
#pragma unroll
                        for (int e = 0; e < P; ++e)
                            buf_st<C, MFFT_NT ? 2 : 0>(make_rsrc_uniform(dst + seg_pos(TX * e, a.out_lseg, a.out_seg_stride), span), vseg, 0u, v[e]);
                    } else {
                        const auto ro = make_rsrc_uniform(dst, (uint32_t)N * sizeof(C));
#pragma unroll
                        for (int e = 0; e < P; ++e) buf_st<C, MFFT_NT ? 2 : 0>(ro, voff, TX * e * sizeof(C), v[e]);
                    }
                } else if constexpr (VAR == V_SEG) {
                    C* __restrict__ lane = dst + seg_pos(tl, a.out_lseg, a.out_seg_stride);
#pragma unroll
                    for (int e = 0; e < P; ++e) st_stream(lane + seg_pos(TX * e, a.out_lseg, a.out_seg_stride), v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) st_stream(dst + tl + TX * e, v[e]);
                }
            }
        }
        if constexpr (DMA) {
            // The next signal has landed once everything issued before this trip's P stores has
            // retired (one store instruction per point held; `active` is always true here).
            MFFT_STAMP(8);
            wait_vm<P>();
            lds_barrier();
            MFFT_STAMP(9);
            if constexpr (FASTSTATS && MFFT_DMA_STATS_LDS != 0) {
                if (a.stats != nullptr && wv == 0) {
                    const int lane = (int)(threadIdx.x & 63u);
                    const double* red = reinterpret_cast<const double*>(smem_raw + DMA_STATS_AT) + 3 * (lane < NW ? lane : 0);
                    double mx = lane < NW ? red[0] : 0.0, sm = lane < NW ? red[1] : 0.0, sk = lane < NW ? red[2] : 0.0;
#pragma unroll
                    for (int o = NW / 2; o > 0; o >>= 1) {
                        const double omx = __shfl_xor(mx, o, 64);
                        mx = omx > mx ? omx : mx;
                        sm += __shfl_xor(sm, o, 64);
                        sk += __shfl_xor(sk, o, 64);
                    }
                    if (lane == 0) {
                        double* st = a.stats + 3 * b;
                        st[0] = mx; st[1] = sm; st[2] = sk;
                    }
                }
            }
        }
    }
#if MFFT_DIAG_STAMPS
    if constexpr (!DMA) {
        // per workgroup: start and end in s_memtime ticks (the counters of different XCDs are not aligned: only
        // differences inside one workgroup mean anything), XCC_ID and HW_ID of wave 0
        if (threadIdx.x == 0 && blockIdx.x < 2048) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_diag_stamps[blockIdx.x * 16 + 0] = diag_t0;
            g_diag_stamps[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memtime();
            g_diag_stamps[blockIdx.x * 16 + 2] = xcc;
            g_diag_stamps[blockIdx.x * 16 + 3] = hw;
        }
    }
    if constexpr (DMA) {
        if (threadIdx.x == 0 && blockIdx.x < 2048) {
            for (int k = 0; k < 16; ++k) g_diag_stamps[blockIdx.x * 16 + k] = stamp_acc[k];
        }
    }
#endif
}

// One resident set of workgroups, or several, all striding over the batch.
template <typename T, int P, class RD, int PADSHIFT, int BLOCK, int VAR, bool DMA = false>
__global__ void MFFT_LAUNCH_BOUNDS(BLOCK, T, RD::n, VAR) fft_lds_kernel(const KArgs<T> a)
{
    constexpr int SLOTS = BLOCK / (RD::n / P);
    lds_body<T, P, RD, PADSHIFT, BLOCK, VAR, DMA>(a, (int64_t)blockIdx.x, (a.batch + SLOTS - 1) / SLOTS, gridDim.x);
}

// The same with a tail: the first a.resident workgroups stride over the groups below a.tail_start, every
// further workgroup takes a.tail_chunk consecutive groups of the rest.  Those are dispatched as the
// striding workgroups of the faster CUs end, so the last part of the batch goes to whoever is free, while
// the bulk pays for tables and window registers once per resident workgroup.  (Two copies of the body:
// one loop with a selected start, end and step schedules measurably worse than the plain striding loop.)
template <typename T, int P, class RD, int PADSHIFT, int BLOCK, int VAR>
__global__ void MFFT_LAUNCH_BOUNDS(BLOCK, T, RD::n, VAR) fft_lds_tail_kernel(const KArgs<T> a)
{
    constexpr int SLOTS = BLOCK / (RD::n / P);
    const unsigned res = (unsigned)a.resident;
    if (blockIdx.x < res) {
        lds_body<T, P, RD, PADSHIFT, BLOCK, VAR, false>(a, (int64_t)blockIdx.x, a.tail_start, res);
    } else {
        const int64_t groups = (a.batch + SLOTS - 1) / SLOTS;
        const int64_t g0 = a.tail_start + (int64_t)(blockIdx.x - res) * a.tail_chunk;
        lds_body<T, P, RD, PADSHIFT, BLOCK, VAR, false>(a, g0, g0 + a.tail_chunk < groups ? g0 + a.tail_chunk : groups, 1u);
    }
}

// ---------------------------------------------------------------------------
// Size table: P points per thread and the pass radices for each 2^k.
// ---------------------------------------------------------------------------
template <int LOG2N> struct SizeCfg;
template <> struct SizeCfg<1>  { static constexpr int P = 2;  using RD = Radices<2, 1, 1, 1>; };
template <> struct SizeCfg<2>  { static constexpr int P = 4;  using RD = Radices<4, 1, 1, 1>; };
template <> struct SizeCfg<3>  { static constexpr int P = 8;  using RD = Radices<8, 1, 1, 1>; };
template <> struct SizeCfg<4>  { static constexpr int P = 4;  using RD = Radices<4, 4, 1, 1>; };
template <> struct SizeCfg<5>  { static constexpr int P = 8;  using RD = Radices<4, 8, 1, 1>; };
template <> struct SizeCfg<6>  { static constexpr int P = 8;  using RD = Radices<8, 8, 1, 1>; };
template <> struct SizeCfg<7>  { static constexpr int P = 16; using RD = Radices<8, 16, 1, 1>; };
template <> struct SizeCfg<8>  { static constexpr int P = 16; using RD = Radices<16, 16, 1, 1>; };
template <> struct SizeCfg<9>  { static constexpr int P = 16; using RD = Radices<2, 16, 16, 1>; };
// 1024 keeps all 15 twiddles of its last pass in registers: on the hann + power batch that is
// 2 % faster than the six-generator form (134.6 vs 137.6 us per 65536 signals)
template <> struct SizeCfg<10> { static constexpr int P = 16; using RD = Radices<MFFT_R1024, false>; };
template <> struct SizeCfg<11> { static constexpr int P = 16; using RD = Radices<8, 16, 16, 1>; };
template <> struct SizeCfg<12> { static constexpr int P = 16; using RD = Radices<16, 16, 16, 1>; };
template <> struct SizeCfg<13> { static constexpr int P = 16; using RD = Radices<2, 16, 16, 16>; };
// 16384 points: 16 per thread in a 1024-thread workgroup (4*16*16*16).  (32 per thread in a 512-thread one,
// with radix-32 first and last passes or with radix-16 butterflies only, measured slower where it matters:
// plain 4.13 against 4.44 TB/s, windowed 2.4 against 3.9.)
template <> struct SizeCfg<14> { static constexpr int P = 16; using RD = Radices<4, 16, 16, 16>; };

// Per-precision exceptions.  8192 complex64 points: 32 points per thread and a radix-32 first pass.
// With 16 points a signal needs a 512-thread workgroup and three LDS exchanges (2*16*16*16); the
// same lengths measured 20-25 % slower in 512-thread workgroups than in 256-thread ones (4096
// points: 4.2 against 5.5 TB/s), the price of barriers over eight waves.  32 points per thread is a
// 256-thread workgroup, two exchanges (32*16*16) and 234-252 VGPRs, two workgroups per CU:
// plain 4.25 -> 4.6 TB/s, windowed 3.7 -> 4.6, Hann + power 3.55 -> 4.25.  complex128 (128 KiB of
// LDS per signal: one workgroup per CU either way) measured the same or slower and keeps 16, and
// so does 4096 complex64 (32 per thread: 5.26 against 5.49 TB/s plain, 3.5 against 5.0 windowed).
template <typename T, int LOG2N> struct SizeFor : SizeCfg<LOG2N> {};
#if MFFT_P8192 == 32
template <> struct SizeFor<float, 13> { static constexpr int P = 32; using RD = Radices<32, 16, 16, 1>; };
#endif

constexpr int kPadShift = 4;   // one spare element per 16: conflict-free ds_write_b64 scatter

// Points per thread of a variant.  The runtime prologue / epilogue variants of the 16384-point kernel
// hold 32 points per thread in a 512-thread workgroup: at 16 points x 1024 threads (128 VGPRs) they
// spill, 3.0 TB/s against 3.9; the fast variants measured the other way round (4.5 against 3.8).
template <typename T, int LOG2N, int VAR> constexpr int points_per_thread()
{
    return (LOG2N == 14 && (VAR == V_GEN_CPLX || VAR == V_GEN_POWER)) ? 32 : SizeFor<T, LOG2N>::P;
}

template <typename T, int LOG2N, int VAR = V_PLAIN> struct BlockCfg {
    static constexpr int TX = (1 << LOG2N) / points_per_thread<T, LOG2N, VAR>();
    // a two-wave signal (2048 points) gets a workgroup of its own: its barriers then span two
    // waves instead of four (+4 % plain complex64); one-wave signals need no barrier and share
    // (the 1024-point complex64 Hann + power kernels: one-wave workgroups, so that the dispatched tail is handed out
    // signal by signal)
    static constexpr bool ONE_WAVE = sizeof(T) == 4 && LOG2N == 10 && (VAR == V_HANN_POWER || VAR == V_HANN_POWER_STATS);
    static constexpr int BLOCK = ONE_WAVE ? MFFT_BLOCK_1024_HANN
                               : (TX == 128 && MFFT_BLOCK_2048 != 0) ? 128 : (TX > MFFT_BLOCK_MIN ? TX : MFFT_BLOCK_MIN);
};

template <typename T, int LOG2N, int VAR = V_PLAIN> constexpr size_t lds_bytes_for()
{
    using SC = SizeFor<T, LOG2N>;
    constexpr int N = 1 << LOG2N;
    constexpr int TX = BlockCfg<T, LOG2N, VAR>::TX, BLOCK = BlockCfg<T, LOG2N, VAR>::BLOCK;
    constexpr int SLOTS = BLOCK / TX;
    constexpr int LDS_ELEMS = ((lds_pad<kPadShift>(N - 1) + 1 + 1) / 2) * 2;
    return (size_t)SLOTS * LDS_ELEMS * sizeof(cx<T>) +
           table_lds_bytes<T, typename SC::RD, TX, points_per_thread<T, LOG2N, VAR>(), VAR>();
}

// persistent grid: resident workgroups per CU (queried once per kernel) x CUs, capped by the work
int blocks_per_cu(const void* kernel, int block, size_t lds);
int pick_grid(int per_cu, int64_t groups, int over = 1);

// Resident sets of workgroups a launch asks for.  One set (a persistent grid: every workgroup strides over
// the batch and pays for its tables once) leaves the CUs' different speeds uncorrected -- per-CU durations
// of the same 32 trips spread by +-5 % -- while workgroups that each take a few trips are handed out by the
// dispatcher as CUs become free.  What that gains against the per-workgroup start-up (tables, window
// registers, the drain of the last stores) was measured per length and variant on 2^26 (complex64) /
// 2^25 (complex128) points per launch, factors 1, 2, 4, 8, 16: the light variants gain 5..12 % at 8..16
// sets, the windowed power variants from 512 points (window samples and twiddles in registers: the
// heaviest start-up, two workgroups per CU) lose 3..15 % and keep one set; the LDS-DMA kernels lose from four.
// Columns: log2 of the length.
template <typename T, int LOG2N, int VAR, bool DMA> constexpr int lds_grid_over()
{
    // LDS-DMA kernels: two sets where the output is complex (8192 points +3..7 %, 16384 points plain +1..2.5 %);
    // with power output, and windowed at 16384 points, two sets measured -2.4..+2.3 % from run to run: one
    if (DMA) return (VAR == V_PLAIN || VAR == V_SEG || (VAR == V_WIN && LOG2N < 14)) ? 2 : 1;
    constexpr int L = LOG2N > 14 ? 14 : LOG2N;
    //                                0   1   2   3   4   5   6   7   8   9  10  11  12  13  14
    constexpr signed char c64[V_COUNT][15] = {
        /* V_PLAIN            */ {16, 16, 16, 16, 16, 16, 16, 16, 16, 16,  8, 16,  8,  1,  1},
        /* V_HANN_POWER       */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  1,  1,  8,  1,  1,  1},
        /* V_GEN_CPLX         */ { 4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  1,  1},
        /* V_GEN_POWER        */ { 4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  1,  1},
        /* V_SEG              */ { 1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1},
        /* V_WIN              */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  8,  8,  8,  1,  1,  1},
        /* V_POW              */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  8,  8,  8,  4,  1,  1},
        /* V_REAL             */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  8,  8, 16,  8,  1,  1},
        /* V_HANN_POWER_STATS */ {16, 16, 16, 16, 16,  1,  4,  8,  8,  1,  1,  1,  1,  1,  1},
        /* V_POW_STATS        */ {16, 16, 16, 16, 16,  8,  8, 16, 16,  4,  8,  8,  4,  1,  1}};
    constexpr signed char c128[V_COUNT][15] = {
        /* V_PLAIN            */ {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16,  8, 16,  1,  1},
        /* V_HANN_POWER       */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  1,  2,  1,  1,  1,  1},
        /* V_GEN_CPLX         */ { 8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  1,  1},
        /* V_GEN_POWER        */ { 8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  8,  1,  1},
        /* V_SEG              */ { 1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1},
        /* V_WIN              */ {16, 16, 16, 16, 16, 16, 16, 16, 16,  4,  8,  8,  4,  1,  1},
        /* V_POW              */ { 8,  8,  8,  8,  8,  8,  8,  8, 16, 16,  2, 16,  2,  1,  1},
        /* V_REAL             */ { 4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  4,  1,  1},
        /* V_HANN_POWER_STATS */ { 8,  8,  8,  8,  8,  8,  8, 16, 16,  4,  2,  1,  1,  1,  1},
        /* V_POW_STATS        */ { 8,  8,  8,  8,  8, 16,  8, 16, 16, 16,  1,  1,  1,  1,  1}};
    return sizeof(T) == 4 ? c64[VAR][L] : c128[VAR][L];
}

// Which kernels are also built with a tail (fft_lds_tail_kernel), the share of the batch it gets (per cent)
// and the groups per tail workgroup.
template <typename T, int LOG2N, int VAR> constexpr bool lds_tail_built()
{
    return LOG2N >= (sizeof(T) == 4 ? 9 : 11) && LOG2N <= 12 && (VAR == V_HANN_POWER || VAR == V_HANN_POWER_STATS);
}
// (measured, tail share in per cent / groups per tail workgroup against none, 2^26 points per launch: 512 points
// 2/1 136.5 -> 134.4 us, 1024 points 2/1 .. 4/1 135.6 -> 134.2, 2048 points 16/2 141.3 -> 138.3, 4096 points
// 10/2 147.7 -> 145.1 and with statistics 163.4 -> 157.8; larger shares lose to the tail workgroups' start-up)
// complex128 (2^25 points per launch): 2048 points 8/1 142.8 -> 139.2, 4096 points 4/1 149.1 -> 145.2; none at 512 / 1024 (-0.5..-4 %)
template <typename T, int LOG2N, int VAR> constexpr int lds_tail_pct()
{
    if (sizeof(T) == 8) return LOG2N == 11 ? 8 : 4;
    return LOG2N == 9 ? 2 : LOG2N == 10 ? 6 : LOG2N == 11 ? 16 : 10;
}
// (1024 complex64 points run in one-wave workgroups, a group is one signal: 6 % in pairs 134.7 -> 131.7 us against
// 133.2 for the best tail of 256-thread workgroups; with statistics 6 % in fours 145.7 -> 143.3 against 144.5)
template <typename T, int LOG2N, int VAR> constexpr int lds_tail_chunk()
{
    if (sizeof(T) == 4 && LOG2N == 10) return VAR == V_HANN_POWER_STATS ? 4 : 2;
    return sizeof(T) == 4 && LOG2N >= 11 ? 2 : 1;
}

// The LDS-DMA pipeline moves 16 bytes per lane: signals (and segments) must start on 16-byte boundaries.
// MFFT_NO_DMA=1 in the environment selects the register-load form (A/B timing, tests).
template <typename T, int VAR> bool dma_usable(const KArgs<T>& a)
{
    const size_t es = sizeof(cx<T>);
    if ((reinterpret_cast<size_t>(a.in) & 15) || ((size_t)a.in_stride * es & 15)) return false;
    if (VAR == V_SEG && ((size_t)a.in_seg_stride * es & 15)) return false;
    const char* off = getenv("MFFT_NO_DMA");
    return !(off && off[0] == '1');
}

template <typename T, int LOG2N, int VAR, bool DMA = false>
int launch_one(const KArgs<T>& a, int grid, hipStream_t s)
{
    using SC = SizeFor<T, LOG2N>;
    constexpr int BLOCK = BlockCfg<T, LOG2N, VAR>::BLOCK;
    constexpr int SLOTS = BLOCK / BlockCfg<T, LOG2N, VAR>::TX;
    if constexpr (!DMA && dma_default<T, (1 << LOG2N), VAR>()) {
        // (segmented stores of the pipeline address a row's segments through one 32-bit offset)
        constexpr int TXK = BlockCfg<T, LOG2N, VAR>::TX;
        const bool span_ok = VAR != V_SEG || ((uint64_t)(seg_pos(TXK - 1, a.out_lseg, a.out_seg_stride) + 1) * sizeof(cx<T>) < (uint64_t(1) << 32));
        if (span_ok && dma_usable<T, VAR>(a)) return launch_one<T, LOG2N, VAR, true>(a, grid, s);
    }
    constexpr int LDS_ELEMS = ((lds_pad<kPadShift>((1 << LOG2N) - 1) + 1 + 1) / 2) * 2;
    constexpr size_t lds = DMA ? ((size_t)SLOTS * LDS_ELEMS + twc_total_elems<typename SC::RD>()) * sizeof(cx<T>) + (MFFT_DMA_WARM != 0 ? BLOCK * 4 : 0) + (BLOCK / 64) * 24
                               : ((SC::RD::npass > 1) ? lds_bytes_for<T, LOG2N, VAR>() : 0);
    static_assert(lds <= 160 * 1024, "LDS of one workgroup");
    auto kern = fft_lds_kernel<T, points_per_thread<T, LOG2N, VAR>(), typename SC::RD, kPadShift, BLOCK, VAR, DMA>;
    static thread_local int attr_dev = -1, per_cu = 1;
    int dev = 0;
    MFFT_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        MFFT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_cu = blocks_per_cu(reinterpret_cast<const void*>(kern), BLOCK, lds);
        attr_dev = dev;
    }
    const int64_t groups = (a.batch + SLOTS - 1) / SLOTS;
    if (groups <= 0) return MFFT_OK;
    if (grid <= 0) grid = pick_grid(per_cu, groups, a.grid_over > 0 ? a.grid_over : lds_grid_over<T, LOG2N, VAR, DMA>());
    if constexpr (!DMA && lds_tail_built<T, LOG2N, VAR>()) {
        // striding bulk + tail handed out by the dispatcher
        const int pct = a.tail_pct > 0 ? a.tail_pct : a.tail_pct == 0 ? lds_tail_pct<T, LOG2N, VAR>() : 0;
        const int chunk = a.tail_chunk_req > 0 ? a.tail_chunk_req : lds_tail_chunk<T, LOG2N, VAR>();
        auto tkern = fft_lds_tail_kernel<T, points_per_thread<T, LOG2N, VAR>(), typename SC::RD, kPadShift, BLOCK, VAR>;
        static thread_local int tattr_dev = -1, tper_cu = 1;
        if (tattr_dev != dev) {
            MFFT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tkern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            tper_cu = blocks_per_cu(reinterpret_cast<const void*>(tkern), BLOCK, lds);      // the striding set must be resident at once
            tattr_dev = dev;
        }
        const int64_t res = pick_grid(tper_cu, groups, 1);
        const int64_t tail = groups * pct / 100;
        if (pct > 0 && a.grid_over <= 0 && res < groups && tail > 0 && res + (tail + chunk - 1) / chunk < (int64_t(1) << 31)) {
            KArgs<T> h = a;
            h.resident = (int)res; h.tail_chunk = chunk; h.tail_start = groups - tail;
            hipLaunchKernelGGL(tkern, dim3((unsigned)(res + (tail + chunk - 1) / chunk)), dim3(BLOCK), lds, s, h);
            MFFT_HIP(hipGetLastError());
            return MFFT_OK;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, a);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

template <typename T, int LOG2N>
int launch_variant(int variant, const KArgs<T>& a, int grid, hipStream_t s)
{
    switch (variant) {
    case V_PLAIN: return launch_one<T, LOG2N, V_PLAIN>(a, grid, s);
    case V_HANN_POWER: return launch_one<T, LOG2N, V_HANN_POWER>(a, grid, s);
    case V_WIN: return launch_one<T, LOG2N, V_WIN>(a, grid, s);
    case V_POW: return launch_one<T, LOG2N, V_POW>(a, grid, s);
    case V_REAL: return launch_one<T, LOG2N, V_REAL>(a, grid, s);
    case V_HANN_POWER_STATS: return launch_one<T, LOG2N, V_HANN_POWER_STATS>(a, grid, s);
    case V_POW_STATS: return launch_one<T, LOG2N, V_POW_STATS>(a, grid, s);
    case V_GEN_CPLX: return launch_one<T, LOG2N, V_GEN_CPLX>(a, grid, s);
    case V_GEN_POWER: return launch_one<T, LOG2N, V_GEN_POWER>(a, grid, s);
    case V_SEG: return launch_one<T, LOG2N, V_SEG>(a, grid, s);
    }
    set_error("unknown kernel variant");
    return MFFT_E_INVALID;
}

template <typename T, int MAXLOG>
int launch_lds_any(int log2n, int variant, const KArgs<T>& a, int grid, hipStream_t s)
{
    switch (log2n) {
#define MFFT_CASE(L) case L: if constexpr (L <= MAXLOG) return launch_variant<T, L>(variant, a, grid, s); else break;
        MFFT_CASE(1) MFFT_CASE(2) MFFT_CASE(3) MFFT_CASE(4) MFFT_CASE(5) MFFT_CASE(6) MFFT_CASE(7)
        MFFT_CASE(8) MFFT_CASE(9) MFFT_CASE(10) MFFT_CASE(11) MFFT_CASE(12) MFFT_CASE(13) MFFT_CASE(14)
#undef MFFT_CASE
    }
    set_error("length not served by the single-kernel path");
    return MFFT_E_UNSUPPORTED;
}

}  // namespace mfft
