// One global Stockham pass for transforms too long for a single LDS residency.
//
// A transform of N = G_0 * G_1 * ... points runs as a few out-of-place passes over
// HBM, each the autosort step of radix G_s ("super-butterflies" of G_s points):
//
//   super-butterfly j in [0, L), L = N / G:
//     reads   x[j + i*L]                                    i < G     (row stride L)
//     twiddle x_i *= exp(-2*pi*i * (j mod NS)*i / (NS*G))              (NS = G_0*...*G_{s-1})
//     G-point DFT
//     writes  y[(j / NS)*NS*G + (j mod NS) + q*NS]          q < G     (row stride NS)
//
// so the output of the last pass is in natural order and no transpose or
// bit-reversal pass exists.  A workgroup owns C consecutive super-butterflies:
// its input is a G x C tile whose rows are C contiguous elements (C*8 B = one
// 128-B line for complex64 at C = 16), its output a G x C tile with the same
// property (for the first pass, where NS = 1, the tile is one contiguous run of
// C*G elements).  The G-point DFTs of the C columns run in LDS with the same
// register butterflies and Stockham exchanges as the batched kernel; the LDS image
// is [position][column] with an odd pitch so that both "lanes along a row" and
// "lanes down a column" are conflict-free.
//
// The twiddle exp(-2*pi*i*k/M), k = (j mod NS)*i < M = NS*G, is built per thread from four
// table reads: with m = j mod NS and rows i = t + TX*e, w_e = W^(m*t) * (W^(m*TX))^e.  Base and
// step each come from a two-level double-precision table (low 2^LB entries, high M/2^LB
// entries, both L2-resident) and the recurrence runs in double for both precisions, so its
// rounding stays far below the complex64 / complex128 tolerances.
#pragma once
#include "lds_kernel.h"

#ifndef MFFT_TILE_TW2
#define MFFT_TILE_TW2 1      // complex64 inter-pass twiddles as products of two factor sets (0 = a double-precision recurrence per element)
#endif
#ifndef MFFT_TILE_VEC
#define MFFT_TILE_VEC 1      // packed-form butterflies (fft_core.h) in the tile passes: +2..7 % on complex64 2^16..2^27 (0 = scalar form)
#endif

#ifndef MFFT_TILE_EVAL_ROOTS
#define MFFT_TILE_EVAL_ROOTS 2 // inter-pass roots evaluated per thread instead of read from the two-level tables: 0 only in the
                               // prefetching kernels, 1 in every tile kernel, 2 also in the complex64 512-point tiles
                               // (2^18: +7 %, 2^19: +4 %, 2^26: +5 %, 2^27: +11 %; all other kernels measured neutral)
#endif

#ifndef MFFT_TILE_PF_HANN
#define MFFT_TILE_PF_HANN 1 // the first pass of a Hann-windowed (and / or zero-padded) transform evaluates the window per thread
                            // (TM_FIRST_HANN) instead of reading its table -- so the prefetching tiles can prefetch there too
#endif

#ifndef MFFT_TILE_PF_GEN
#define MFFT_TILE_PF_GEN 0 // 1: so does their last pass with the general epilogue (scale, conjugate, table, segmented rows)
#endif

#ifndef MFFT_TILE_PF_POWER
#define MFFT_TILE_PF_POWER 1 // the power-spectrum last pass of the 512-point complex128 tiles prefetches too (252 registers, no
                             // scratch; 2^18: +5 %, 2^19: +2 %, 2^27: +6 %, with statistics +1.5-2 %; other lengths unchanged)
#endif

#ifndef MFFT_TILE_PREFETCH
#define MFFT_TILE_PREFETCH 1 // complex128 512-point tiles (one resident workgroup per CU): next tile's loads issued behind the first butterflies
#endif

namespace mfft {

// Plain or non-temporal tile accesses, chosen per pass at run time (uniform branch around the
// whole load / store loop).  Reads of a pass are dead once consumed while its writes feed the
// next pass, so when a transform fits the 256 MiB Infinity Cache, streaming the reads past the
// caches keeps the intermediate resident (2^24 complex64: 179 -> 158 us); for working sets far
// above the cache, and for two-pass lengths, plain loads measured faster.
template <typename T, bool NT> __device__ __forceinline__ cx<T> tile_ld(const cx<T>* p)
{
    if constexpr (NT) {
        using V = typename nt_vec<T>::type;
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
        return cx<T>{v.x, v.y};
    } else {
        return *p;
    }
}
template <typename T, bool NT> __device__ __forceinline__ void tile_st(cx<T>* p, cx<T> v)
{
    if constexpr (NT) {
        using V = typename nt_vec<T>::type;
        V w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, reinterpret_cast<V*>(p));
    } else {
        *p = v;
    }
}

template <typename T> struct TArgs {
    const void* in;
    void* out;
    const cx<T>* tw;      // inner per-pass tables for the G-point DFT
    const cx<double>* tw_lo;   // exp(-2*pi*i*k/M),        k < 2^lb      (double for both precisions)
    const cx<double>* tw_hi;   // exp(-2*pi*i*(k<<lb)/M),  k < M >> lb
    const T* win;         // FIRST: real window over the n_in input samples, or nullptr
    const cx<T>* pre;     // FIRST: complex premultiplier, or nullptr
    const cx<T>* post;    // LAST: complex postmultiplier over the n_out bins, or nullptr
    double* stats;        // LAST + power: 3 doubles per signal, or nullptr
    int64_t L;            // N / G: input row stride = number of super-butterflies
    int64_t NS;           // product of the radices of the earlier passes
    int64_t batch;
    int64_t in_stride;    // elements between signals in `in`
    int64_t out_stride;   // elements between signals in `out`
    int64_t out_pitch;    // LAST_PLAIN: elements between consecutive rows of a column-batched result (W = dense; 1 for ordinary signals)
    int64_t n_in;         // FIRST: samples present per signal
    int64_t n_out;        // LAST: bins stored per signal
    int64_t in_seg, in_seg_stride;     // FIRST: segmented input layout (0 = dense)
    int64_t out_seg, out_seg_stride;   // LAST: segmented output layout (0 = dense)
    // LAST_TWID: out(k, c) *= exp(-2*pi*i * k*(tw_c0 + c) / tw_n), k = row, c = column of the
    // (n x W) array (the four-step inter-step twiddle), evaluated per thread (tile_root)
    int64_t tw_c0;
    uint64_t fs_mask;     // tw_n - 1 (tw_n is a power of two)
    int lb;
    int log2_w;           // column-batched transforms: W = 1 << log2_w contiguous columns share every
                          // butterfly (NS counts elements, i.e. includes W); 0 for ordinary signals
    int log2_ns;          // NS = 1 << log2_ns
    int log2_tiles;       // tiles per signal = (L / C) = 1 << log2_tiles
    int real_in, conj_in, conj_out;
    int win_pairs;        // FIRST: `win` holds one weight for the real and one for the imaginary part of every sample
    int narrow;           // tile width bits (see launch_tile_mode)
    int nt;               // bit 0: non-temporal loads, bit 1: non-temporal stores (plain modes only)
    int xcd;              // 1: XCD-aware tile order (narrow tiles whose neighbours share cache lines)
    int grid_over;        // host only: resident sets of workgroups the launch asks for (prefetching kernels always one)
    int buf;              // plain loads (bit 2) / four-column tiles' plain stores (bit 3) as buffer instructions: the tile's origin
                          // in the descriptor, the row group in the scalar offset, one 32-bit lane offset.  Needs a signal of
                          // less than 2^32 bytes (checked at launch).  Measured on MI355X against global loads with a scalar
                          // base: four-column tiles +1..7 % (2^19 x 128 complex64 526 -> 491 us), wider tiles +1.5..3.6 %
                          // (complex64) / 0..1.4 % (complex128) while the plan's data is at most 128 MiB and -4..5 % beyond
    T scale;
    // fused real-input last pass (tile_r2c_kernel.h): raw outputs of the two self-mirrored columns, 2 G per signal; 1 = |X|^2 out
    cx<T>* side;
    int r2c_power;
    // FIRST_HANN: the window 0.5 - 0.5 cos(2 pi i / (n_in - 1)) is evaluated from the turn fraction i * win_inv_m;
    // win_step = ((TX * L) mod (n_in - 1)) / (n_in - 1), the fraction between a thread's consecutive samples
    double win_inv_m, win_step;
};

// sin and cos of |theta| <= pi/4 in double: Taylor polynomials through theta^15 / theta^16 (truncation
// 5e-17 / 2e-18), Horner with fused multiply-adds; 1.2e-16 absolute over the whole range.  About
// thirty instructions where the library's full-range sincospi is several hundred with branches.
__device__ __forceinline__ void sincos_eighth(double theta, double* s, double* c)
{
    const double z = theta * theta;
    double ps = -1.0 / 1307674368000.0;           // -1/15!
    ps = fma(ps, z, 1.0 / 6227020800.0);          //  1/13!
    ps = fma(ps, z, -1.0 / 39916800.0);           // -1/11!
    ps = fma(ps, z, 1.0 / 362880.0);              //  1/9!
    ps = fma(ps, z, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    *s = fma(theta * z, ps, theta);
    double pc = 1.0 / 20922789888000.0;           //  1/16!
    pc = fma(pc, z, -1.0 / 87178291200.0);        // -1/14!
    pc = fma(pc, z, 1.0 / 479001600.0);           //  1/12!
    pc = fma(pc, z, -1.0 / 3628800.0);            // -1/10!
    pc = fma(pc, z, 1.0 / 40320.0);
    pc = fma(pc, z, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    pc = fma(pc, z, -0.5);
    *c = fma(pc, z, 1.0);
}

// exp(-2*pi*i*k/n) for n = 2^log2n and k < n, to a few 1e-16 for every k: the angle is
// reduced to the nearest quarter turn in integers and the remainder, at most an eighth of a turn,
// goes through sincos_eighth.  Used where a table would be too large to stay in cache: the exponents
// k*(c0 + c) mod n of the four-step twiddle are scattered over all of n, and two-level tables of
// 2^15 + 2^16 entries read at four random places per thread cost the last column pass of a
// 2^31-point transform as much L2 traffic again as its own tile.
__device__ __forceinline__ cx<double> tile_root(uint64_t k, int log2n)      // n = 2^log2n, 2 <= n <= 2^40
{
    const uint64_t k4 = k << 2;
    const uint64_t q = (k4 + (uint64_t(1) << (log2n - 1))) >> log2n;             // nearest quarter turn
    const int64_t rem = (int64_t)k4 - (int64_t)(q << log2n);                      // |rem| <= n/2
    const double inv2n = __longlong_as_double((long long)(1023 - (log2n + 1)) << 52);   // 2^-(log2n + 1), exact
    double sa, ca;
    sincos_eighth(3.141592653589793 * ((double)rem * inv2n), &sa, &ca);           // angle = 2*pi*rem/(4n)
    double c, sn;
    switch (q & 3) {
    case 0: c = ca; sn = sa; break;
    case 1: c = -sa; sn = ca; break;
    case 2: c = -ca; sn = -sa; break;
    default: c = sa; sn = -ca; break;
    }
    return cx<double>{c, -sn};
}

// (cos, sin) of 2*pi*f for a turn fraction 0 <= f <= 1 (nearest quarter turn, remainder through sincos_eighth)
__device__ __forceinline__ cx<double> tile_turn(double f)
{
    const double q = rint(4.0 * f);
    double sa, ca;
    sincos_eighth(6.283185307179586 * fma(-0.25, q, f), &sa, &ca);
    switch ((int)q & 3) {
    case 0: return cx<double>{ca, sa};
    case 1: return cx<double>{-sa, ca};
    case 2: return cx<double>{-ca, -sa};
    default: return cx<double>{sa, -ca};
    }
}

// MID: plain strided in / strided out.  FIRST_*: sample order in, contiguous runs out, no
// inter-pass twiddle.  LAST_*: natural-order bins out.  *_PLAIN = dense full-length complex
// data and nothing fused; *_GEN / LAST_POWER carry the runtime prologue / epilogue.
enum TileMode { TM_MID = 0, TM_FIRST_PLAIN = 1, TM_FIRST_GEN = 2, TM_LAST_PLAIN = 3, TM_LAST_GEN = 4,
                TM_LAST_POWER = 5, TM_LAST_TWID = 6,
                TM_FIRST_HANN = 7,     // dense complex input of n_in <= N samples, Hann window evaluated, zero padding (prefetching tiles only)
                TM_FIRST_PRE = 8 };    // dense complex input of n_in <= N samples times ONE table (complex premultiplier or real window, or
                                       // none), zero padding: what the apps' own calls need (chirp-z, demodulation, window), 32-bit indices

// Columns per tile.  Wide (default): 512-byte rows where LDS allows -- 64 complex64 columns up to
// 128-point tiles, 32 at 256 points, 16 at 512; 32 complex128 columns up to 256 points, 16 at 512.
// Narrow (column-batched arrays with fewer columns than a wide tile): 256-byte rows, 128 at 512
// points.  Measured on MI355X, a G x run strided copy moves 4.4-5.0 TB/s at 128-B runs, 5.0-5.3
// TB/s at 256-B runs and ~5.5 TB/s from 512 B; on the transforms wide tiles are worth +5..8 %
// (complex64, 2^19..2^22 points) and +2..13 % (complex128, every length), while 64 complex64
// columns at 256 points (one 1024-thread workgroup per CU) lose 6-9 %.
// Radices 1024, 2048, 4096 (complex64, two-pass plans of 2^19..2^24 points): four columns, i.e.
// 32-byte rows -- all that fits LDS at 4096 points.  Four neighbouring tiles then share every
// 128-byte line, which only pays when they run on the same XCD (TArgs::xcd): measured as bare copies
// of the 4096 x 4 shape, 2.3 TB/s in dispatch order against 5.5 TB/s (strided in, contiguous out)
// and 1.9 against 4.7 TB/s (strided in and out) with each XCD walking one contiguous tile range.
// complex128 also has 1024 x 8 tiles (a full 128-byte line per row, the same LDS footprint as 512 x 16) and
// 2048 x 4 (64-byte rows): with them lengths of 2^28 .. 2^31 points take three passes instead of four.
#ifndef MFFT_C128_1024_COLS
#define MFFT_C128_1024_COLS 3     // log2 of the columns of a complex128 1024-point tile (2: the four-column form, for A/B builds)
#endif
template <typename T, int LOG2G, bool NARROW> struct TileCols {
    static constexpr int BIG = (sizeof(T) == 8 && LOG2G == 10) ? MFFT_C128_1024_COLS : 2;
    static constexpr int WIDE = LOG2G >= 10 ? BIG : (sizeof(T) == 4 ? (LOG2G <= 7 ? 6 : (LOG2G == 8 ? 5 : 4)) : (LOG2G <= 8 ? 5 : 4));
    static constexpr int SLIM = LOG2G >= 10 ? BIG : ((sizeof(T) == 4 ? 5 : 4) - (LOG2G >= 9 ? 1 : 0));
    static constexpr int LOG2C = NARROW ? SLIM : WIDE;
};

// LDS image of a tile: [position][column] with an odd pitch (C + 1) so that lanes along a row and
// lanes down a column are both conflict-free; four-column tiles keep pitch C (the padded pitch would
// not fit LDS at 4096 points) and turn their first-pass output through a column-major image instead.
template <int C> constexpr int tile_pitch() { return C >= 8 ? C + 1 : C; }
template <int G> constexpr int tile_colpitch() { return lds_pad<kPadShift>(G - 1) + 1 + 8; }   // = 8 mod 32 elements

// Points per thread and radices of a tile's G-point column transforms: those of the LDS-resident kernel of
// that length, except that the complex128 1024-point tiles run 4 * 16 * 16 with six-generator tables (408
// entries, 6.4 KiB: with the fifteen-power tables of 16 * 4 * 16 the 1024 x 8 image would not fit LDS).
template <typename T, int LOG2G> struct TileRadices { using RD = typename SizeCfg<LOG2G>::RD; };
template <> struct TileRadices<double, 10> { using RD = Radices<4, 16, 16, 1, true>; };
template <typename T, int LOG2G> struct TileCfg {
    using RD = typename TileRadices<T, LOG2G>::RD;
    static constexpr int G = 1 << LOG2G;
    static constexpr int P = SizeCfg<LOG2G>::P;
    static constexpr int TX = G / P;
};

// position -> LDS element index for column c: padded position times the odd pitch
template <int PADSHIFT, int PITCH> __device__ __forceinline__ int tile_addr(int pos, int c)
{
    return lds_pad<PADSHIFT>(pos) * PITCH + c;
}

// RAW: barriers that wait for LDS traffic only, so that global loads and stores issued earlier stay in
// flight across the exchange (the prefetching tiles)
template <bool RAW> __device__ __forceinline__ void tile_barrier()
{
    if constexpr (RAW) lds_barrier(); else __syncthreads();
}

template <typename T, class RD, int S, int P, int TX, int PADSHIFT, int PITCH, bool RAW = false>
__device__ __forceinline__ void tile_exchange(cx<T> (&v)[P], int t, int c, cx<T>* lds)
{
    constexpr int R = RD::r[S], U = P / R, NS = RD::ns(S);
    tile_barrier<RAW>();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = t + TX * u;
        const int base = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
        for (int q = 0; q < R; ++q) lds[tile_addr<PADSHIFT, PITCH>(base + q * NS, c)] = v[u + U * q];
    }
    tile_barrier<RAW>();
#pragma unroll
    for (int e = 0; e < P; ++e) v[e] = lds[tile_addr<PADSHIFT, PITCH>(t + TX * e, c)];
}

#ifndef MFFT_BIG_RADIX
#define MFFT_BIG_RADIX 0     // 1 = also build the 2048- and 4096-point tile radices (tuning builds; MFFT_SPLIT selects them)
#endif
#ifndef MFFT_TILE_OCC
#define MFFT_TILE_OCC 1      // 1 = hold the plain modes at the VGPR count that lets two workgroups share a CU
#endif
// Waves per SIMD the register allocator must leave room for.  A workgroup of B threads is B/256
// waves per SIMD; two of them fit a CU's LDS for every tile shape, but only if each wave stays at
// 512 / (B/128) VGPRs.  Several plain modes compiled to just above that (130-154 against 128), i.e.
// to one resident workgroup and no overlap between its load, compute and store phases.
// Measured: +1..4 % on every complex64 length; complex128 (whose spills under the cap reach 40-280
// bytes per lane) lost up to 11 %, so it keeps the compiler's own allocation.
template <typename T, int BLOCK, int MODE> constexpr int tile_min_waves()
{
    constexpr bool plain = MODE == TM_MID || MODE == TM_FIRST_PLAIN || MODE == TM_LAST_PLAIN || MODE == TM_LAST_TWID ||
                           MODE == TM_FIRST_HANN;
    return (MFFT_TILE_OCC != 0 && sizeof(T) == 4 && plain && BLOCK <= 512) ? BLOCK / 128 : 1;
}

// Prefetch.  A complex128 tile of 512 points x 16 columns or 256 x 32 fills LDS, so ONE 512-thread
// workgroup is resident per CU -- two waves per SIMD, 256 registers each -- and nothing overlaps its
// load, butterfly and store phases.  The 512-point kernels therefore request the next tile into a
// second register set behind their first exchange; the loads travel during the remaining two passes
// and the stores.  For that the barriers of the exchanges must not drain the vector-memory counter
// (tile_barrier<true>), and nothing else inside the loop may load: the twiddle roots are evaluated
// (tile_root) instead of read from the tables.
// (The 256 x 32 tiles, two passes and one exchange, measured no gain -- 2^24: -1..2 %, 2^31: first
// pass -6 %, middle passes +6 % -- and keep the plain loop.)
template <typename T, int LOG2G, int LOG2C, int MODE> constexpr bool tile_prefetches()
{
    using TC = TileCfg<T, LOG2G>;
    using RD = typename TC::RD;
    constexpr int G = TC::G, C = 1 << LOG2C, BLOCK = TC::TX * C;
    constexpr int LDS_ELEMS = tile_pitch<C>() == C ? tile_colpitch<G>() * C : (lds_pad<kPadShift>(G - 1) + 1) * tile_pitch<C>();
    constexpr bool PLAIN_IN = MODE == TM_MID || MODE == TM_FIRST_PLAIN || MODE == TM_LAST_PLAIN || MODE == TM_LAST_TWID ||
                              (MFFT_TILE_PF_POWER != 0 && MODE == TM_LAST_POWER) || (MFFT_TILE_PF_GEN != 0 && MODE == TM_LAST_GEN) ||
                              MODE == TM_FIRST_HANN;
    constexpr bool ONE_WG = 2 * (size_t)LDS_ELEMS * sizeof(cx<T>) > 160 * 1024;
    return MFFT_TILE_PREFETCH != 0 && MFFT_DIAG_NOFFT == 0 && sizeof(T) == 8 && ONE_WG && BLOCK == 512 && RD::npass >= 3 && PLAIN_IN;
}

template <typename T, int LOG2G, int LOG2C, int MODE>
__global__ void __launch_bounds__(((TileCfg<T, LOG2G>::TX) << LOG2C), (tile_min_waves<T, ((TileCfg<T, LOG2G>::TX) << LOG2C), MODE>()))
fft_tile_kernel(const TArgs<T> a)
{
    constexpr bool FIRST = (MODE == TM_FIRST_PLAIN || MODE == TM_FIRST_GEN || MODE == TM_FIRST_HANN || MODE == TM_FIRST_PRE);
    using C_ = cx<T>;
    using TC = TileCfg<T, LOG2G>;
    using RD = typename TC::RD;
    constexpr int G = TC::G, P = TC::P, TX = TC::TX;
    constexpr int C = 1 << LOG2C;
    constexpr int BLOCK = TX * C;
    constexpr int PITCH = tile_pitch<C>();
    constexpr int PADSHIFT = kPadShift;
    constexpr int NPASS = RD::npass;
    constexpr int COLPITCH = tile_colpitch<G>();
    constexpr int LDS_ELEMS = PITCH == C ? COLPITCH * C : (lds_pad<PADSHIFT>(G - 1) + 1) * PITCH;
    constexpr bool PF = tile_prefetches<T, LOG2G, LOG2C, MODE>();

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C_* lds = reinterpret_cast<C_*>(smem_raw);
    C_* twl = lds + LDS_ELEMS;                // per-pass tables of the G-point DFT

    const int c = threadIdx.x % C;            // column of the tile = lane-fastest
    const int t = threadIdx.x / C;            // thread within the column's DFT
    for (int i = threadIdx.x; i < tw_total_elems<RD>(); i += BLOCK) twl[i] = a.tw[i];
    __syncthreads();

    // NS, L and the tile count are powers of two: shifts and masks, no 64-bit division
    const int64_t total = a.batch << a.log2_tiles;
    const int64_t ns_mask = a.NS - 1;
    // Workgroup w runs on XCD w % 8 (round-robin dispatch).  With a.xcd (four-column tiles) every XCD
    // takes one contiguous range of tiles and walks it in dispatch order, so that neighbouring
    // tiles, which share cache lines at that width, meet in one L2 instead of eight.
    constexpr bool SLIM4 = LOG2C <= 2;
    const bool xcd_map = SLIM4 && a.xcd != 0 && (gridDim.x & 7) == 0 && total >= 64;
    const int64_t w_off = xcd_map ? (int64_t)(blockIdx.x & 7) * ((total + 7) >> 3) : 0;
    const int64_t w_end = xcd_map ? ((total + 7) >> 3) : total;
    const int64_t w_step = xcd_map ? (gridDim.x >> 3) : gridDim.x;
    // dense tile of super-butterfly group w into registers (the plain modes' load)
    [[maybe_unused]] const auto load_plain = [&](int64_t w, C_ (&dst)[P]) {
        const int64_t b = w >> a.log2_tiles;
        const int64_t j0 = (w & ((int64_t(1) << a.log2_tiles) - 1)) * C;
        const C_* __restrict__ p = reinterpret_cast<const C_*>(a.in) + b * a.in_stride + j0 +
                                   ((uint32_t)c + (uint32_t)t * (uint32_t)a.L);
        const int64_t pstep = (int64_t)TX * a.L;
        if constexpr (MODE == TM_FIRST_HANN) {
            // n_in <= N samples per signal: indices past the last sample read that sample (zeroed in the trip)
            const C_* __restrict__ sig = reinterpret_cast<const C_*>(a.in) + b * a.in_stride;
            const int64_t i0 = j0 + ((uint32_t)c + (uint32_t)t * (uint32_t)a.L), last = a.n_in - 1;
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const int64_t i = i0 + e * pstep;
                dst[e] = tile_ld<T, false>(sig + (i < last ? i : last));
            }
        } else if (a.nt & 1) {
#pragma unroll
            for (int e = 0; e < P; ++e) { dst[e] = tile_ld<T, true>(p); p += pstep; }
        } else {
#pragma unroll
            for (int e = 0; e < P; ++e) { dst[e] = tile_ld<T, false>(p); p += pstep; }
        }
    };
    // the four table entries behind a tile's inter-pass twiddles (base and step, low and high part)
    [[maybe_unused]] const auto tw_request = [&](int64_t jj, cx<double> (&o)[4]) {
        const uint32_t m = (uint32_t)((jj & ns_mask) >> a.log2_w);   // NS / W <= 2^30
        const uint32_t lomask = (1u << a.lb) - 1;
        const uint64_t kb = (uint64_t)m * (uint32_t)t;        // base exponent  m*t      < NS*G
        const uint64_t ks = (uint64_t)m * (uint32_t)TX;       // step exponent  m*TX     < NS*G
        o[0] = a.tw_lo[(uint32_t)kb & lomask]; o[1] = a.tw_hi[kb >> a.lb];
        o[2] = a.tw_lo[(uint32_t)ks & lomask]; o[3] = a.tw_hi[ks >> a.lb];
    };
    const int64_t wl0 = xcd_map ? (blockIdx.x >> 3) : blockIdx.x;
    // One tile.  v: its samples -- already requested when PF --, vn: where a prefetching kernel requests
    // the next tile's.  Returns false past the last tile.
    const auto trip = [&](int64_t wl, C_ (&v)[P], [[maybe_unused]] C_ (&vn)[P]) -> bool {
        const int64_t w = w_off + wl;
        if (SLIM4 && w >= total) return false;
        const int64_t b = w >> a.log2_tiles;
        const int64_t j0 = (w & ((int64_t(1) << a.log2_tiles) - 1)) * C;
        const int64_t j = j0 + c;

        // ---- load the G x C tile (row i = t + TX*e, column c) --------------------
        // Four-column tiles address as a wave-uniform 64-bit base (tile origin + e row groups: scalar
        // registers) plus one 32-bit lane offset (column + t rows), so sixteen accesses cost one address
        // register (they run in 1024-thread workgroups: 128 VGPRs); they always use plain accesses --
        // a streaming load would not keep the line for the three neighbours that share it.  Wider
        // tiles step a per-lane pointer, which measured 2-4 % faster there.
        const uint32_t lane_in = (uint32_t)c + (uint32_t)t * (uint32_t)a.L;          // < N / P
        const C_* __restrict__ in_base = reinterpret_cast<const C_*>(a.in) + b * a.in_stride + j0;
        const int64_t pstep = (int64_t)TX * a.L;
        if constexpr (MODE == TM_FIRST_PLAIN) {
            if constexpr (PF) {
                // loaded above / by the previous trip
            } else if (a.buf & 4) {
                // buffer form: the tile's origin in the descriptor, the row group in the scalar offset, one lane offset
                const auto rs = make_rsrc(in_base, 0xffffffffu);
                const uint32_t voff = lane_in * (uint32_t)sizeof(C_), sstep = (uint32_t)pstep * (uint32_t)sizeof(C_);
                if (a.nt & 1) {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 2>(rs, voff, (uint32_t)e * sstep);
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 0>(rs, voff, (uint32_t)e * sstep);
                }
            } else if constexpr (SLIM4) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = tile_ld<T, false>(in_base + e * pstep + lane_in);
            } else {
                const C_* __restrict__ p = in_base + lane_in;
                if (a.nt & 1) {
#pragma unroll
                    for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, true>(p); p += pstep; }
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, false>(p); p += pstep; }
                }
            }
        } else if constexpr (MODE == TM_FIRST_HANN) {
            // samples: requested above / by the previous trip (prefetching tiles), else loaded here.  Window of sample i = i0 + e*pstep: the point
            // exp(2 pi i * i/(n_in - 1)) is stepped along the thread's constant stride (double recurrence, P steps)
            if constexpr (!PF) {
                // buffer loads: one lane offset, wave-uniform row-group offsets, and the hardware range check returns
                // zero past the last sample -- zero padding without clamps or predicates (N * sizeof < 2^32)
                const auto rs = make_rsrc(reinterpret_cast<const C_*>(a.in) + b * a.in_stride, (uint32_t)(a.n_in * (int64_t)sizeof(C_)));
                const uint32_t voff = ((uint32_t)c + (uint32_t)t * (uint32_t)a.L) * (uint32_t)sizeof(C_);
                const uint32_t s0 = (uint32_t)j0 * (uint32_t)sizeof(C_), sstep = (uint32_t)TX * (uint32_t)a.L * (uint32_t)sizeof(C_);
                if (a.nt & 1) {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 2>(rs, voff, s0 + (uint32_t)e * sstep);
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 0>(rs, voff, s0 + (uint32_t)e * sstep);
                }
            }
            const int64_t i0 = j + (int64_t)t * a.L;
            if constexpr (!PF) {
                if (a.win) {      // samples past the last one are zero already
                    cx<double> z = tile_turn((double)(i0 < a.n_in ? i0 : 0) * a.win_inv_m);
                    const cx<double> zs = tile_turn(a.win_step);
#pragma unroll
                    for (int e = 0; e < P; ++e) {
                        const double wv = fma(-0.5, z.x, 0.5);
                        if constexpr (sizeof(T) == 4) {
                            const float wf = (float)wv;
                            v[e].x *= wf; v[e].y *= wf;
                        } else {
                            v[e].x *= wv; v[e].y *= wv;
                        }
                        z = cx<double>{z.x * zs.x - z.y * zs.y, z.x * zs.y + z.y * zs.x};
                    }
                }
            } else {
                cx<double> z = tile_turn((double)(i0 < a.n_in ? i0 : 0) * a.win_inv_m);
                double ws = a.win_step;
                asm volatile("" : "+v"(ws));       // evaluated per trip: hoisted out of the loop the step costs four registers for its whole length
                const cx<double> zs = tile_turn(ws);
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    const double wv = (i0 + e * pstep < a.n_in) ? (a.win ? fma(-0.5, z.x, 0.5) : 1.0) : 0.0;    // zero padding
                    v[e].x = (T)((double)v[e].x * wv);
                    v[e].y = (T)((double)v[e].y * wv);
                    z = cx<double>{z.x * zs.x - z.y * zs.y, z.x * zs.y + z.y * zs.x};
                }
            }
        } else if constexpr (MODE == TM_FIRST_PRE) {
            // sample index of v[e]: i0 + e * istep < N <= 2^31.  Loads and table reads are issued unconditionally at
            // the index clamped to the last sample (they stay in flight together); samples past it are zeroed.
            const uint32_t last = (uint32_t)(a.n_in - 1);
            const uint32_t i0 = (uint32_t)j + (uint32_t)t * (uint32_t)a.L, istep = (uint32_t)TX * (uint32_t)a.L;
            const C_* __restrict__ sig = reinterpret_cast<const C_*>(a.in) + b * a.in_stride;
            const auto at = [&](int e) { const uint32_t i = i0 + (uint32_t)e * istep; return i < last ? i : last; };
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = sig[at(e)];
            if (a.pre) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = cmul(v[e], a.pre[at(e)]);
            } else if (a.win) {
#pragma unroll
                for (int e = 0; e < P; ++e) { const T wv = a.win[at(e)]; v[e].x *= wv; v[e].y *= wv; }
            }
#pragma unroll
            for (int e = 0; e < P; ++e) {
                if (i0 + (uint32_t)e * istep > last) v[e] = C_{0, 0};             // zero padding
            }
        } else if constexpr (MODE == TM_FIRST_GEN) {
            // runtime prologue: unconditional loads at clamped indices, options as uniform
            // branches around whole loops (keeps the tile's loads in flight together)
            const int64_t last = a.n_in - 1;
            int64_t idx[P];
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const int64_t i = j + (int64_t)(t + TX * e) * a.L;   // sample index in the signal
                idx[e] = i < last ? i : last;
            }
            const int64_t base = b * a.in_stride;
            if (a.in_seg > 0) {
                if (a.real_in) {
#pragma unroll
                    for (int e = 0; e < P; ++e)
                        v[e] = C_{reinterpret_cast<const T*>(a.in)[base + seg_index(idx[e], a.in_seg, a.in_seg_stride)], 0};
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e)
                        v[e] = reinterpret_cast<const C_*>(a.in)[base + seg_index(idx[e], a.in_seg, a.in_seg_stride)];
                }
            } else if (a.real_in) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = C_{reinterpret_cast<const T*>(a.in)[base + idx[e]], 0};
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = reinterpret_cast<const C_*>(a.in)[base + idx[e]];
            }
            if (a.conj_in) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e].y = -v[e].y;
            }
            if (a.win) {
                if (a.win_pairs) {
                    // packed real samples z[j] = x[2j] + i x[2j+1]: each part takes its own weight
                    const C_* __restrict__ wp = reinterpret_cast<const C_*>(a.win);
#pragma unroll
                    for (int e = 0; e < P; ++e) { const C_ w2 = wp[idx[e]]; v[e].x *= w2.x; v[e].y *= w2.y; }
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) { const T wv = a.win[idx[e]]; v[e].x *= wv; v[e].y *= wv; }
                }
            }
            if (a.pre) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = cmul(v[e], a.pre[idx[e]]);
            }
#pragma unroll
            for (int e = 0; e < P; ++e) {
                if (j + (int64_t)(t + TX * e) * a.L > last) v[e] = C_{0, 0};   // zero padding
            }
        } else {
            if constexpr (PF) {
                // loaded above / by the previous trip
            } else if (a.buf & 4) {
                // buffer form: the tile's origin in the descriptor, the row group in the scalar offset, one lane offset
                const auto rs = make_rsrc(in_base, 0xffffffffu);
                const uint32_t voff = lane_in * (uint32_t)sizeof(C_), sstep = (uint32_t)pstep * (uint32_t)sizeof(C_);
                if (a.nt & 1) {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 2>(rs, voff, (uint32_t)e * sstep);
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) v[e] = buf_ld<C_, 0>(rs, voff, (uint32_t)e * sstep);
                }
            } else if constexpr (SLIM4) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = tile_ld<T, false>(in_base + e * pstep + lane_in);
            } else {
                const C_* __restrict__ p = in_base + lane_in;
                if (a.nt & 1) {
#pragma unroll
                    for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, true>(p); p += pstep; }
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, false>(p); p += pstep; }
                }
            }
#if !MFFT_DIAG_NOFFT
            cx<double> w, step;
            if constexpr (PF || MFFT_TILE_EVAL_ROOTS == 1 || (MFFT_TILE_EVAL_ROOTS == 2 && sizeof(T) == 4 && LOG2G >= 9)) {
                // evaluated, not read: table loads issued here would retire behind the previous trip's
                // stores (the counter is in order), i.e. drain them, and prefetched entries cost sixteen
                // registers these kernels do not have
                const uint32_t m = (uint32_t)((j & ns_mask) >> a.log2_w);
                const int log2m = a.log2_ns - a.log2_w + LOG2G;           // the tables' modulus (NS / W) * G
                w = tile_root((uint64_t)m * (uint32_t)t, log2m);
                step = tile_root((uint64_t)m * (uint32_t)TX, log2m);
            } else {
                cx<double> tq[4];
                tw_request(j, tq);
                w = cmul(tq[0], tq[1]);
                step = cmul(tq[2], tq[3]);
            }
            if constexpr (sizeof(T) == 4 && P % 4 == 0 && MFFT_TILE_TW2 != 0) {
                // complex64: W^(m (t + TX e)) = B_k S_r with e = 4k + r, B_k = base step^(4k), S_r = step^r.
                // Six double-precision products build the P/4 + 3 factors; the P combinations and the
                // products with the data are packed single-precision ones (a twiddle is then two
                // roundings from exact, like a table entry times a derived power in the LDS kernels)
                // instead of a P-step double recurrence and P double-precision data products.
                const cx<double> s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
                const C_ S1{(T)step.x, (T)step.y}, S2{(T)s2.x, (T)s2.y}, S3{(T)s3.x, (T)s3.y};
#pragma unroll
                for (int k = 0; k < P / 4; ++k) {
                    const C_ B{(T)w.x, (T)w.y};
                    constexpr bool VEC = MFFT_TILE_VEC != 0;
                    v[4 * k] = cmul<T, VEC>(v[4 * k], B);
                    v[4 * k + 1] = cmul<T, VEC>(v[4 * k + 1], cmul<T, VEC>(B, S1));
                    v[4 * k + 2] = cmul<T, VEC>(v[4 * k + 2], cmul<T, VEC>(B, S2));
                    v[4 * k + 3] = cmul<T, VEC>(v[4 * k + 3], cmul<T, VEC>(B, S3));
                    w = cmul(w, s4);
                }
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    const double xr = (double)v[e].x, xi = (double)v[e].y;
                    v[e] = C_{(T)(xr * w.x - xi * w.y), (T)(xr * w.y + xi * w.x)};
                    w = cmul(w, step);
                }
            }
#endif
        }

        // ---- G-point DFT of every column, in LDS -----------------------------------
#if !MFFT_DIAG_NOFFT
        pass_compute<T, RD, 0, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
        if constexpr (NPASS > 1) {
            tile_exchange<T, RD, 0, P, TX, PADSHIFT, PITCH, PF>(v, t, c, lds);
            if constexpr (PF) {
                // Requests behind the first exchange: by now the previous trip's stores, which read the
                // register set these loads refill, have drained, and the loads have the remaining passes to
                // arrive.  (The last trip asks for its own tile again -- an L2 hit -- rather than branch
                // around the requests.)
                asm volatile("" ::: "memory");
                load_plain(wl + w_step < w_end ? w + w_step : w, vn);
                asm volatile("" ::: "memory");
            }
            pass_compute<T, RD, 1, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
        }
        if constexpr (NPASS > 2) {
            tile_exchange<T, RD, 1, P, TX, PADSHIFT, PITCH, PF>(v, t, c, lds);
            pass_compute<T, RD, 2, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
        }
        if constexpr (NPASS > 3) {
            tile_exchange<T, RD, 2, P, TX, PADSHIFT, PITCH, PF>(v, t, c, lds);
            pass_compute<T, RD, 3, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
        }
#endif
        // v[e] = output q = t + TX*e of column c

        // ---- store -------------------------------------------------------------------
        if constexpr (FIRST) {
            // NS = 1: y[(j0 + c)*G + q] -- the tile is one contiguous run of C*G elements,
            // q fastest.  Turn the registers through LDS so that lanes run along q.
            tile_barrier<PF>();
#pragma unroll
            for (int e = 0; e < P; ++e) {
                if constexpr (PITCH == C) lds[c * COLPITCH + lds_pad<PADSHIFT>(t + TX * e)] = v[e];
                else lds[tile_addr<PADSHIFT, PITCH>(t + TX * e, c)] = v[e];
            }
            tile_barrier<PF>();
            C_* __restrict__ dst = reinterpret_cast<C_*>(a.out) + b * a.out_stride + j0 * G;
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const int r = threadIdx.x + BLOCK * e;      // element of the run: r = cc*G + q
                const int cc = r / G, q = r % G;
                C_ val;
                if constexpr (PITCH == C) val = lds[cc * COLPITCH + lds_pad<PADSHIFT>(q)];
                else val = lds[tile_addr<PADSHIFT, PITCH>(q, cc)];
                if constexpr (SLIM4) tile_st<T, false>(dst + BLOCK * e + threadIdx.x, val);
                else if (a.nt & 2) tile_st<T, true>(dst + r, val);
                else tile_st<T, false>(dst + r, val);
            }
        } else if constexpr (MODE == TM_MID) {
            // a tile never straddles a block of NS super-butterflies (C <= NS, both powers of two)
            C_* __restrict__ out_base = reinterpret_cast<C_*>(a.out) + b * a.out_stride +
                                        ((j0 >> a.log2_ns) << a.log2_ns) * G + (j0 & ns_mask);
            const uint32_t lane_out = (uint32_t)c + (uint32_t)t * (uint32_t)a.NS;
            const int64_t qstep = (int64_t)TX * a.NS;
            if (SLIM4 && (a.buf & 8)) {
                const auto rs = make_rsrc(out_base, 0xffffffffu);
                const uint32_t voff = lane_out * (uint32_t)sizeof(C_), sstep = (uint32_t)qstep * (uint32_t)sizeof(C_);
#pragma unroll
                for (int e = 0; e < P; ++e) buf_st<C_, 0>(rs, voff, (uint32_t)e * sstep, v[e]);
            } else if constexpr (SLIM4) {
#pragma unroll
                for (int e = 0; e < P; ++e) tile_st<T, false>(out_base + e * qstep + lane_out, v[e]);
            } else {
                C_* __restrict__ q = out_base + lane_out;
                if (a.nt & 2) {
#pragma unroll
                    for (int e = 0; e < P; ++e) { tile_st<T, true>(q, v[e]); q += qstep; }
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) { tile_st<T, false>(q, v[e]); q += qstep; }
                }
            }
        } else if constexpr (MODE == TM_LAST_TWID) {
            // last pass of a column-batched transform with the four-step twiddle folded in:
            // row k = (j >> log2_w) + q*(NS >> log2_w), column cc = tw_c0 + (j & (W-1));
            // exponent k*cc = base + q*step: two roots of unity per thread and a recurrence in double
            // (kept below the column DFT: evaluated above it the two roots live through the butterflies)
#pragma unroll
            for (int e = 0; e < P; ++e) asm volatile("" : "+v"(v[e].x), "+v"(v[e].y));
            const uint64_t cc = (uint64_t)(a.tw_c0 + (j & ((int64_t(1) << a.log2_w) - 1)));
            const uint64_t nsr = (uint64_t)(a.NS >> a.log2_w);
            const uint64_t kb = (((uint64_t)(j >> a.log2_w) + (uint64_t)t * nsr) * cc) & a.fs_mask;
            const uint64_t ks = ((uint64_t)TX * nsr * cc) & a.fs_mask;
            const int fs_log2n = __builtin_popcountll(a.fs_mask);
            cx<double> w = tile_root(kb, fs_log2n);
            const cx<double> step = tile_root(ks, fs_log2n);
            C_* __restrict__ q = reinterpret_cast<C_*>(a.out) + b * a.out_stride + j + (int64_t)t * a.NS;
            const int64_t qstep = (int64_t)TX * a.NS;
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const double xr = (double)v[e].x, xi = (double)v[e].y;
                const C_ val{(T)(xr * w.x - xi * w.y), (T)(xr * w.y + xi * w.x)};
                if (a.nt & 2) tile_st<T, true>(q, val);
                else tile_st<T, false>(q, val);
                q += qstep;
                w = cmul(w, step);
            }
        } else if constexpr (MODE == TM_LAST_PLAIN) {
            // last pass: NS*G = N, so j < NS and bin k = j + q*NS.  Column-batched plans may store rows of
            // the (n x W) result out_pitch elements apart (a column block of a wider array): row k of the
            // result is (j >> log2_w) + q * (NS >> log2_w), its column j & (W - 1); out_pitch = W is dense
            if (a.scale != (T)1) {                  // np.fft.fft's norm factors, dt * F: not a reason for the general epilogue
#pragma unroll
                for (int e = 0; e < P; ++e) { v[e].x *= a.scale; v[e].y *= a.scale; }
            }
            const int64_t nso = (a.NS >> a.log2_w) * a.out_pitch;
            C_* __restrict__ out_base = reinterpret_cast<C_*>(a.out) + b * a.out_stride + (j0 >> a.log2_w) * a.out_pitch +
                                        (j0 & ((int64_t(1) << a.log2_w) - 1));
            const uint32_t lane_out = (uint32_t)c + (uint32_t)t * (uint32_t)nso;
            const int64_t qstep = (int64_t)TX * nso;
            if (SLIM4 && (a.buf & 8)) {
                const auto rs = make_rsrc(out_base, 0xffffffffu);
                const uint32_t voff = lane_out * (uint32_t)sizeof(C_), sstep = (uint32_t)qstep * (uint32_t)sizeof(C_);
#pragma unroll
                for (int e = 0; e < P; ++e) buf_st<C_, 0>(rs, voff, (uint32_t)e * sstep, v[e]);
            } else if constexpr (SLIM4) {
#pragma unroll
                for (int e = 0; e < P; ++e) tile_st<T, false>(out_base + e * qstep + lane_out, v[e]);
            } else {
                C_* __restrict__ q = out_base + lane_out;
                if (a.nt & 2) {
#pragma unroll
                    for (int e = 0; e < P; ++e) { tile_st<T, true>(q, v[e]); q += qstep; }
                } else {
#pragma unroll
                    for (int e = 0; e < P; ++e) { tile_st<T, false>(q, v[e]); q += qstep; }
                }
            }
        } else {
            double mx = 0.0, sm = 0.0, sk = 0.0;
            // output bin of v[e]: k = j + (t + TX*e) * NS, formed where it is used (an array of them costs
            // 2 P registers the prefetching kernels do not have)
            const int64_t kk0 = j + (int64_t)t * a.NS, kkstep = (int64_t)TX * a.NS;
            const auto kk = [&](int e) { return kk0 + e * kkstep; };
#pragma unroll
            for (int e = 0; e < P; ++e) { v[e].x *= a.scale; v[e].y *= a.scale; }
            if (a.conj_out) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e].y = -v[e].y;
            }
            if (a.post) {
#pragma unroll
                for (int e = 0; e < P; ++e) v[e] = cmul(v[e], a.post[kk(e) < a.n_out ? kk(e) : a.n_out - 1]);
            }
            const int64_t obase = b * a.out_stride;
            if constexpr (MODE == TM_LAST_POWER) {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    const T s = v[e].x * v[e].x + v[e].y * v[e].y;
                    const int64_t k = kk(e);
                    if (k < a.n_out) {
                        reinterpret_cast<T*>(a.out)[obase + seg_index(k, a.out_seg, a.out_seg_stride)] = s;
                        const double f = (k < (a.n_out + 1) / 2) ? (double)k : (double)(k - a.n_out);
                        mx = (double)s > mx ? (double)s : mx;
                        sm += (double)s;
                        sk += f * (double)s;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    if (kk(e) < a.n_out)
                        reinterpret_cast<C_*>(a.out)[obase + seg_index(kk(e), a.out_seg, a.out_seg_stride)] = v[e];
                }
            }
            if constexpr (MODE == TM_LAST_POWER) {
                if (a.stats != nullptr) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const double omx = __shfl_xor(mx, o, 64);
                        mx = omx > mx ? omx : mx;
                        sm += __shfl_xor(sm, o, 64);
                        sk += __shfl_xor(sk, o, 64);
                    }
                    if ((threadIdx.x & 63) == 0) {
                        double* st = a.stats + 3 * b;
                        atomicMax(reinterpret_cast<unsigned long long*>(st),
                                  (unsigned long long)__double_as_longlong(mx));
                        atomicAdd(st + 1, sm);
                        atomicAdd(st + 2, sk);
                    }
                }
            }
        }
        return true;
    };
    if constexpr (PF) {
        // Two register sets that swap roles from trip to trip (no copies: a trip's stores are still
        // reading the set the next trip's prefetch must not yet overwrite).  The first tile is requested
        // and awaited before the loop, so that inside it the only loads pending at the top of a trip
        // are those the previous trip issued ahead of its stores.
        C_ va[P], vb[P];
        if (wl0 < w_end) {
            load_plain(w_off + wl0, va);
#pragma unroll
            for (int e = 0; e < P; ++e) asm volatile("" : "+v"(va[e].x), "+v"(va[e].y));
        }
        for (int64_t wl = wl0; wl < w_end; wl += 2 * w_step) {
            trip(wl, va, vb);
            if (wl + w_step < w_end) trip(wl + w_step, vb, va);
        }
    } else {
        for (int64_t wl = wl0; wl < w_end; wl += w_step) {
            C_ v[P];
            if (!trip(wl, v, v)) break;
        }
    }
}

template <typename T, int LOG2G, int LOG2C> constexpr size_t tile_lds_bytes()
{
    constexpr int G = 1 << LOG2G, C = 1 << LOG2C;
    using RD = typename TileCfg<T, LOG2G>::RD;
    constexpr size_t image = tile_pitch<C>() == C ? (size_t)tile_colpitch<G>() * C
                                                  : (size_t)(lds_pad<kPadShift>(G - 1) + 1) * tile_pitch<C>();
    return (image + (size_t)tw_total_elems<RD>()) * sizeof(cx<T>);
}

template <typename T, int LOG2G, int LOG2C, int MODE>
int launch_tile_one(const TArgs<T>& a, hipStream_t s)
{
    constexpr int BLOCK = TileCfg<T, LOG2G>::TX << LOG2C;
    constexpr size_t lds = tile_lds_bytes<T, LOG2G, LOG2C>();
    static_assert(BLOCK <= 1024, "tile workgroup too large");
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    auto kern = fft_tile_kernel<T, LOG2G, LOG2C, MODE>;
    static thread_local int attr_dev = -1, per_cu = 1;
    int dev = 0;
    MFFT_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        MFFT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_cu = blocks_per_cu(reinterpret_cast<const void*>(kern), BLOCK, lds);
        attr_dev = dev;
    }
    const int64_t total = a.batch << a.log2_tiles;
    if (total <= 0) return MFFT_OK;
    // The prefetching kernels chain one tile's loads behind the previous tile's exchanges and want exactly
    // one resident set of workgroups striding over the tiles; the others gain 1.5-4 % from a grid several
    // times that, which lets the dispatcher even out the tails (2^24: 141.8 -> 139.7 us, 2^26: 650 -> 623 us),
    // and the four-column tiles of large batches 14 % from twice as many again (2^19 x 128: 485 -> 415 us).
    const int over = tile_prefetches<T, LOG2G, LOG2C, MODE>() ? 1 : (LOG2C <= 2 ? 2 : 1) * a.grid_over;
    const int grid = pick_grid(per_cu, total, over);
    if (a.buf) {
        // byte offsets of the buffer form stay below 2^32: G rows of L elements in, G rows of NS (or out_pitch) out
        const int64_t nso = (MODE == TM_LAST_PLAIN) ? (a.NS >> a.log2_w) * a.out_pitch : a.NS;
        const int64_t lim = (int64_t(1) << 32) / (int64_t)sizeof(cx<T>);
        TArgs<T> h = a;
        if ((a.L << LOG2G) > lim) h.buf &= ~4;
        if ((nso << LOG2G) > lim) h.buf &= ~8;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, h);
        MFFT_HIP(hipGetLastError());
        return MFFT_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, a);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

// which tiles have the evaluated-window first pass (all; MFFT_TILE_PF_HANN=0 builds none, for A/B)
template <typename T, int LOG2G> constexpr bool tile_first_hann()
{
    return MFFT_TILE_PF_HANN != 0 && MFFT_DIAG_NOFFT == 0;
}

// every mode of one radix at one tile width
template <typename T, int LOG2G, int LC>
int launch_tile_modes_at(int mode, const TArgs<T>& a, hipStream_t s)
{
    switch (mode) {
    case TM_MID: return launch_tile_one<T, LOG2G, LC, TM_MID>(a, s);
    case TM_FIRST_PLAIN: return launch_tile_one<T, LOG2G, LC, TM_FIRST_PLAIN>(a, s);
    case TM_FIRST_GEN: return launch_tile_one<T, LOG2G, LC, TM_FIRST_GEN>(a, s);
    case TM_FIRST_PRE: return launch_tile_one<T, LOG2G, LC, TM_FIRST_PRE>(a, s);
    case TM_LAST_PLAIN: return launch_tile_one<T, LOG2G, LC, TM_LAST_PLAIN>(a, s);
    case TM_LAST_GEN: return launch_tile_one<T, LOG2G, LC, TM_LAST_GEN>(a, s);
    case TM_LAST_POWER: return launch_tile_one<T, LOG2G, LC, TM_LAST_POWER>(a, s);
    case TM_LAST_TWID: return launch_tile_one<T, LOG2G, LC, TM_LAST_TWID>(a, s);
    case TM_FIRST_HANN:
        if constexpr (tile_first_hann<T, LOG2G>()) {
            if (a.n_in < 2) break;
            TArgs<T> h = a;
            const int64_t m = a.n_in - 1;
            h.win_inv_m = (double)(1.0L / (long double)m);
            h.win_step = (double)((long double)(((int64_t)TileCfg<T, LOG2G>::TX * a.L) % m) / (long double)m);
            return launch_tile_one<T, LOG2G, LC, TM_FIRST_HANN>(h, s);
        } else {
            break;
        }
    }
    set_error("unknown tile mode");
    return MFFT_E_INVALID;
}

// TArgs::narrow: bit 0 = the narrow width class of a column-batched array with few columns (TileCols), bit 1 =
// the four-column form of the complex128 1024-point tiles (plans whose data stays in the Infinity Cache)
template <typename T, int LOG2G>
int launch_tile_mode(int mode, const TArgs<T>& a, hipStream_t s)
{
    constexpr int LC = TileCols<T, LOG2G, false>::LOG2C;
    constexpr int LN = TileCols<T, LOG2G, true>::LOG2C;
    if ((a.narrow & 1) && LN != LC) {
        // only the modes a column-batched transform uses exist at the narrow width
        switch (mode) {
        case TM_MID: return launch_tile_one<T, LOG2G, LN, TM_MID>(a, s);
        case TM_LAST_PLAIN: return launch_tile_one<T, LOG2G, LN, TM_LAST_PLAIN>(a, s);
        case TM_LAST_TWID: return launch_tile_one<T, LOG2G, LN, TM_LAST_TWID>(a, s);
        }
        set_error("tile mode not built at the narrow width");
        return MFFT_E_UNSUPPORTED;
    }
    if constexpr (sizeof(T) == 8 && LOG2G == 10 && LC != 2) {
        if (a.narrow & 2) return launch_tile_modes_at<T, LOG2G, 2>(mode, a, s);
    }
    return launch_tile_modes_at<T, LOG2G, LC>(mode, a, s);
}

template <typename T>
int launch_tile_any(int log2g, int mode, const TArgs<T>& a, hipStream_t s)
{
    switch (log2g) {
    case 4: return launch_tile_mode<T, 4>(mode, a, s);
    case 5: return launch_tile_mode<T, 5>(mode, a, s);
    case 6: return launch_tile_mode<T, 6>(mode, a, s);
    case 7: return launch_tile_mode<T, 7>(mode, a, s);
    case 8: return launch_tile_mode<T, 8>(mode, a, s);
    case 9: return launch_tile_mode<T, 9>(mode, a, s);
    case 10: return launch_tile_mode<T, 10>(mode, a, s);
    case 11: if constexpr (sizeof(T) == 8 || MFFT_BIG_RADIX != 0) return launch_tile_mode<T, 11>(mode, a, s); else break;
#if MFFT_BIG_RADIX      // complex64 2048 and 4096: measured slower than three passes everywhere; kept for the measurement
    case 12: if constexpr (sizeof(T) == 4) return launch_tile_mode<T, 12>(mode, a, s); else break;
#endif
    }
    set_error("tile radix outside 16..1024 (complex128: ..2048; complex64 tuning builds: ..4096)");
    return MFFT_E_UNSUPPORTED;
}

template <typename T> int tile_log2c_for(int log2g, int narrow_bits)
{
    const bool narrow = (narrow_bits & 1) != 0;
    switch (log2g) {
    case 4: return narrow ? TileCols<T, 4, true>::LOG2C : TileCols<T, 4, false>::LOG2C;
    case 5: return narrow ? TileCols<T, 5, true>::LOG2C : TileCols<T, 5, false>::LOG2C;
    case 6: return narrow ? TileCols<T, 6, true>::LOG2C : TileCols<T, 6, false>::LOG2C;
    case 7: return narrow ? TileCols<T, 7, true>::LOG2C : TileCols<T, 7, false>::LOG2C;
    case 8: return narrow ? TileCols<T, 8, true>::LOG2C : TileCols<T, 8, false>::LOG2C;
    case 9: return narrow ? TileCols<T, 9, true>::LOG2C : TileCols<T, 9, false>::LOG2C;
    case 10: return (sizeof(T) == 8 && (narrow_bits & 2)) ? 2 : TileCols<T, 10, false>::LOG2C;
    case 11: case 12: return 2;
    }
    return 0;
}

// the radices / table shape of a tile's column transforms, for the host code that builds their twiddle tables
template <typename T, int LOG2G> void tile_geometry_fill(LdsGeometry* g)
{
    using RD = typename TileCfg<T, LOG2G>::RD;
    *g = LdsGeometry{};
    g->n = 1 << LOG2G;
    g->p = TileCfg<T, LOG2G>::P;
    g->tx = TileCfg<T, LOG2G>::TX;
    g->npass = RD::npass;
    for (int i = 0; i < 4; ++i) g->radix[i] = RD::r[i];
    g->split16 = RD::split16 ? 1 : 0;
    g->tw_elems = (size_t)tw_total_elems<RD>();
}
template <typename T> bool tile_geometry_for(int log2g, LdsGeometry* g)
{
    switch (log2g) {
#define MFFT_TG(L) case L: tile_geometry_fill<T, L>(g); return true;
        MFFT_TG(4) MFFT_TG(5) MFFT_TG(6) MFFT_TG(7) MFFT_TG(8) MFFT_TG(9) MFFT_TG(10) MFFT_TG(11) MFFT_TG(12)
#undef MFFT_TG
    }
    return false;
}

}  // namespace mfft
