// Host-side plumbing shared by the translation units of libmfft.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/mfft.h"
#include "fft_core.h"

namespace mfft {

void set_error(const std::string& msg);

// position of element j under a segmented layout (seg == 0: dense)
__host__ __device__ inline int64_t seg_index(int64_t j, int64_t seg, int64_t seg_stride)
{
    return seg > 0 ? (j / seg) * seg_stride + (j % seg) : j;
}
// the same for a power-of-two segment given as a shift (lseg = 30 describes a dense signal)
__host__ __device__ inline int64_t seg_pos(int64_t j, int lseg, int64_t seg_stride)
{
    return (j >> lseg) * seg_stride + (j & ((int64_t(1) << lseg) - 1));
}
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define MFFT_HIP(call)                                                        \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return ::mfft::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// exp(-2*pi*i*k/n) in long double with exact quarter-turn reduction.
void unit_root(int64_t k, int64_t n, long double* re, long double* im);

inline int ilog2(int64_t n)
{
    int l = 0;
    while ((int64_t(1) << l) < n) ++l;
    return l;
}
inline bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// ---------------------------------------------------------------------------
// Arguments of the LDS-resident Stockham kernels (one struct for all variants;
// fields a variant does not use are ignored).
// ---------------------------------------------------------------------------
template <typename T> struct KArgs {
    const void* in;        // complex (or real when real_in) input
    void* out;             // complex output, or real power spectrum
    const cx<T>* tw;       // per-pass twiddle tables, concatenated
    const T* win;          // real window of n_in samples, or nullptr
    const cx<T>* pre;      // complex premultiplier (window*demod*chirp) of n_in samples, or nullptr
    const cx<T>* post;     // complex postmultiplier of n_out bins, applied after conj_out, or nullptr
    double* stats;         // 3 doubles per transform, or nullptr
    int64_t batch;
    int64_t in_stride;     // elements between consecutive signals in `in`
    int64_t out_stride;    // elements between consecutive spectra in `out`
    // Segmented layouts (general variants only; seg == 0 means dense): element j of a signal
    // sits at (j / seg) * seg_stride + (j % seg).  This is how pencil / four-step exchanges
    // are packed and unpacked inside the transform's own load and store.
    int64_t in_seg, in_seg_stride;
    int64_t out_seg, out_seg_stride;
    int n_in;              // samples present per signal (<= N); the rest are zero
    int n_out;             // bins stored per signal (<= N)
    int real_in;           // input elements are real
    int conj_in;           // conjugate on load  (inverse transform = conj . forward . conj)
    int conj_out;          // conjugate on store
    T scale;
    // V_SEG: power-of-two segments as shifts.  Element j sits at (j >> lseg) * seg_stride +
    // (j & ((1 << lseg) - 1)); lseg = 30 describes a dense signal.
    int in_lseg, out_lseg;
    int grid_over;         // host only: resident sets of workgroups to launch, 0 = the measured default (lds_grid_over)
    // Set by the launcher: workgroups below `resident` stride (by `resident`) over the groups below tail_start,
    // each further one takes tail_chunk consecutive groups from tail_start on.
    int64_t tail_start;
    int resident, tail_chunk;
    int tail_pct, tail_chunk_req;      // host only (tuning aid): share of the batch left to the tail (-1 none, 0 the kernel's default), groups per tail workgroup
};

// Segments the plain segmented variant (V_SEG) serves without staging: powers of two up to 2^30 that hold
// whole rows of a signal's n / p elements (p points per thread), or -- shorter -- at least one wave's 64
// elements and one 1 KiB piece of the LDS-DMA loads.
inline bool seg_is_fast(int64_t seg, int64_t n, int points_per_thread, int dtype)
{
    if (seg == 0) return true;
    if (!is_pow2(seg) || seg > (int64_t(1) << 30)) return false;
    return seg >= n / points_per_thread || seg >= (dtype == MFFT_C64 ? 128 : 64);
}

// kernel variants
enum Variant {
    V_PLAIN = 0,       // complex in (full length, contiguous) -> scaled complex out
    V_HANN_POWER = 1,  // complex in * register-resident window -> power out   (BASELINE config 2)
    V_GEN_CPLX = 2,    // general prologue (pad / window / premultiplier / real / conj) -> complex out
    V_GEN_POWER = 3,   // general prologue -> power out (+ optional per-signal stats)
    V_SEG = 4,         // V_PLAIN over power-of-two segmented layouts (exchange packing of the multi-GPU paths)
    V_WIN = 5,         // complex in * register-resident window -> scaled complex out
    V_POW = 6,         // complex in -> power out
    V_REAL = 7,        // real in (full length, contiguous) -> scaled complex out
    V_HANN_POWER_STATS = 8,   // V_HANN_POWER + per-signal max / sum / centroid (the apps' own flag set)
    V_POW_STATS = 9,   // V_POW + the same statistics
    V_COUNT = 10
};

// Description of how one power-of-two length is decomposed inside the LDS kernel.
struct LdsGeometry {
    int n, p, tx, block, slots, npass, radix[4];
    int split16;           // radix-16 passes store six twiddle generators instead of fifteen powers
    size_t lds_bytes;
    size_t tw_elems;       // total twiddle-table entries
};

// Largest single-kernel lengths
constexpr int kMaxLdsLog2C64 = 14;   // 16384 points, 128 KiB + padding
constexpr int kMaxLdsLog2C128 = 13;  // 8192 points

bool lds_geometry(int dtype, int log2n, LdsGeometry* g);

// Launch the LDS kernel for 2^log2n points.  grid <= 0 lets the launcher pick a
// persistent grid from the device's CU count.
int launch_lds_c64(int log2n, int variant, const KArgs<float>& a, int grid, hipStream_t s);
int launch_lds_c128(int log2n, int variant, const KArgs<double>& a, int grid, hipStream_t s);

// Fill a host vector with the per-pass twiddle tables of the LDS kernel.
void build_lds_twiddles(const LdsGeometry& g, std::vector<long double>* re, std::vector<long double>* im);

int device_cu_count();

// ---------------------------------------------------------------------------
// Lane-distributed kernels for the shortest lengths, see wave_kernel.h
// ---------------------------------------------------------------------------
template <typename T> struct WaveArgs;
constexpr int kWaveMinLog2C64 = 2, kWaveMaxLog2C64 = 7;     // 4..128 points
constexpr int kWaveMinLog2C128 = 1, kWaveMaxLog2C128 = 6;   // 2..64 points
int launch_wave_c64(int log2n, bool win, bool power, bool real_in, const WaveArgs<float>& a, hipStream_t s);
int launch_wave_c128(int log2n, bool win, bool power, bool real_in, const WaveArgs<double>& a, hipStream_t s);

// ---------------------------------------------------------------------------
// Arguments of the global-pass (tile) kernels, see tile_kernel.h
// ---------------------------------------------------------------------------
template <typename T> struct TArgs;
enum { kTileMinLog2G = 4, kTileMaxLog2G = 9, kTileMaxLog2GNarrow = 12 };   // 10..12: four-column tiles (complex128: 1024 x 8 and 2048 x 4)
int launch_tile_c64(int log2g, int mode, const TArgs<float>& a, hipStream_t s);
int launch_tile_c128(int log2g, int mode, const TArgs<double>& a, hipStream_t s);
int launch_tile_r2c_c64(int log2g, const TArgs<float>& a, hipStream_t s);     // last pass of a packed real signal's half-length
int launch_tile_r2c_c128(int log2g, const TArgs<double>& a, hipStream_t s);   // transform, split into the n bins fused (tile_r2c_kernel.h)
int tile_log2c_c64(int log2g, int narrow);    // log2 of the columns per tile for this radix and width class (TArgs::narrow bits)
int tile_log2c_c128(int log2g, int narrow);
bool tile_geometry_c64(int log2g, LdsGeometry* g);    // radices and table shape of a tile's column transforms
bool tile_geometry_c128(int log2g, LdsGeometry* g);
bool tile_first_hann_c64();                    // is the first pass that evaluates the Hann window (TM_FIRST_HANN) built?
bool tile_first_hann_c128();

}  // namespace mfft
