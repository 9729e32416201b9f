// Host side of libmfft.so: the C ABI declared in include/mfft.h, plan
// construction (twiddle / window tables, workspace) and the small element-wise
// kernels that surround the transform.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <new>

#include "common.h"
#include "tile_kernel.h"
#include "wave_kernel.h"

namespace mfft {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
    return e == hipErrorOutOfMemory ? MFFT_E_NOMEM : MFFT_E_HIP;
}

// ---------------------------------------------------------------------------
// exact-reduction roots of unity
// ---------------------------------------------------------------------------
void unit_root(int64_t k, int64_t n, long double* re, long double* im)
{
    static const long double two_pi = 6.283185307179586476925286766559005768L;
    k %= n;
    if (k < 0) k += n;
    const __int128 k4 = (__int128)4 * k;
    const int64_t q = (int64_t)((k4 + n / 2) / n);          // nearest quarter turn
    const int64_t rem = (int64_t)(k4 - (__int128)q * n);    // |rem| <= n/2
    const long double a = two_pi * (long double)rem / (4.0L * (long double)n);
    const long double ca = cosl(a), sa = sinl(a);
    long double c, s;
    switch (q & 3) {
    case 0: c = ca; s = sa; break;
    case 1: c = -sa; s = ca; break;
    case 2: c = -ca; s = -sa; break;
    default: c = sa; s = -ca; break;
    }
    *re = c;
    *im = -s;   // exp(-i*theta)
}

void build_lds_twiddles(const LdsGeometry& g, std::vector<long double>* re, std::vector<long double>* im)
{
    re->clear();
    im->clear();
    int ns = 1;
    for (int s = 0; s < g.npass; ++s) {
        const int r = g.radix[s];
        if (s > 0) {
            // powers stored per pass: all of 1..r-1, or the generators of a radix-16 / radix-32 pass
            const int rows = tw_rows(r, g.split16 != 0);
            for (int row = 0; row < rows; ++row) {
                const int i = (rows == r - 1) ? row + 1 : tw_split_power(row);
                for (int m = 0; m < ns; ++m) {
                    long double c, d;
                    unit_root((int64_t)m * i, (int64_t)ns * r, &c, &d);
                    re->push_back(c);
                    im->push_back(d);
                }
            }
        }
        ns *= r;
    }
}

int device_cu_count()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        cached = p.multiProcessorCount;
        cached_dev = dev;
    }
    return cached;
}

// Is `s` recording into a graph (hipStreamBeginCapture by the caller, torch.cuda.graph)?  Nothing may be
// allocated then, and events recorded outside the capture may not be waited for.
static bool stream_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}
static int first_use_alloc(void** ptr, size_t bytes, hipStream_t s)
{
    if (stream_capturing(s)) {
        set_error("this execution needs a buffer the plan allocates on first use: run it once outside stream capture first");
        return MFFT_E_UNSUPPORTED;
    }
    MFFT_HIP(hipMalloc(ptr, bytes));
    return MFFT_OK;
}

int blocks_per_cu(const void* kernel, int block, size_t lds)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    return per_cu;
}

int pick_grid(int per_cu, int64_t groups, int over)
{
    int64_t g = (int64_t)per_cu * device_cu_count() * (over > 0 ? over : 1);
    if (g > groups) g = groups;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mfft

#include "helper_kernels.h"
#include "copy_pool.h"

// Device scratch of the entry point that allocates nothing (mfft_lsq_normal_static): per-block
// partial sums of the normal equations.  One copy per GPU the library runs on, held until the
// process ends.  In host code this name is only a handle for hipGetSymbolAddress -- kernels get
// the per-device address resolved in static_scratch().
constexpr size_t kScratchDoubles = size_t(1) << 18;          // 2 MiB
static __device__ __align__(256) double scratch[kScratchDoubles];

namespace mfft {

// ---------------------------------------------------------------------------
// power-of-two engine: one LDS-resident kernel, or 2-4 global Stockham passes
// ---------------------------------------------------------------------------
static size_t real_size(int dtype) { return dtype == MFFT_C64 ? sizeof(float) : sizeof(double); }

// What surrounds one transform: where the data is and which prologue / epilogue applies.
template <typename T> struct Io {
    const void* in = nullptr;
    void* out = nullptr;
    int64_t in_stride = 0, out_stride = 0;
    int64_t in_seg = 0, in_seg_stride = 0, out_seg = 0, out_seg_stride = 0;
    const T* win = nullptr;
    const cx<T>* pre = nullptr;
    const cx<T>* post = nullptr;
    double* stats = nullptr;
    int64_t n_in = 0, n_out = 0;
    int real_in = 0, conj_in = 0, conj_out = 0;
    int win_pairs = 0;     // `win` weighs the real and the imaginary part of every sample separately (packed real input)
    T scale = 1;
    bool power = false;
    bool r2c = false;      // the transform is the half-length one of a packed real signal: the last pass splits its bins into the
                           // 2 n bins of the signal's own spectrum (or their power) at `out`, signals out_stride elements apart
    int64_t batch = 0;     // > 0: transform only this many signals (a slab of the plan's batch)
};

template <typename T>
static int upload_complex(const std::vector<long double>& re, const std::vector<long double>& im, void** d)
{
    std::vector<cx<T>> h(re.size());
    for (size_t i = 0; i < re.size(); ++i) h[i] = cx<T>{(T)re[i], (T)im[i]};
    const size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(cx<T>);
    MFFT_HIP(hipMalloc(d, bytes));
    if (!h.empty()) MFFT_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    return MFFT_OK;
}
static int upload_complex_dt(int dtype, const std::vector<long double>& re, const std::vector<long double>& im,
                             void** d)
{
    return dtype == MFFT_C64 ? upload_complex<float>(re, im, d) : upload_complex<double>(re, im, d);
}

template <typename T> static int upload_real(const std::vector<long double>& v, void** d)
{
    std::vector<T> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = (T)v[i];
    MFFT_HIP(hipMalloc(d, (h.empty() ? 1 : h.size()) * sizeof(T)));
    if (!h.empty()) MFFT_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return MFFT_OK;
}

struct GlobalPass {
    int log2g = 0;
    int64_t ns = 1;          // product of earlier radices
    int lb = 0;              // low-table bits
    void* d_tw = nullptr;    // inner tables of the G-point DFT
    void* d_lo = nullptr;    // exp(-2 pi i k / M), k < 2^lb           (ns > 1 only; always double)
    void* d_hi = nullptr;    // exp(-2 pi i (k << lb) / M), k < M >> lb
};

struct Pow2Engine {
    int64_t n = 0, batch = 0;
    int dtype = 0, log2n = 0;
    bool single = true;
    LdsGeometry geo{};
    void* d_tw = nullptr;               // single-kernel twiddles
    void* d_wave_tw = nullptr;          // W_N^j, j < N/2, for the lane-distributed kernel (shortest lengths)
    std::vector<GlobalPass> passes;     // tiled decomposition
    void* work[2] = {nullptr, nullptr};
    size_t work_bytes = 0;
    void* stage_in = nullptr;           // dense copies around a general single-kernel transform whose
    void* stage_out = nullptr;          // caller layout is segmented (allocated on first use)
    void* r2c_side = nullptr;           // fused real-input last pass: raw outputs of the two self-mirrored columns (first use)
    int64_t bytes = 0;                  // device memory held
    int tile_nt = 0;                    // TArgs::nt of the plain tile passes
    int tile_nt_pass[4] = {-1, -1, -1, -1};   // per-pass value where it differs from tile_nt (-1: tile_nt)
    int narrow = 0;                     // tile width bits (TArgs::narrow)
    int tail_pct = 0, tail_chunk = 0;   // MFFT_LDS_TAIL=pct,chunk at build (tuning aid): share of the batch the tail workgroups of fft_lds_tail_kernel get
                                        // (negative: none, 0: per length) and the groups each of them takes
    int lds_over = 0;                   // MFFT_LDS_GRID_OVER at build (tuning aid): the same for the single-kernel transforms, 0 = per length and variant
    int grid_over = 4;                  // MFFT_TILE_GRID_OVER at build (tuning aid): resident sets of workgroups a non-prefetching tile pass is launched with
    int tile_buf = -1;                  // MFFT_TILE_BUF at build (tuning aid): TArgs::buf of every pass; -1 = by tile width and data size
    bool table_window = false;          // MFFT_TABLE_WINDOW at build: windowed first passes read the window table (tuning aid)
    int64_t inner = 1;                  // column-batched: transform along the middle axis of (batch, n, inner)
    int64_t fs_n = 0, fs_c0 = 0;        // four-step twiddle exp(-2 pi i k (c0 + c) / fs_n) on the output, 0 = none
    int64_t out_ld = 0;                 // column-batched: elements between rows of the result (0 = inner, dense)

    int build(int64_t n_, int64_t batch_, int dtype_, int64_t inner_ = 1, int64_t fs_n_ = 0, int64_t fs_c0_ = 0);
    void release();
    int launches() const { return single ? 1 : (int)passes.size(); }
    // exec() derives the statistics of a dense power spectrum from a reduction after the last pass
    // (and so overwrites all of them) whenever the transform takes several passes
    bool stats_by_reduction(bool power, int64_t out_seg) const { return !single && power && out_seg == 0 && work[0] != nullptr; }
    // can the last pass carry the split of a packed real signal?  (tile passes, two tiles or more per signal at the last radix)
    bool r2c_fusable() const
    {
        if (single || inner != 1 || passes.size() < 2) return false;
        const int lg = passes.back().log2g;
        if (dtype == MFFT_C64 && lg > 10) return false;
        return log2n - lg - tile_log2c_of(lg) >= 1;
    }
    int tile_log2c_of(int log2g) const;
    template <typename T> int exec(const Io<T>& io, hipStream_t s);
    template <typename T> int reduce_stats(const T* S, int64_t n_out, int64_t stride, int64_t nb, double* stats, hipStream_t s);
};

static int tile_log2c(int dtype, int log2g, int narrow)
{
    return dtype == MFFT_C64 ? tile_log2c_c64(log2g, narrow) : tile_log2c_c128(log2g, narrow);
}

int Pow2Engine::tile_log2c_of(int log2g) const { return tile_log2c(dtype, log2g, narrow); }

// How a length above one LDS residency is split into global passes (log2 of each radix, first
// pass first).  Fewest passes first; within that the radices are balanced, larger ones first.
// MFFT_SPLIT="8,8,8" overrides the choice (tuning aid; must multiply to the length).
static int choose_split(int log2n, int dtype, std::vector<int>* out, int log2_inner = 0, int narrow = 0, int64_t batch = 1)
{
    out->clear();
    if (const char* env = getenv("MFFT_SPLIT")) {
        int sum = 0;
        for (const char* p = env; *p;) {
            const int v = (int)strtol(p, const_cast<char**>(&p), 10);
            if (v < kTileMinLog2G || v > (!log2_inner ? (dtype == MFFT_C64 ? kTileMaxLog2GNarrow : 11) : kTileMaxLog2G)) break;
            out->push_back(v);
            sum += v;
            if (*p == ',') ++p;
        }
        if (sum == log2n && out->size() >= (log2_inner ? 1u : 2u) && out->size() <= 4) return MFFT_OK;
        out->clear();
    }
    // 2^19 points: 1024 x 512 in two passes (four-column first pass, XCD-aware order) measured 107 us per
    // 2^24 complex64 points against 156 us for three passes, and 110 against 135 us per 2^23 complex128
    // points (the chirp-z length of the reference apps' default transform); the larger narrow radices,
    // and 1024 x 1024 for complex128 (152 against 135 us), lost.  A single complex128 signal is 128 of those
    // 256-thread tiles, half the CUs: it keeps three passes (22.7 against 23.6 us plain, but 60 against 56 us
    // inside the apps' chirp-z).
    if (!log2_inner && log2n == 19 && (dtype == MFFT_C64 || batch >= 2) && !getenv("MFFT_NO_NARROW_RADIX")) {
        out->push_back(10);
        out->push_back(9);
        return MFFT_OK;
    }
    // One 2^20-point signal (8 / 16 MiB: launch- and latency-bound) takes 1024 x 1024 in two passes: 27.6 -> 21.1 us
    // complex64, 34.3 -> 31.4 us complex128, alternated with the three-pass split inside one process; with two or
    // more signals the preference changes sign from batch to batch (2: +10 %, 4: -3 %, 8: -9 %, 16: +20 %) and
    // three passes stay.
    if (!log2_inner && log2n == 20 && batch == 1 && !getenv("MFFT_NO_NARROW_RADIX")) {
        out->push_back(10);
        out->push_back(10);
        return MFFT_OK;
    }
    // complex128 signals of 2^28 .. 2^31 points: three passes through the 1024 x 8 tiles (128-byte rows, 4.4-4.9
    // TB/s per pass against 5.0-5.3 for the 256-byte rows of the 512 x 16 ones) instead of four through narrower
    // radices -- 2^28 6.82 -> 5.21 ms as 1024 * 512 * 512, 2^29 13.6 -> 10.9 as 1024 * 1024 * 512, 2^30 27.7 -> 24.2
    // as 1024^3; 2^31 needs one 2048 x 4 pass (64-byte rows: 2.7-2.9 TB/s), first, where its output is one
    // contiguous run per tile: 54.5 -> 51.9 ms (2048 in the middle or last: 53.4).  Alternated with the
    // four-pass splits in one process.
    if (!log2_inner && dtype == MFFT_C128 && log2n >= 28 && log2n <= 31 && !getenv("MFFT_FOUR_PASSES")) {
        const int third = log2n - 20;                     // 8 .. 11
        if (third <= 10) {                                // (10, 9, 9), (10, 10, 9), (10, 10, 10)
            out->push_back(10);
            out->push_back(third >= 9 ? 10 : 9);
            out->push_back(third >= 10 ? 10 : 9);
        } else {
            out->push_back(11);
            out->push_back(10);
            out->push_back(10);
        }
        return MFFT_OK;
    }
    const int maxg = kTileMaxLog2G;
    const int npass = (log2n + maxg - 1) / maxg;
    if (npass > 4) { set_error("length above the four-pass range"); return MFFT_E_UNSUPPORTED; }
    for (int s = 0; s < npass; ++s) out->push_back(log2n / npass + (s < log2n % npass ? 1 : 0));
    // complex64 column-batched transforms of 2^12..2^16 points: a 256-point last pass (32-column tiles) and the
    // rest first -- alternated with the balanced split inside one process, us per transform: 16384 x 16384
    // (BASELINE config 4's column step) 1820 -> 1720-1776, 16384 x 2048 222 -> 210, 8192 x 16384 870 -> 846,
    // 32768 x 8192 1790 -> 1766, 4096 x 16384 426 -> 415; 512-point passes lose there (128-byte rows).
    if (log2_inner && !narrow && dtype == MFFT_C64 && log2n >= 12 && log2n <= 16 && !getenv("MFFT_BALANCED_COLUMNS")) {
        out->clear();
        out->push_back(log2n - 8);
        out->push_back(8);
    }
    for (size_t s = 0; s < out->size(); ++s) {
        const int g = (*out)[s];
        if (g < kTileMinLog2G) { set_error("tile radix too small"); return MFFT_E_UNSUPPORTED; }
        // later passes need NS >= columns per tile; the tile count L/C must be >= 1
        if (log2_inner) {
            // column-batched: a tile is C adjacent columns of one butterfly
            if (log2_inner < tile_log2c(dtype, g, narrow)) { set_error("fewer columns than a tile is wide"); return MFFT_E_UNSUPPORTED; }
        } else if (log2n - g < tile_log2c(dtype, g, narrow)) { set_error("tile wider than the pass"); return MFFT_E_UNSUPPORTED; }
    }
    if (!log2_inner && out->size() >= 2 && (*out)[0] < tile_log2c(dtype, (*out)[1], narrow)) { set_error("first radix below tile width"); return MFFT_E_UNSUPPORTED; }
    return MFFT_OK;
}

int Pow2Engine::build(int64_t n_, int64_t batch_, int dtype_, int64_t inner_, int64_t fs_n_, int64_t fs_c0_)
{
    n = n_; batch = batch_; dtype = dtype_; inner = inner_; fs_n = fs_n_; fs_c0 = fs_c0_;
    log2n = ilog2(n);
    table_window = getenv("MFFT_TABLE_WINDOW") != nullptr;
    if (const char* e = getenv("MFFT_TILE_BUF")) tile_buf = atoi(e);
    if (const char* e = getenv("MFFT_LDS_TAIL")) { if (sscanf(e, "%d,%d", &tail_pct, &tail_chunk) != 2) tail_pct = tail_chunk = 0; }
    if (const char* e = getenv("MFFT_LDS_GRID_OVER")) lds_over = atoi(e) < 0 ? 0 : atoi(e) > 64 ? 64 : atoi(e);
    if (const char* e = getenv("MFFT_TILE_GRID_OVER")) grid_over = atoi(e) < 1 ? 1 : atoi(e) > 64 ? 64 : atoi(e);
    if (!is_pow2(n) || n < 2) { set_error("engine length must be a power of two >= 2"); return MFFT_E_INVALID; }
    if (!is_pow2(inner)) { set_error("the inner extent of a column-batched transform must be a power of two"); return MFFT_E_UNSUPPORTED; }
    const int maxlog = dtype == MFFT_C64 ? kMaxLdsLog2C64 : kMaxLdsLog2C128;
    std::vector<long double> re, im;
    if (log2n <= maxlog && inner == 1) {
        single = true;
        if (!lds_geometry(dtype, log2n, &geo)) { set_error("no single-kernel geometry"); return MFFT_E_UNSUPPORTED; }
        build_lds_twiddles(geo, &re, &im);
        bytes += (int64_t)re.size() * 2 * (int64_t)real_size(dtype);
        int rc = upload_complex_dt(dtype, re, im, &d_tw);
        if (rc) return rc;
        const bool wave_ok = dtype == MFFT_C64 ? (log2n >= kWaveMinLog2C64 && log2n <= kWaveMaxLog2C64)
                                               : (log2n >= kWaveMinLog2C128 && log2n <= kWaveMaxLog2C128);
        if (wave_ok && !getenv("MFFT_NO_WAVE")) {
            re.resize((size_t)(n / 2)); im.resize((size_t)(n / 2));
            for (int64_t j = 0; j < n / 2; ++j) unit_root(j, n, &re[(size_t)j], &im[(size_t)j]);
            rc = upload_complex_dt(dtype, re, im, &d_wave_tw);
        }
        return rc;
    }
    single = false;
    std::vector<int> split;
    const int log2_inner = ilog2(inner);
    const bool try_wide = !(inner > 1 && getenv("MFFT_NARROW_COLUMNS"));   // tuning aid: narrow tiles for every column-batched plan
    int rc0 = try_wide ? choose_split(log2n, dtype, &split, log2_inner, 0, batch) : MFFT_E_UNSUPPORTED;
    narrow = 0;
    if (rc0 && inner > 1) {             // fewer columns than a wide tile: the narrow width
        narrow = 1;
        rc0 = choose_split(log2n, dtype, &split, log2_inner, 1, batch);
    }
    if (rc0) return rc0;
    // complex128 1024-point tiles: four columns while the plan's data stays in the Infinity Cache (there the 64-byte
    // rows cost nothing and twice as many tiles fill the CUs: 2^19 x 8 plain 61 against 70 us, 2^20 x 1 Hann + power
    // 32.4 against 35.1), eight columns -- a full line per row -- beyond it
    if (dtype == MFFT_C128 && inner == 1 && (size_t)n * (size_t)batch * 16 <= (size_t(384) << 20)) narrow |= 2;
    const int npass = (int)split.size();
    int64_t ns = inner;                 // counts elements: the columns are an innermost, finished "pass"
    for (int s = 0; s < npass; ++s) {
        GlobalPass gp;
        gp.log2g = split[(size_t)s];
        gp.ns = ns;
        LdsGeometry g{};
        if (!(dtype == MFFT_C64 ? tile_geometry_c64(gp.log2g, &g) : tile_geometry_c128(gp.log2g, &g))) { set_error("no geometry for tile radix"); return MFFT_E_UNSUPPORTED; }
        build_lds_twiddles(g, &re, &im);
        int rc = upload_complex_dt(dtype, re, im, &gp.d_tw);
        if (rc) { passes.push_back(gp); return rc; }
        bytes += (int64_t)re.size() * 2 * (int64_t)real_size(dtype);
        if (s > 0 || inner > 1) {
            const int64_t m = (ns / inner) << gp.log2g;
            const int log2m = ilog2(m);
            gp.lb = log2m < 8 ? log2m : 8;
            const int64_t lo_n = int64_t(1) << gp.lb, hi_n = m >> gp.lb;
            re.resize((size_t)lo_n); im.resize((size_t)lo_n);
            for (int64_t k = 0; k < lo_n; ++k) unit_root(k, m, &re[(size_t)k], &im[(size_t)k]);
            rc = upload_complex<double>(re, im, &gp.d_lo);
            if (rc) { passes.push_back(gp); return rc; }
            re.resize((size_t)hi_n); im.resize((size_t)hi_n);
            for (int64_t k = 0; k < hi_n; ++k) unit_root(k << gp.lb, m, &re[(size_t)k], &im[(size_t)k]);
            rc = upload_complex<double>(re, im, &gp.d_hi);
            if (rc) { passes.push_back(gp); return rc; }
            bytes += (lo_n + hi_n) * 2 * (int64_t)sizeof(double);
        }
        passes.push_back(gp);
        ns <<= gp.log2g;
    }
    if (fs_n > 0 && (!is_pow2(fs_n) || inner == 1)) { set_error("the fused twiddle needs a power-of-two modulus on a column-batched plan"); return MFFT_E_UNSUPPORTED; }
    if (fs_n > (int64_t(1) << 40)) { set_error("fused twiddle modulus above 2^40"); return MFFT_E_UNSUPPORTED; }   // tile_root's integer reduction
    if (fs_n == 1) fs_n = 0;            // exp(-2 pi i k c / 1) = 1: nothing to multiply by
    const size_t wbytes = (size_t)n * (size_t)inner * (size_t)batch * 2 * real_size(dtype);
    // Cache policy of the plain tile passes, measured on MI355X with every per-pass combination
    // (2^16..2^24 points, 16 MiB..1 GiB per launch, both precisions): up to about the Infinity
    // Cache's size the best or tied-best choice is streaming loads everywhere, temporal stores in
    // every pass but the last (the next pass finds its input on chip) and streaming stores in the
    // last one -- 4..16 % over plain accesses on two-pass lengths, 3..9 % over fully streaming
    // passes on three-pass ones; beyond that size fully streaming passes win by 1..5 %.
    tile_nt = 3;
    if (inner == 1 && wbytes <= (size_t(384) << 20))
        for (size_t p = 0; p + 1 < passes.size() && p < 4; ++p) tile_nt_pass[p] = 1;
    // A last pass of 512-point complex128 tiles (the prefetching kernels) keeps its stores temporal at
    // every size: its 256-byte rows, written 8 MiB apart, merge in L2 -- 2^27 points 2.68 -> 2.49 ms, the
    // 2^18 x 1024 column step of an eight-rank 2^31-point transform 3.80 -> 3.42 ms, 2^18 x 8192 unchanged
    // (alternated inside one process; 256-point last passes measured equal or worse and keep streaming).
    // The 1024-point tiles likewise (128-byte rows 16 MiB apart): 2^30 points as 1024^3 26.7 -> 24.2 ms, 2^28 as
    // 512 * 512 * 1024 6.4 -> 5.3 ms.
    if (dtype == MFFT_C128 && !passes.empty() && passes.size() <= 4 && passes.back().log2g >= 9 && wbytes > (size_t(384) << 20))
        tile_nt_pass[passes.size() - 1] = 1;
    if (const char* env = getenv("MFFT_TILE_NT")) {                            // tuning aid
        tile_nt = atoi(env) & 3;
        for (int& v : tile_nt_pass) v = -1;
    }
    if (const char* env = getenv("MFFT_TILE_NT_PASS")) {                       // tuning aid: "3,0,2"
        int k = 0;
        for (const char* c = env; *c && k < 4; ++c) if (*c >= '0' && *c <= '3') tile_nt_pass[k++] = *c - '0';
    }
    // One workspace: the passes alternate between it and the caller's output buffer (see exec); a
    // second one is allocated on first use by the executions that cannot do that.
    work_bytes = wbytes;
    if (passes.size() > 1) {
        MFFT_HIP(hipMalloc(&work[0], wbytes));
        bytes += (int64_t)wbytes;
    }
    return MFFT_OK;
}

void Pow2Engine::release()
{
    if (d_tw) (void)hipFree(d_tw);
    if (d_wave_tw) (void)hipFree(d_wave_tw);
    d_wave_tw = nullptr;
    for (auto& gp : passes)
        for (void* q : {gp.d_tw, gp.d_lo, gp.d_hi})
            if (q) (void)hipFree(q);
    for (void* q : work)
        if (q) (void)hipFree(q);
    for (void* q : {stage_in, stage_out, r2c_side})
        if (q) (void)hipFree(q);
    stage_in = stage_out = r2c_side = nullptr;
    d_tw = nullptr; work[0] = work[1] = nullptr; passes.clear();
}

template <typename T> int Pow2Engine::exec(const Io<T>& io, hipStream_t s)
{
    const int64_t nb = io.batch > 0 ? io.batch : batch;
    if (single && io.win_pairs) { set_error("pair windows exist in the tile passes only"); return MFFT_E_UNSUPPORTED; }
    if (single) {
        KArgs<T> a{};
        a.in = io.in; a.out = io.out;
        a.tw = static_cast<const cx<T>*>(d_tw);
        a.win = io.win; a.pre = io.pre; a.post = io.post; a.stats = io.stats;
        a.batch = nb;
        a.in_stride = io.in_stride; a.out_stride = io.out_stride;
        a.in_seg = io.in_seg; a.in_seg_stride = io.in_seg_stride;
        a.out_seg = io.out_seg; a.out_seg_stride = io.out_seg_stride;
        a.n_in = (int)io.n_in; a.n_out = (int)io.n_out;
        a.real_in = io.real_in; a.conj_in = io.conj_in; a.conj_out = io.conj_out;
        a.scale = io.scale;
        a.grid_over = lds_over;
        a.tail_pct = tail_pct; a.tail_chunk_req = tail_chunk;
        const bool full = io.n_in == n && io.n_out == n && io.in_stride == n && io.out_stride == n &&
                          io.in_seg == 0 && io.out_seg == 0;
        const bool bare_or_stats = !io.pre && !io.post && !io.real_in && !io.conj_in && !io.conj_out && full;
        const bool bare = bare_or_stats && !io.stats;
        const bool seg_ok = seg_is_fast(io.in_seg, n, geo.p, dtype) && seg_is_fast(io.out_seg, n, geo.p, dtype);
        const bool bare_seg = !io.pre && !io.post && !io.real_in && !io.conj_in && !io.conj_out && !io.stats &&
                              !io.win && !io.power && io.n_in == n && io.n_out == n && seg_ok;
        int variant;
        if (io.power && io.win && bare) variant = V_HANN_POWER;
        else if (io.power && io.win && bare_or_stats) variant = V_HANN_POWER_STATS;
        else if (io.win && bare) variant = V_WIN;
        else if (io.power && bare) variant = V_POW;
        else if (io.power && bare_or_stats) variant = V_POW_STATS;
        // plain real -> complex: its own fast variant where that measured faster than the general one
        // (complex64, 256..4096 points: +4..25 %; shorter lengths and complex128 measured the same or slower)
        else if (io.real_in && !io.power && !io.win && !io.pre && !io.post && !io.conj_in && !io.conj_out && !io.stats && full &&
                 sizeof(T) == 4 && log2n >= 8 && log2n <= 12)
            variant = V_REAL;
        // the largest lengths run faster through V_SEG's scalar-offset addressing even when dense
        // (16384 complex64: 260 -> 237 us per 2^26 points)
        else if (!io.power && !io.win && bare && log2n < 13) variant = V_PLAIN;
        else if (bare_seg) {
            variant = V_SEG;
            a.in_lseg = io.in_seg ? ilog2(io.in_seg) : 30;     // 30: one segment covers any signal
            a.out_lseg = io.out_seg ? ilog2(io.out_seg) : 30;
        }
        else variant = io.power ? V_GEN_POWER : V_GEN_CPLX;
        const bool bare_real = io.real_in && !io.power && !io.win && !io.pre && !io.post && !io.conj_in && !io.conj_out &&
                               !io.stats && full;
        if ((bare || bare_real) && d_wave_tw) {
            // shortest lengths: the signal lives across lanes (DPP / swizzle exchanges, 16 B per lane)
            WaveArgs<T> w{};
            w.in = io.in;
            w.out = io.out;
            w.tw = static_cast<const cx<T>*>(d_wave_tw);
            w.win = io.win;
            w.units = nb * n * (int64_t)sizeof(cx<T>) / 16;
            w.scale = io.scale;
            w.grid_over = lds_over;
            if constexpr (sizeof(T) == 4) return launch_wave_c64(log2n, io.win != nullptr, io.power, io.real_in != 0, w, s);
            else return launch_wave_c128(log2n, io.win != nullptr, io.power, io.real_in != 0, w, s);
        }
        const bool general = variant == V_GEN_POWER || variant == V_GEN_CPLX;
        const bool stage_i = general && io.in_seg > 0, stage_o = general && io.out_seg > 0;
        const size_t in_esz = io.real_in ? sizeof(T) : sizeof(cx<T>), out_esz = io.power ? sizeof(T) : sizeof(cx<T>);
        const auto seg_copy = [&](const void* src, void* dst, size_t esz, int64_t count, int64_t bstride, int64_t sg,
                                  int64_t sgs, int to_dense) -> int {
            const int64_t total = nb * count;
            int64_t blocks = (total + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            if (blocks < 1) return MFFT_OK;
            if (esz == 4) hipLaunchKernelGGL(seg_copy_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, (float*)dst, nb, count, bstride, sg, sgs, to_dense);
            else if (esz == 8) hipLaunchKernelGGL(seg_copy_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const double*)src, (double*)dst, nb, count, bstride, sg, sgs, to_dense);
            else hipLaunchKernelGGL(seg_copy_kernel<double2>, dim3((unsigned)blocks), dim3(256), 0, s, (const double2*)src, (double2*)dst, nb, count, bstride, sg, sgs, to_dense);
            MFFT_HIP(hipGetLastError());
            return MFFT_OK;
        };
        if (stage_i) {
            if (!stage_in) { const int ra = first_use_alloc(&stage_in, (size_t)batch * (size_t)io.n_in * in_esz, s); if (ra) return ra; }
            const int rc = seg_copy(io.in, stage_in, in_esz, io.n_in, io.in_stride, io.in_seg, io.in_seg_stride, 1);
            if (rc) return rc;
            a.in = stage_in; a.in_stride = io.n_in; a.in_seg = a.in_seg_stride = 0;
        }
        if (stage_o) {
            if (!stage_out) { const int ra = first_use_alloc(&stage_out, (size_t)batch * (size_t)io.n_out * out_esz, s); if (ra) return ra; }
            a.out = stage_out; a.out_stride = io.n_out; a.out_seg = a.out_seg_stride = 0;
        }
        int rc;
        if constexpr (sizeof(T) == 4) rc = launch_lds_c64(log2n, variant, a, 0, s);
        else rc = launch_lds_c128(log2n, variant, a, 0, s);
        if (rc) return rc;
        if (stage_o) return seg_copy(stage_out, io.out, out_esz, io.n_out, io.out_stride, io.out_seg, io.out_seg_stride, 0);
        return MFFT_OK;
    }
    const int np = (int)passes.size();
    // Intermediate passes ping-pong between the workspace and the output buffer itself when that is
    // a dense complex array of the transform's size -- pass p writes the output when np - 1 - p is
    // even -- which halves the workspace and keeps one buffer fewer competing for the Infinity
    // Cache.  Not possible when the first pass would overwrite its own input (in-place execution of
    // an odd number of passes) or the output is shorter, strided, segmented or real: two workspaces.
    // (A fused real-input last pass writes 2 n bins or their power: room for the n-element intermediates either way.)
    bool via_out = np >= 3 && ((!io.power && io.out_seg == 0 && io.out_stride == n * inner && io.n_out == n) || io.r2c) &&
                   !(io.in == io.out && (np & 1)) && !getenv("MFFT_WORK2");
    if (np >= 3 && !via_out && !work[1]) {
        const int ra = first_use_alloc(&work[1], work_bytes, s);
        if (ra) return ra;
        bytes += (int64_t)work_bytes;
    }
    if (io.r2c) {
        // the fused real-input last pass: checked, and its side buffer taken, before anything is enqueued
        if (!r2c_fusable() || io.post || io.stats || io.conj_out || io.out_seg) { set_error("no fused real-input pass for this plan"); return MFFT_E_UNSUPPORTED; }
        if (!r2c_side) {
            const size_t sb = (size_t)batch * 2 * (size_t(1) << passes.back().log2g) * sizeof(cx<T>);
            const int ra = first_use_alloc(&r2c_side, sb, s);
            if (ra) return ra;
            bytes += (int64_t)sb;
        }
    }
    // statistics of dense power spectra: a reduction of their own after the last pass (no atomics)
    const bool stats_after = io.stats && io.power && io.out_seg == 0 && work[0] != nullptr;
    const void* src = io.in;
    int64_t src_stride = io.in_stride;
    const int64_t nn = n * inner;       // elements per transform unit (one signal, or one n x inner array)
    const int log2nn = log2n + ilog2(inner);
    for (int p = 0; p < np; ++p) {
        const GlobalPass& gp = passes[(size_t)p];
        const bool first = p == 0, last = p == np - 1;
        TArgs<T> a{};
        a.in = src;
        a.in_stride = src_stride;
        a.out = last ? io.out : (via_out ? (((np - 1 - p) & 1) ? work[0] : io.out) : work[p & 1]);
        a.out_stride = last ? io.out_stride : nn;
        a.out_pitch = (last && out_ld > 0) ? out_ld : inner;
        a.tw = static_cast<const cx<T>*>(gp.d_tw);
        a.tw_lo = static_cast<const cx<double>*>(gp.d_lo);
        a.tw_hi = static_cast<const cx<double>*>(gp.d_hi);
        a.L = nn >> gp.log2g;
        a.NS = gp.ns;
        a.log2_ns = ilog2(gp.ns);
        a.log2_w = ilog2(inner);
        a.log2_tiles = log2nn - gp.log2g - tile_log2c(dtype, gp.log2g, narrow);
        a.narrow = narrow;
        a.batch = nb;
        a.lb = gp.lb;
        a.nt = (p < 4 && tile_nt_pass[p] >= 0) ? tile_nt_pass[p] : tile_nt;
        a.xcd = tile_log2c(dtype, gp.log2g, narrow) <= 2 ? 1 : 0;   // four-column tiles: neighbours share cache lines,
        if (a.xcd) a.nt = 0;                                         // which streaming accesses would not keep in L2
        // buffer-form accesses: four-column tiles always, wider ones while the plan's data is at most 128 MiB
        a.grid_over = grid_over;
        a.buf = tile_buf >= 0 ? tile_buf
                              : ((tile_log2c(dtype, gp.log2g, narrow) <= 2 || (inner == 1 && work_bytes <= (size_t(128) << 20))) ? 12 : 0);
        a.scale = 1;
        if (first) {
            a.win = io.win; a.pre = io.pre; a.n_in = io.n_in;
            a.win_pairs = io.win_pairs;
            a.real_in = io.real_in; a.conj_in = io.conj_in;
            a.in_seg = io.in_seg; a.in_seg_stride = io.in_seg_stride;
        }
        if (last) {
            a.post = io.post; a.stats = stats_after ? nullptr : io.stats; a.n_out = io.n_out;
            a.conj_out = io.conj_out; a.scale = io.scale;
            a.out_seg = io.out_seg; a.out_seg_stride = io.out_seg_stride;
        }
        const bool in_plain = !io.win && !io.pre && !io.real_in && !io.conj_in && io.n_in == n && io.in_seg == 0;
        const bool out_plain = !io.post && !io.stats && !io.conj_out && io.n_out == n && io.out_seg == 0 && !io.power;
        // dense complex samples under the Hann window and / or zero-padded: the window is evaluated, no table read
        const bool in_hann = (io.win || io.n_in < n) && !io.win_pairs && !io.pre && !io.real_in && !io.conj_in && io.in_seg == 0 &&
                             io.n_in >= 2 && io.n_in <= n && n * (int64_t)sizeof(cx<T>) < (int64_t(1) << 32) && inner == 1 && !(narrow & 1) &&
                             (sizeof(T) == 8 ? tile_first_hann_c128() : tile_first_hann_c64()) && !table_window;
        // dense complex samples times at most one table, zero-padded: the lean prologue (the apps' chirp-z and
        // demodulating plans); real input, conjugation, segments and pair windows keep the general one
        const bool in_pre = !io.real_in && !io.conj_in && io.in_seg == 0 && !io.win_pairs && !(io.win && io.pre) &&
                            io.n_in >= 1 && io.n_in <= n && n <= (int64_t(1) << 31) && !getenv("MFFT_GENERAL_PROLOGUE");
        int mode = first ? (in_plain ? TM_FIRST_PLAIN : (in_hann ? TM_FIRST_HANN : (in_pre ? TM_FIRST_PRE : TM_FIRST_GEN)))
                         : (last ? (io.power ? TM_LAST_POWER : (out_plain ? TM_LAST_PLAIN : TM_LAST_GEN)) : TM_MID);
        if (inner > 1) {
            // column-batched: every pass is a strided-in / strided-out pass (the first one with
            // unit twiddles); prologues and epilogues other than the four-step twiddle do not apply
            if (!in_plain || !out_plain) { set_error("column-batched plans take plain complex data"); return MFFT_E_UNSUPPORTED; }
            mode = last ? (fs_n > 0 ? TM_LAST_TWID : TM_LAST_PLAIN) : TM_MID;
            if (last && fs_n > 0) {
                a.fs_mask = (uint64_t)fs_n - 1;
                a.tw_c0 = fs_c0;
            }
        }
        int rc;
        if (last && io.r2c) {
            a.side = static_cast<cx<T>*>(r2c_side);
            a.r2c_power = io.power ? 1 : 0;
            a.scale = io.scale;
            if constexpr (sizeof(T) == 4) rc = launch_tile_r2c_c64(gp.log2g, a, s);
            else rc = launch_tile_r2c_c128(gp.log2g, a, s);
        } else if constexpr (sizeof(T) == 4) rc = launch_tile_c64(gp.log2g, mode, a, s);
        else rc = launch_tile_c128(gp.log2g, mode, a, s);
        if (rc) return rc;
        src = a.out;
        src_stride = nn;
    }
    if (stats_after) return reduce_stats<T>(static_cast<const T*>(io.out), io.n_out, io.out_stride, nb, io.stats, s);
    return MFFT_OK;
}

// {max, sum, centroid} of `nb` finished power spectra by a two-stage reduction; the pass workspace, dead once
// the last pass has read it, holds the per-block partials
template <typename T> int Pow2Engine::reduce_stats(const T* S, int64_t n_out, int64_t stride, int64_t nb, double* stats, hipStream_t s)
{
    int64_t nblk = (n_out + 4095) / 4096;
    if (nblk > 256) nblk = 256;
    if (!work[0] || (size_t)nb * (size_t)nblk * 3 * sizeof(double) > work_bytes || nb > 65535) { set_error("statistics scratch too small"); return MFFT_E_UNSUPPORTED; }
    double* partial = static_cast<double*>(work[0]);
    hipLaunchKernelGGL(stats_partial_kernel<T>, dim3((unsigned)nblk, (unsigned)nb), dim3(256), 0, s, S, n_out, stride, partial);
    hipLaunchKernelGGL(stats_final_kernel, dim3((unsigned)nb), dim3(64), 0, s, (const double*)partial, (int)nblk, stats);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct Plan {
    int64_t n = 0, n_in = 0, batch = 0;
    int dtype = 0;
    uint32_t flags = 0;
    int device = 0;
    enum Kind { ONE_POINT, POW2, CHIRPZ, COLUMNS } kind = POW2;
    int64_t inner = 1;       // COLUMNS: contiguous columns sharing each transform, data is (batch, n, inner)
    int64_t fs_n = 0, fs_c0 = 0;   // COLUMNS: fused four-step twiddle (0 = none)
    mfft_layout lin{0, 0, 0}, lout{0, 0, 0};   // batch_stride 0 = dense default
    Pow2Engine eng;          // length n (POW2) or the padded length m (CHIRPZ)
    int64_t m = 0;           // chirp-z convolution length
    void* d_win = nullptr;   // real window, n_in samples (MFFT_HANN)
    void* d_pre = nullptr;   // complex premultiplier, n_in samples (DEMOD and/or chirp)
    void* d_chirp = nullptr; // exp(-i pi j^2 / n), j < n            (CHIRPZ)
    void* d_filt = nullptr;  // conj(FFT_m(conj chirp, wrapped)), m   (CHIRPZ)
    void* d_conv = nullptr;  // batch * m complex workspace            (CHIRPZ)
    bool packed_real = false;// POW2 + MFFT_REAL_IN above one LDS residency: the engine transforms n/2 packed samples
    void* d_half = nullptr;  // batch * n/2 complex: that transform's result, split into the n bins by r2c_split_kernel
    // staging for mfft_exec_host
    void* d_in = nullptr;
    void* d_out = nullptr;
    double* d_stats = nullptr;
    // pipelined staging for large mfft_exec_host transfers (built on first use)
    void* pin_in[2] = {nullptr, nullptr};
    void* pin_out[2] = {nullptr, nullptr};
    hipStream_t s_run = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int64_t pipe_slab = 0;        // signals per slab
    char* d_lines = nullptr;      // candidate lists, picked lines, line positions, Gram blocks, time grid (mfft_lines_host)
    char* h_lines = nullptr;      // pinned mirror of the small results in d_lines
    char* h_spec = nullptr;       // pinned landing buffer of the spectra (mfft_lines_host, medium sizes)
    int lines_want = 0;
    bool lines_dirty = true;       // candidate counters need clearing before the next scan
    // mfft_lines_host downloads the (normalised) spectra on a stream of their own while the line search runs
    hipStream_t s_spec = nullptr;
    hipEvent_t ev_spec = nullptr;
    void* d_norm = nullptr;        // normalised copy of the spectra (the line search reads the unnormalised ones meanwhile)
    int64_t* d_peaks = nullptr;   // {max, sum, centroid} | count | candidate indices (mfft_exec_host_peaks)
    int64_t* h_peaks = nullptr;   // pinned mirror of d_peaks
    int64_t peak_cap = 0;
    int64_t workspace = 0;
    int launches = 1;
    std::mutex mu;
    // Executions of a plan that owns per-execution state (pass workspaces, the chirp-z convolution
    // buffer, the demodulation table) are ordered on the device as well: an execution on another
    // stream waits for the previous one's completion event.
    std::mutex order_mu;
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool shares_state = false;     // set at build: executions must be ordered (see plan_exec_ordered)
    hipStream_t s_lines = nullptr; // mfft_lines_host runs on a stream of the plan, so sessions on other plans overlap
    double* d_lsq_part = nullptr;  // per-block partial sums of the line fit's normal equations (mfft_lines_host)
    size_t lsq_part_doubles = 0;
    // samples a line search left on the device (mfft_lines_extract_host) for mfft_lines_gram_host
    bool resident = false;
    double* d_fit = nullptr;       // columns (64) | demodulation frequency (1) | normal equations (2*64*64 + 2*64 + 1) | partial sums
    double* h_fit = nullptr;       // pinned mirror of the first three
};

// symmetric Hann window of m samples, np.hanning's formula, evaluated in long double
static void hann_window(int64_t m, std::vector<long double>* w)
{
    static const long double pi = 3.141592653589793238462643383279502884L;
    w->resize((size_t)m);
    if (m == 1) { (*w)[0] = 1.0L; return; }
    for (int64_t j = 0; j < m; ++j)
        (*w)[(size_t)j] = 0.5L + 0.5L * cosl(pi * (long double)(2 * j - m + 1) / (long double)(m - 1));
    (*w)[0] = 0.0L;
    (*w)[(size_t)m - 1] = 0.0L;
}

static void plan_free(Plan* p)
{
    if (!p) return;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(p->device);
    p->eng.release();
    for (void* q : {p->d_win, p->d_pre, p->d_chirp, p->d_filt, p->d_conv, p->d_half, p->d_in, p->d_out, (void*)p->d_stats,
                    (void*)p->d_peaks, (void*)p->d_lines})
        if (q) (void)hipFree(q);
    if (p->s_spec) (void)hipStreamDestroy(p->s_spec);
    if (p->s_lines) (void)hipStreamDestroy(p->s_lines);
    if (p->ev_spec) (void)hipEventDestroy(p->ev_spec);
    if (p->d_norm) (void)hipFree(p->d_norm);
    if (p->d_lsq_part) (void)hipFree(p->d_lsq_part);
    if (p->d_fit) (void)hipFree(p->d_fit);
    if (p->h_fit) (void)hipHostFree(p->h_fit);
    p->s_spec = nullptr; p->s_lines = nullptr; p->ev_spec = nullptr; p->d_norm = nullptr; p->d_lsq_part = nullptr;
    for (void* q : {p->pin_in[0], p->pin_in[1], p->pin_out[0], p->pin_out[1], (void*)p->h_peaks, (void*)p->h_lines, (void*)p->h_spec})
        if (q) (void)hipHostFree(q);
    for (hipEvent_t e : {p->ev_in[0], p->ev_in[1], p->ev_run[0], p->ev_run[1], p->ev_out[0], p->ev_out[1], p->last_done})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {p->s_run, p->s_out})
        if (st) (void)hipStreamDestroy(st);
    if (have_prev) (void)hipSetDevice(prev);
    delete p;
}

template <typename T> static int launch_premul(Plan* p, double mu, double t0, double dt, int demod, hipStream_t s)
{
    const int n = (int)p->n_in;
    hipLaunchKernelGGL(premul_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<cx<T>*>(p->d_pre),
                       static_cast<const T*>(p->d_win), static_cast<const cx<T>*>(p->d_chirp), n, mu, t0, dt,
                       demod);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

template <typename T> static int build_chirpz(Plan* p)
{
    // chirp c[j] = exp(-i pi j^2 / n) and the wrapped filter, both generated on the device
    const int64_t n = p->n, m = p->m;
    if (n >= (int64_t(1) << 31)) { set_error("chirp-z lengths are limited to 2^31 - 1"); return MFFT_E_UNSUPPORTED; }
    MFFT_HIP(hipMalloc(&p->d_chirp, (size_t)n * sizeof(cx<T>)));
    {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(chirp_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, nullptr,
                           static_cast<cx<T>*>(p->d_chirp), n);
        MFFT_HIP(hipGetLastError());
    }
    void* d_b = nullptr;
    MFFT_HIP(hipMalloc(&d_b, (size_t)m * sizeof(cx<T>)));
    {
        int64_t blocks = (m + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(chirp_filter_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, nullptr,
                           static_cast<const cx<T>*>(p->d_chirp), static_cast<cx<T>*>(d_b), n, p->n_in, m);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(e, "chirp_filter_kernel", __FILE__, __LINE__); }
    }
    int rc;
    hipError_t e = hipMalloc(&p->d_filt, (size_t)m * sizeof(cx<T>));
    if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    // one-signal engine for the filter transform; conj_out stores conj(FFT(b)), the form the
    // first stage multiplies by
    Pow2Engine one;
    rc = one.build(m, 1, p->dtype);
    if (rc == MFFT_OK) {
        Io<T> io;
        io.in = d_b; io.out = p->d_filt; io.in_stride = m; io.out_stride = m;
        io.n_in = m; io.n_out = m; io.conj_out = 1;
        rc = one.exec<T>(io, nullptr);
        if (rc == MFFT_OK) {
            e = hipDeviceSynchronize();
            if (e != hipSuccess) rc = hip_fail(e, "hipDeviceSynchronize", __FILE__, __LINE__);
        }
    }
    one.release();
    (void)hipFree(d_b);
    if (rc) return rc;
    MFFT_HIP(hipMalloc(&p->d_conv, (size_t)m * (size_t)p->batch * sizeof(cx<T>)));
    MFFT_HIP(hipMalloc(&p->d_pre, (size_t)p->n_in * sizeof(cx<T>)));
    p->workspace += (int64_t)((size_t)n + 2 * (size_t)m + (size_t)m * (size_t)p->batch + (size_t)p->n_in) * (int64_t)sizeof(cx<T>);
    if (!(p->flags & MFFT_DEMOD)) {
        rc = launch_premul<T>(p, 0.0, 0.0, 0.0, 0, nullptr);   // static premultiplier: window * chirp
        if (rc) return rc;
        MFFT_HIP(hipDeviceSynchronize());
    }
    return MFFT_OK;
}

static int plan_build_parts(Plan* p);

// Does an execution of this plan touch device state that belongs to the plan rather than to the call?
// Pass workspaces (several global passes), the chirp-z convolution buffer, the per-call demodulation
// table, and the dense staging copies around a general single-kernel transform of a segmented layout.
static bool plan_shares_state(const Plan* p)
{
    if (p->kind == Plan::CHIRPZ || (p->flags & MFFT_DEMOD)) return true;
    if (p->kind != Plan::POW2 && p->kind != Plan::COLUMNS) return false;
    if (!p->eng.single) return true;
    if (p->lin.seg == 0 && p->lout.seg == 0) return false;
    // segmented: only the plain transform over the segments seg_is_fast() names runs without staging
    // (V_SEG in Pow2Engine::exec)
    return !(p->flags == 0 && p->n_in == p->n && seg_is_fast(p->lin.seg, p->n, p->eng.geo.p, p->dtype) &&
             seg_is_fast(p->lout.seg, p->n, p->eng.geo.p, p->dtype));
}

static int plan_build(Plan* p)
{
    const int rc = plan_build_parts(p);
    if (rc == MFFT_OK) p->shares_state = plan_shares_state(p);
    return rc;
}

static int plan_build_parts(Plan* p)
{
    int rc;
    if (p->flags & MFFT_HANN) {
        std::vector<long double> w;
        hann_window(p->n_in, &w);
        rc = p->dtype == MFFT_C64 ? upload_real<float>(w, &p->d_win) : upload_real<double>(w, &p->d_win);
        if (rc) return rc;
        p->workspace += p->n_in * (int64_t)real_size(p->dtype);
    }
    if (p->kind == Plan::COLUMNS) {
        rc = p->eng.build(p->n, p->batch, p->dtype, p->inner, p->fs_n, p->fs_c0);
        if (rc) return rc;
        p->workspace += p->eng.bytes;
        p->launches = p->eng.launches();
        return MFFT_OK;
    }
    if (p->n == 1) {
        p->kind = Plan::ONE_POINT;
        if (p->flags & MFFT_DEMOD) MFFT_HIP(hipMalloc(&p->d_pre, 2 * real_size(p->dtype)));
        p->launches = 1 + ((p->flags & MFFT_DEMOD) ? 1 : 0);
        return MFFT_OK;
    }
    if (is_pow2(p->n)) {
        p->kind = Plan::POW2;
        // Real input whose HALF-length transform still takes several passes: packed form.  z[j] = x[2j] + i x[2j+1]
        // is the input buffer itself read as complex, every pass moves half the bytes of the complex route, and the
        // last pass splits the n/2 bins into the n of the real signal's spectrum as it stores them
        // (tile_r2c_kernel.h).  (Shorter real signals are read as reals by the single-kernel variants: one pass
        // either way.)  Measured against the promoting route (alternated in one process, 2^24 points per launch):
        // complex128 0.56-0.75 of its time from 2^15 points on, complex64 0.62-0.87 from 2^16 on -- with a separate
        // split kernel it was 0.78-0.89 and paid only from 2^20 complex64 points.
        int from = p->dtype == MFFT_C64 ? 16 : 15;
        if (const char* e = getenv("MFFT_R2C_FROM")) from = atoi(e);        // tuning aid
        if ((p->flags & MFFT_REAL_IN) && !(p->flags & (MFFT_DEMOD | MFFT_INVERSE)) && ilog2(p->n) >= from && (p->n_in & 1) == 0 &&
            !p->lin.batch_stride && !p->lin.seg && !p->lout.batch_stride && !p->lout.seg && !getenv("MFFT_NO_R2C")) {
            p->packed_real = true;
            rc = p->eng.build(p->n / 2, p->batch, p->dtype);
            if (rc) return rc;
            // The split is fused into the last pass wherever that pass has two tiles or more per signal (every length
            // this route takes); MFFT_R2C_SPLIT_KERNEL=1 keeps the separate kernel and its n/2-element intermediate (A/B)
            size_t half_bytes = 0;
            if (!p->eng.r2c_fusable() || getenv("MFFT_R2C_SPLIT_KERNEL")) {
                half_bytes = (size_t)p->batch * (size_t)(p->n / 2) * 2 * real_size(p->dtype);
                MFFT_HIP(hipMalloc(&p->d_half, half_bytes));
            }
            p->workspace += p->eng.bytes + (int64_t)half_bytes;
            p->launches = p->eng.launches() + 1 + ((p->flags & MFFT_STATS) ? 2 : 0);
            return MFFT_OK;
        }
        rc = p->eng.build(p->n, p->batch, p->dtype);
        if (rc) return rc;
        if (p->flags & MFFT_DEMOD) {
            MFFT_HIP(hipMalloc(&p->d_pre, (size_t)p->n_in * 2 * real_size(p->dtype)));
            p->workspace += p->n_in * 2 * (int64_t)real_size(p->dtype);
        }
        p->workspace += p->eng.bytes;
        p->launches = p->eng.launches() + ((p->flags & MFFT_DEMOD) ? 1 : 0);
        return MFFT_OK;
    }
    // arbitrary length: chirp-z on the next power of two that holds the linear convolution
    p->kind = Plan::CHIRPZ;
    p->m = 1;
    while (p->m < p->n + p->n_in - 1) p->m <<= 1;
    rc = p->eng.build(p->m, p->batch, p->dtype);
    if (rc) return rc;
    rc = p->dtype == MFFT_C64 ? build_chirpz<float>(p) : build_chirpz<double>(p);
    if (rc) return rc;
    p->workspace += p->eng.bytes;
    p->launches = 2 * p->eng.launches() + ((p->flags & MFFT_DEMOD) ? 1 : 0);
    return MFFT_OK;
}

template <typename T>
static int exec_typed(Plan* p, const void* d_in, void* d_out, double scale, double mu, double t0, double dt,
                      double* d_stats, hipStream_t s, int64_t slab = 0, bool tables = true)
{
    // slab > 0: d_in / d_out / d_stats point at a run of `slab` signals of the batch (the pipelined
    // host path); `tables` = rebuild the per-call premultiplier (first slab only)
    const int64_t nb = slab > 0 ? slab : p->batch;
    const bool demod = (p->flags & MFFT_DEMOD) != 0;
    const bool power = (p->flags & MFFT_POWER) != 0;
    const bool inverse = (p->flags & MFFT_INVERSE) != 0;
    double* stats = (p->flags & MFFT_STATS) ? d_stats : nullptr;
    // kernels that accumulate the statistics with atomics need them cleared; the multi-pass engine
    // (chirp-z and long transforms) writes all three values from a reduction of its own
    if (stats && !p->packed_real && !p->eng.stats_by_reduction(power, p->lout.seg))
        MFFT_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 3 * (size_t)nb, s));
    if (demod && tables) {
        const int rc = launch_premul<T>(p, mu, t0, dt, 1, s);
        if (rc) return rc;
    }
    const T* win = static_cast<const T*>(p->d_win);
    const cx<T>* pre = static_cast<const cx<T>*>(p->d_pre);

    if (p->kind == Plan::ONE_POINT) {
        const int64_t blocks = (nb + 255) / 256;
        if (p->lout.batch_stride > 1) { set_error("1-point plans take dense output only"); return MFFT_E_UNSUPPORTED; }
        hipLaunchKernelGGL(n1_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, d_in, d_out, nb,
                           p->lin.batch_stride ? p->lin.batch_stride : p->n_in,
                           (p->flags & MFFT_REAL_IN) ? 1 : 0, demod ? (const T*)nullptr : win,
                           demod ? pre : (const cx<T>*)nullptr, (T)scale, inverse ? 1 : 0, power ? 1 : 0, stats);
        MFFT_HIP(hipGetLastError());
        return MFFT_OK;
    }
    if (p->kind == Plan::COLUMNS) {
        if (scale != 1.0) { set_error("column-batched plans do not scale their output"); return MFFT_E_INVALID; }
        Io<T> io;
        io.in = d_in; io.out = d_out;
        io.in_stride = p->n * p->inner;
        io.out_stride = p->n * (p->eng.out_ld > 0 ? p->eng.out_ld : p->inner);
        io.n_in = io.n_out = p->n;
        io.batch = slab;
        return p->eng.exec<T>(io, s);
    }
    if (p->kind == Plan::POW2 && p->packed_real) {
        const int64_t M = p->n / 2;
        Io<T> io;                                       // the input read as n_in/2 complex samples
        io.in = d_in; io.out = p->d_half;
        io.in_stride = p->n_in / 2; io.out_stride = M;
        io.win = win; io.win_pairs = win ? 1 : 0;       // the n_in real weights, taken two at a time
        io.n_in = p->n_in / 2; io.n_out = M;
        io.batch = slab;
        if (!p->d_half) {
            // the last pass forms the n bins itself (no M-element intermediate, no split kernel)
            io.out = d_out; io.out_stride = p->n; io.n_out = p->n;
            io.r2c = true; io.power = power; io.scale = (T)scale;
            const int rc = p->eng.exec<T>(io, s);
            if (rc) return rc;
            if (stats) return p->eng.reduce_stats<T>(static_cast<const T*>(d_out), p->n, p->n, nb, stats, s);
            return MFFT_OK;
        }
        int rc = p->eng.exec<T>(io, s);
        if (rc) return rc;
        int64_t blocks = (M + 1 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (nb > 65535) { set_error("at most 65535 real signals per call above one LDS residency"); return MFFT_E_UNSUPPORTED; }
        const dim3 grid((unsigned)blocks, (unsigned)nb);
        if (power)
            hipLaunchKernelGGL((r2c_split_kernel<T, true>), grid, dim3(256), 0, s, static_cast<const cx<T>*>(p->d_half), d_out, M,
                               ilog2(p->n), M, p->n, (T)scale);
        else
            hipLaunchKernelGGL((r2c_split_kernel<T, false>), grid, dim3(256), 0, s, static_cast<const cx<T>*>(p->d_half), d_out, M,
                               ilog2(p->n), M, p->n, (T)scale);
        MFFT_HIP(hipGetLastError());
        if (stats) return p->eng.reduce_stats<T>(static_cast<const T*>(d_out), p->n, p->n, nb, stats, s);
        return MFFT_OK;
    }
    if (p->kind == Plan::POW2) {
        Io<T> io;
        io.in = d_in; io.out = d_out;
        io.in_stride = p->lin.batch_stride ? p->lin.batch_stride : p->n_in;
        io.out_stride = p->lout.batch_stride ? p->lout.batch_stride : p->n;
        io.in_seg = p->lin.seg; io.in_seg_stride = p->lin.seg_stride;
        io.out_seg = p->lout.seg; io.out_seg_stride = p->lout.seg_stride;
        io.win = demod ? nullptr : win;          // with DEMOD the window is folded into pre
        io.pre = demod ? pre : nullptr;
        io.stats = stats;
        io.n_in = p->n_in; io.n_out = p->n;
        io.real_in = (p->flags & MFFT_REAL_IN) ? 1 : 0;
        io.conj_in = inverse; io.conj_out = inverse;
        io.scale = (T)scale;
        io.power = power;
        io.batch = slab;
        return p->eng.exec<T>(io, s);
    }
    // chirp-z: X[k] = c[k] * sum_j (x[j] c[j]) conj(c)[k - j]
    if (inverse) { set_error("MFFT_INVERSE is implemented for power-of-two lengths only"); return MFFT_E_UNSUPPORTED; }
    Io<T> a;                                        // stage 1: conj(FFT_m(x * pre) ) * conj(FFT_m(b))
    a.in = d_in; a.out = p->d_conv;
    a.in_stride = p->lin.batch_stride ? p->lin.batch_stride : p->n_in;
    a.in_seg = p->lin.seg; a.in_seg_stride = p->lin.seg_stride;
    a.out_stride = p->m;
    a.pre = pre;                                    // window * demod * chirp
    a.post = static_cast<const cx<T>*>(p->d_filt);
    a.n_in = p->n_in; a.n_out = p->m;
    a.real_in = (p->flags & MFFT_REAL_IN) ? 1 : 0;
    a.conj_out = 1;
    a.batch = slab;
    int rc = p->eng.exec<T>(a, s);
    if (rc) return rc;
    Io<T> b;                                        // stage 2: conj(FFT_m(.)) / m * c[k], first n bins
    b.in = p->d_conv; b.out = d_out;
    b.in_stride = p->m;
    b.out_stride = p->lout.batch_stride ? p->lout.batch_stride : p->n;
    b.out_seg = p->lout.seg; b.out_seg_stride = p->lout.seg_stride;
    b.post = static_cast<const cx<T>*>(p->d_chirp);
    b.stats = stats;
    b.n_in = p->m; b.n_out = p->n;
    b.conj_out = 1;
    b.scale = (T)(scale / (double)p->m);
    b.power = power;
    b.batch = slab;
    return p->eng.exec<T>(b, s);
}

static int plan_exec(Plan* p, const void* d_in, void* d_out, double scale, double mu, double t0, double dt,
                     double* d_stats, hipStream_t s, int64_t slab = 0, bool tables = true)
{
    return p->dtype == MFFT_C64 ? exec_typed<float>(p, d_in, d_out, scale, mu, t0, dt, d_stats, s, slab, tables)
                                : exec_typed<double>(p, d_in, d_out, scale, mu, t0, dt, d_stats, s, slab, tables);
}

// Every execution goes through here.  A plan that owns per-execution device state is executed by one
// thread at a time, and an execution on another stream than the previous one first waits, on the
// device, for that one's completion event -- whichever entry point issued it (mfft_exec on the
// caller's stream, the host entry points on the null stream or on the plan's own streams).
static int plan_exec_ordered(Plan* p, const void* d_in, void* d_out, double scale, double mu, double t0, double dt,
                             double* d_stats, hipStream_t s, int64_t slab = 0, bool tables = true)
{
    if (!p->shares_state) return plan_exec(p, d_in, d_out, scale, mu, t0, dt, d_stats, s, slab, tables);
    std::lock_guard<std::mutex> order(p->order_mu);
    if (stream_capturing(s)) {
        // recorded into the caller's graph: the graph orders it; the caller orders launches of that graph against
        // other executions of this plan (events from outside a capture cannot be waited for inside it)
        return plan_exec(p, d_in, d_out, scale, mu, t0, dt, d_stats, s, slab, tables);
    }
    if (!p->last_done) MFFT_HIP(hipEventCreateWithFlags(&p->last_done, hipEventDisableTiming));
    if (p->have_last && p->last_stream != s) MFFT_HIP(hipStreamWaitEvent(s, p->last_done, 0));
    const int rc = plan_exec(p, d_in, d_out, scale, mu, t0, dt, d_stats, s, slab, tables);
    // also after a failure part-way: kernels of the passes that were launched may still be running
    MFFT_HIP(hipEventRecord(p->last_done, s));
    p->last_stream = s;
    p->have_last = true;
    return rc;
}

// ---------------------------------------------------------------------------
// Host side of mfft_exec_host for large batches.  One thread moves pageable memory at 3-5 GB/s
// (page faults on fresh numpy arrays included) while the bus takes 57 GB/s from pinned memory, so
// the batch is cut into slabs that a few threads copy through two pinned buffers per direction
// while the previous slab is on the bus and in the kernels.
// ---------------------------------------------------------------------------
static int pipe_prepare(Plan* p, size_t in_row, size_t out_row)
{
    if (p->pipe_slab > 0) return MFFT_OK;
    const size_t target = size_t(16) << 20;                    // bytes per slab and direction
    const size_t row = in_row > out_row ? in_row : out_row;
    int64_t slab = (int64_t)(target / row);
    if (slab < 1) slab = 1;
    // handles that already exist (an earlier attempt that failed part-way) are kept, not overwritten
    hipError_t e = hipSuccess;
    for (int q = 0; q < 2 && e == hipSuccess; ++q) {
        if (!p->pin_in[q]) e = hipHostMalloc(&p->pin_in[q], (size_t)slab * in_row, hipHostMallocDefault);
        if (e == hipSuccess && !p->pin_out[q]) e = hipHostMalloc(&p->pin_out[q], (size_t)slab * out_row, hipHostMallocDefault);
        if (e == hipSuccess && !p->ev_in[q]) e = hipEventCreateWithFlags(&p->ev_in[q], hipEventDisableTiming);
        if (e == hipSuccess && !p->ev_run[q]) e = hipEventCreateWithFlags(&p->ev_run[q], hipEventDisableTiming);
        if (e == hipSuccess && !p->ev_out[q]) e = hipEventCreateWithFlags(&p->ev_out[q], hipEventDisableTiming);
    }
    if (e == hipSuccess && !p->s_run) e = hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking);
    if (e == hipSuccess && !p->s_out) e = hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_fail(e, "pipelined staging", __FILE__, __LINE__);
    p->pipe_slab = slab;
    return MFFT_OK;
}

// h_in -> (threads) pinned -> H2D -> kernels -> D2H -> pinned -> (threads) h_out.  Two slabs are in
// flight per direction; the download copies run on their own threads, concurrently with the upload
// copy of the following slab (a fresh output array costs page faults on top of the copy: measured
// 15 ms per 512 MiB against 8 ms for the upload side, one after the other before).
static int exec_host_pipelined(Plan* p, const char* h_in, char* h_out, size_t in_row, size_t out_row, double scale,
                               double mu, double t0, double dt)
{
    int rc = pipe_prepare(p, in_row, out_row);
    if (rc) return rc;
    CopyPool& pin = CopyPool::in();
    CopyPool& pout = CopyPool::out();
    const int64_t slab = p->pipe_slab, batch = p->batch;
    const int64_t nslab = (batch + slab - 1) / slab;
    char* d_in = static_cast<char*>(p->d_in);
    char* d_out = static_cast<char*>(p->d_out);
    uint64_t out_ticket[2] = {0, 0};                   // download copy still reading pin_out[q]
    const auto drain = [&] { pout.wait(out_ticket[0]); pout.wait(out_ticket[1]); };
    // no return while a download copy still writes into the caller's array
#define PIPE_HIP(expr)                                                              \
    do {                                                                            \
        const hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { drain(); return hip_fail(e_, #expr, __FILE__, __LINE__); } \
    } while (0)
    for (int64_t k = 0; k < nslab; ++k) {
        const int q = (int)(k & 1);
        const int64_t b0 = k * slab, cnt = batch - b0 < slab ? batch - b0 : slab;
        if (k >= 2) PIPE_HIP(hipEventSynchronize(p->ev_in[q]));             // pin_in[q] has left for the device
        pin.copy(p->pin_in[q], h_in + (size_t)b0 * in_row, (size_t)cnt * in_row);
        PIPE_HIP(hipMemcpyAsync(d_in + (size_t)b0 * in_row, p->pin_in[q], (size_t)cnt * in_row, hipMemcpyHostToDevice, p->s_run));
        PIPE_HIP(hipEventRecord(p->ev_in[q], p->s_run));
        rc = plan_exec_ordered(p, d_in + (size_t)b0 * in_row, d_out + (size_t)b0 * out_row, scale, mu, t0, dt,
                               p->d_stats ? p->d_stats + 3 * b0 : nullptr, p->s_run, cnt, k == 0);
        if (rc) { (void)hipStreamSynchronize(p->s_run); (void)hipStreamSynchronize(p->s_out); drain(); return rc; }
        if (h_out) {
            PIPE_HIP(hipEventRecord(p->ev_run[q], p->s_run));
            PIPE_HIP(hipStreamWaitEvent(p->s_out, p->ev_run[q], 0));
            pout.wait(out_ticket[q]);                                        // slab k-2 has left pin_out[q]
            PIPE_HIP(hipMemcpyAsync(p->pin_out[q], d_out + (size_t)b0 * out_row, (size_t)cnt * out_row, hipMemcpyDeviceToHost, p->s_out));
            PIPE_HIP(hipEventRecord(p->ev_out[q], p->s_out));
            if (k >= 1) {                                                    // previous slab: start its download copy
                PIPE_HIP(hipEventSynchronize(p->ev_out[q ^ 1]));
                out_ticket[q ^ 1] = pout.start(h_out + (size_t)(b0 - slab) * out_row, p->pin_out[q ^ 1], (size_t)slab * out_row);
            }
        }
    }
    if (h_out) {
        const int q = (int)((nslab - 1) & 1);
        const int64_t b0 = (nslab - 1) * slab;
        PIPE_HIP(hipEventSynchronize(p->ev_out[q]));
        out_ticket[q] = pout.start(h_out + (size_t)b0 * out_row, p->pin_out[q], (size_t)(batch - b0) * out_row);
        drain();
    }
    PIPE_HIP(hipStreamSynchronize(p->s_run));
    return MFFT_OK;
#undef PIPE_HIP
}

// RAII: make the plan's device current for the duration of a call
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace mfft

// ===========================================================================
// C ABI
// ===========================================================================
using namespace mfft;

// Device / pinned blocks that helper entry points keep between calls (released by mfft_release_cached)
struct LsqScratch { double* dev = nullptr; double* pin = nullptr; size_t cap = 0; };
static std::mutex g_lsq_mu;
static LsqScratch g_lsq_pool[64];                       // mfft_lsq_normal_host: one per device
struct TwiddleTable { int device = -1; int64_t n = 0; cx<double>* d = nullptr; int lo_bits = 0; uint64_t used = 0; };
static std::mutex g_tab_mu;
static TwiddleTable g_tabs[8];                          // mfft_twiddle_tile: most recent moduli
static uint64_t g_tab_tick = 0;

extern "C" {

int mfft_version(void) { return 100; }

int mfft_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return MFFT_E_NODEVICE;
    }
    return n;
}

int mfft_release_cached(void)
{
    int prev = 0;
    const bool have = hipGetDevice(&prev) == hipSuccess;
    {
        std::lock_guard<std::mutex> lock(g_lsq_mu);
        for (int dev = 0; dev < 64; ++dev) {
            LsqScratch& sc = g_lsq_pool[dev];
            if (!sc.dev && !sc.pin) continue;
            (void)hipSetDevice(dev);
            if (sc.dev) (void)hipFree(sc.dev);
            if (sc.pin) (void)hipHostFree(sc.pin);
            sc = LsqScratch{};
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_tab_mu);
        for (TwiddleTable& t : g_tabs) {
            if (!t.d) continue;
            (void)hipSetDevice(t.device);
            (void)hipFree(t.d);
            t = TwiddleTable{};
        }
    }
    if (have) (void)hipSetDevice(prev);
    return MFFT_OK;
}

int mfft_device_memory(int64_t* free_bytes, int64_t* total_bytes)
{
    if (!free_bytes || !total_bytes) { set_error("null argument"); return MFFT_E_INVALID; }
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    size_t f = 0, t = 0;
    MFFT_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return MFFT_OK;
}

const char* mfft_strerror(int code)
{
    switch (code) {
    case MFFT_OK: return "success";
    case MFFT_E_INVALID: return "invalid argument";
    case MFFT_E_UNSUPPORTED: return "unsupported request";
    case MFFT_E_NOMEM: return "out of memory";
    case MFFT_E_HIP: return "HIP runtime error";
    case MFFT_E_NODEVICE: return "no usable device";
    }
    return "unknown error";
}

const char* mfft_last_error(void) { return g_err.c_str(); }

static int layout_ok(const mfft_layout* l, int64_t n)
{
    if (!l) return 1;
    if (l->batch_stride < 0 || l->seg < 0 || l->seg_stride < 0) return 0;
    if (l->seg > 0 && (l->seg_stride < l->seg || l->seg > n)) return 0;
    return 1;
}

int mfft_plan_1d_layout(mfft_plan* out, int64_t n, int64_t n_in, int64_t batch, int dtype, uint32_t flags,
                        const mfft_layout* in_layout, const mfft_layout* out_layout)
{
    if (!out) { set_error("null plan pointer"); return MFFT_E_INVALID; }
    *out = nullptr;
    if (n < 1) { set_error("Invalid number of FFT data points (" + std::to_string(n) + ") specified."); return MFFT_E_INVALID; }
    if (n_in < 1 || n_in > n) { set_error("n_in must satisfy 1 <= n_in <= n"); return MFFT_E_INVALID; }
    if (batch < 1) { set_error("batch must be >= 1"); return MFFT_E_INVALID; }
    if (dtype != MFFT_C64 && dtype != MFFT_C128) { set_error("dtype must be MFFT_C64 or MFFT_C128"); return MFFT_E_INVALID; }
    const uint32_t known = MFFT_HANN | MFFT_POWER | MFFT_DEMOD | MFFT_REAL_IN | MFFT_STATS | MFFT_INVERSE;
    if (flags & ~known) { set_error("unknown flag bits"); return MFFT_E_INVALID; }
    if ((flags & MFFT_STATS) && !(flags & MFFT_POWER)) { set_error("MFFT_STATS needs MFFT_POWER"); return MFFT_E_INVALID; }
    if (!layout_ok(in_layout, n_in) || !layout_ok(out_layout, n)) { set_error("bad layout descriptor"); return MFFT_E_INVALID; }
    if (n == 1 && out_layout && out_layout->batch_stride > 1) { set_error("1-point plans take dense output only"); return MFFT_E_UNSUPPORTED; }
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    Plan* p = new (std::nothrow) Plan;
    if (!p) { set_error("host allocation failed"); return MFFT_E_NOMEM; }
    p->n = n; p->n_in = n_in; p->batch = batch; p->dtype = dtype; p->flags = flags;
    if (in_layout) p->lin = *in_layout;
    if (out_layout) p->lout = *out_layout;
    // a segment at least as long as the signal is no segmentation: keep the fast kernels
    if (p->lin.seg >= n_in) p->lin.seg = p->lin.seg_stride = 0;
    if (p->lout.seg >= n) p->lout.seg = p->lout.seg_stride = 0;
    if (p->lin.batch_stride == n_in && p->lin.seg == 0) p->lin.batch_stride = 0;
    if (p->lout.batch_stride == n && p->lout.seg == 0) p->lout.batch_stride = 0;
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; set_error("hipGetDevice failed"); return MFFT_E_HIP; }
    const int rc = plan_build(p);
    if (rc != MFFT_OK) { plan_free(p); return rc; }
    *out = reinterpret_cast<mfft_plan>(p);
    return MFFT_OK;
}

int mfft_plan_1d(mfft_plan* out, int64_t n, int64_t n_in, int64_t batch, int dtype, uint32_t flags)
{
    return mfft_plan_1d_layout(out, n, n_in, batch, dtype, flags, nullptr, nullptr);
}

int mfft_plan_1d_columns(mfft_plan* out, int64_t n, int64_t inner, int64_t batch, int dtype,
                         int64_t twiddle_n, int64_t twiddle_c0)
{
    return mfft_plan_1d_columns_ld(out, n, inner, batch, dtype, twiddle_n, twiddle_c0, 0);
}

int mfft_plan_1d_columns_ld(mfft_plan* out, int64_t n, int64_t inner, int64_t batch, int dtype,
                            int64_t twiddle_n, int64_t twiddle_c0, int64_t out_ld)
{
    if (!out) { set_error("null plan pointer"); return MFFT_E_INVALID; }
    *out = nullptr;
    if (n < 1) { set_error("Invalid number of FFT data points (" + std::to_string(n) + ") specified."); return MFFT_E_INVALID; }
    if (inner < 1 || batch < 1) { set_error("inner and batch must be >= 1"); return MFFT_E_INVALID; }
    if (dtype != MFFT_C64 && dtype != MFFT_C128) { set_error("dtype must be MFFT_C64 or MFFT_C128"); return MFFT_E_INVALID; }
    if (twiddle_n < 0 || twiddle_c0 < 0) { set_error("bad twiddle description"); return MFFT_E_INVALID; }
    if (!is_pow2(n) || n < (int64_t(1) << kTileMinLog2G)) {
        set_error("column-batched plans serve power-of-two lengths >= 16");
        return MFFT_E_UNSUPPORTED;
    }
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    Plan* p = new (std::nothrow) Plan;
    if (!p) { set_error("host allocation failed"); return MFFT_E_NOMEM; }
    p->kind = Plan::COLUMNS;
    p->n = n; p->n_in = n; p->batch = batch; p->dtype = dtype; p->flags = 0;
    p->inner = inner; p->fs_n = twiddle_n; p->fs_c0 = twiddle_c0;
    if (out_ld == inner) out_ld = 0;
    if (out_ld != 0) {
        // rows of the result out_ld elements apart: the plain last pass only, 32-bit row offsets inside a tile
        if (out_ld < inner || twiddle_n > 1 || n > (int64_t(1) << 32) / out_ld) {
            delete p;
            set_error("out_ld must be >= inner with n * out_ld <= 2^32, and excludes the fused twiddle");
            return out_ld < inner ? MFFT_E_INVALID : MFFT_E_UNSUPPORTED;
        }
        p->eng.out_ld = out_ld;
    }
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; set_error("hipGetDevice failed"); return MFFT_E_HIP; }
    const int rc = plan_build(p);
    if (rc != MFFT_OK) { plan_free(p); return rc; }
    *out = reinterpret_cast<mfft_plan>(p);
    return MFFT_OK;
}

int mfft_plan_destroy(mfft_plan plan)
{
    if (!plan) return MFFT_OK;
    plan_free(reinterpret_cast<Plan*>(plan));
    return MFFT_OK;
}

int mfft_plan_workspace_bytes(mfft_plan plan, int64_t* bytes)
{
    if (!plan || !bytes) { set_error("null argument"); return MFFT_E_INVALID; }
    *bytes = reinterpret_cast<Plan*>(plan)->workspace;
    return MFFT_OK;
}

int mfft_plan_launches(mfft_plan plan, int* launches)
{
    if (!plan || !launches) { set_error("null argument"); return MFFT_E_INVALID; }
    *launches = reinterpret_cast<Plan*>(plan)->launches;
    return MFFT_OK;
}

int mfft_exec(mfft_plan plan, const void* d_in, void* d_out, double scale, double mu, double t0, double dt,
              double* d_stats, void* stream)
{
    if (!plan || !d_in || !d_out) { set_error("null argument"); return MFFT_E_INVALID; }
    Plan* p = reinterpret_cast<Plan*>(plan);
    if ((p->flags & MFFT_STATS) && !d_stats) { set_error("plan has MFFT_STATS but d_stats is null"); return MFFT_E_INVALID; }
    DeviceGuard g(p->device);
    if (!g.ok) { set_error("cannot select the plan's device"); return MFFT_E_HIP; }
    return plan_exec_ordered(p, d_in, d_out, scale, mu, t0, dt, d_stats, static_cast<hipStream_t>(stream));
}

int mfft_exec_host(mfft_plan plan, const void* h_in, void* h_out, double scale, double mu, double t0,
                   double dt, double* h_stats)
{
    if (!plan || !h_in) { set_error("null argument"); return MFFT_E_INVALID; }
    Plan* p = reinterpret_cast<Plan*>(plan);
    if ((p->flags & MFFT_STATS) && !h_stats) { set_error("plan has MFFT_STATS but h_stats is null"); return MFFT_E_INVALID; }
    if (!h_out && !(p->flags & MFFT_STATS)) { set_error("h_out may be null only when the plan returns stats"); return MFFT_E_INVALID; }
    if (p->lin.batch_stride || p->lin.seg || p->lout.batch_stride || p->lout.seg || p->eng.out_ld > 0) {
        set_error("mfft_exec_host serves dense layouts only; use mfft_exec for strided / segmented plans");
        return MFFT_E_UNSUPPORTED;
    }
    DeviceGuard g(p->device);
    if (!g.ok) { set_error("cannot select the plan's device"); return MFFT_E_HIP; }
    std::lock_guard<std::mutex> lock(p->mu);
    const size_t rs = real_size(p->dtype);
    const size_t in_bytes = (size_t)p->batch * (size_t)p->n_in * (size_t)p->inner * rs * ((p->flags & MFFT_REAL_IN) ? 1 : 2);
    const size_t out_bytes = (size_t)p->batch * (size_t)p->n * (size_t)p->inner * rs * ((p->flags & MFFT_POWER) ? 1 : 2);
    if (!p->d_in) MFFT_HIP(hipMalloc(&p->d_in, in_bytes));
    if (!p->d_out) MFFT_HIP(hipMalloc(&p->d_out, out_bytes));
    if ((p->flags & MFFT_STATS) && !p->d_stats)
        MFFT_HIP(hipMalloc((void**)&p->d_stats, sizeof(double) * 3 * (size_t)p->batch));
    const size_t in_row = in_bytes / (size_t)p->batch, out_row = out_bytes / (size_t)p->batch;
    const bool pipelined = p->kind != Plan::COLUMNS && p->batch >= 4 && in_bytes + (h_out ? out_bytes : 0) >= (size_t(48) << 20) &&
                           in_row <= (size_t(64) << 20) && !getenv("MFFT_NO_PIPELINE");
    if (pipelined) {
        const int rc = exec_host_pipelined(p, static_cast<const char*>(h_in), static_cast<char*>(h_out), in_row, out_row,
                                           scale, mu, t0, dt);
        if (rc != MFFT_OK) return rc;
    } else {
        MFFT_HIP(hipMemcpy(p->d_in, h_in, in_bytes, hipMemcpyHostToDevice));
        const int rc = plan_exec_ordered(p, p->d_in, p->d_out, scale, mu, t0, dt, p->d_stats, nullptr);
        if (rc != MFFT_OK) return rc;
        if (h_out) MFFT_HIP(hipMemcpy(h_out, p->d_out, out_bytes, hipMemcpyDeviceToHost));
    }
    if (p->flags & MFFT_STATS)
        MFFT_HIP(hipMemcpy(h_stats, p->d_stats, sizeof(double) * 3 * (size_t)p->batch, hipMemcpyDeviceToHost));
    return MFFT_OK;
}

static int amp_args(AmpArgs* a, const double* w, const double* c0, const double* vrow, int nk, double hbar)
{
    if (!w || !c0 || !vrow || nk < 1 || nk > 32 || hbar == 0.0) { set_error("bad block description (1 <= nk <= 32)"); return MFFT_E_INVALID; }
    a->nk = nk;
    a->hbar = hbar;
    memcpy(a->w, w, sizeof(double) * (size_t)nk);
    memcpy(a->c0, c0, sizeof(double) * 2 * (size_t)nk);
    memcpy(a->vrow, vrow, sizeof(double) * 2 * (size_t)nk);
    return MFFT_OK;
}

int mfft_return_amplitude(void* d_out, int64_t n, double t0, double dt, const double* h_w, const double* h_c0,
                          const double* h_vrow, int nk, double hbar, void* stream)
{
    if (!d_out || n < 0) { set_error("bad argument"); return MFFT_E_INVALID; }
    AmpArgs a;
    const int rc = amp_args(&a, h_w, h_c0, h_vrow, nk, hbar);
    if (rc) return rc;
    if (n == 0) return MFFT_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(return_amplitude_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<cx<double>*>(d_out), n, (const double*)nullptr, t0, dt, a);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_return_amplitude_host(void* h_out, int64_t n, const double* h_times, double t0, double dt,
                               const double* h_w, const double* h_c0, const double* h_vrow, int nk, double hbar)
{
    if (!h_out || n < 0) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    AmpArgs a;
    int rc = amp_args(&a, h_w, h_c0, h_vrow, nk, hbar);
    if (rc) return rc;
    if (n == 0) return MFFT_OK;
    void* d_out = nullptr;
    double* d_t = nullptr;
    MFFT_HIP(hipMalloc(&d_out, (size_t)n * sizeof(cx<double>)));
    hipError_t e = hipSuccess;
    if (h_times) {
        e = hipMalloc((void**)&d_t, (size_t)n * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(d_t, h_times, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(return_amplitude_kernel, dim3((unsigned)blocks), dim3(256), 0, nullptr,
                           static_cast<cx<double>*>(d_out), n, (const double*)d_t, t0, dt, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, (size_t)n * sizeof(cx<double>), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (d_t) (void)hipFree(d_t);
    if (e != hipSuccess) return hip_fail(e, "return_amplitude", __FILE__, __LINE__);
    return MFFT_OK;
}

int mfft_lsq_normal_host(const void* h_x, const double* h_win, const double* h_times, int64_t n, double mu,
                         const double* h_E, int m, double* h_G, double* h_r, double* h_yy)
{
    if (!h_x || !h_times || !h_E || !h_G || !h_r || !h_yy || n < 1) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (m < 1 || m > 64) { set_error("1 <= m <= 64 columns"); return MFFT_E_INVALID; }
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    LsqArgs a;
    a.m = m;
    a.mu = mu;
    for (int i = 0; i < 64; ++i) a.E[i] = i < m ? h_E[i] : 0.0;
    // One device block and one pinned host block per device, kept and grown as needed (a call takes
    // ~50 us; concurrent callers queue on the lock):  x (2n) | times (n) | win (n) | results (2mm + 2m + 1)
    std::mutex& pool_mu = g_lsq_mu;
    LsqScratch (&pool)[64] = g_lsq_pool;
    int device = 0;
    MFFT_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64) { set_error("device index out of range"); return MFFT_E_UNSUPPORTED; }
    const size_t nres = 2 * (size_t)m * m + 2 * (size_t)m + 1;
    const size_t doubles = 4 * (size_t)n + nres;
    std::lock_guard<std::mutex> lock(pool_mu);
    LsqScratch& sc = pool[device];
    if (sc.cap < doubles) {
        if (sc.dev) (void)hipFree(sc.dev);
        if (sc.pin) (void)hipHostFree(sc.pin);
        sc = LsqScratch{};
        const size_t want = doubles + doubles / 2;
        MFFT_HIP(hipMalloc((void**)&sc.dev, want * sizeof(double)));
        MFFT_HIP(hipHostMalloc((void**)&sc.pin, want * sizeof(double), hipHostMallocDefault));
        sc.cap = want;
    }
    double* d_x = sc.dev;
    double* d_t = sc.dev + 2 * n;
    double* d_w = sc.dev + 3 * n;
    double* d_res = sc.dev + 4 * n;
    memcpy(sc.pin, h_x, 2 * (size_t)n * sizeof(double));
    memcpy(sc.pin + 2 * n, h_times, (size_t)n * sizeof(double));
    if (h_win) memcpy(sc.pin + 3 * n, h_win, (size_t)n * sizeof(double));
    hipError_t e = hipMemcpyAsync(sc.dev, sc.pin, (h_win ? 4 : 3) * (size_t)n * sizeof(double), hipMemcpyHostToDevice, nullptr);
    if (e == hipSuccess) e = hipMemsetAsync(d_res, 0, nres * sizeof(double), nullptr);
    if (e == hipSuccess) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 64) blocks = 64;
        for (int c0 = 0; c0 < m && e == hipSuccess; c0 += 16) {      // 16 columns of G per launch
            a.c0 = c0;
            a.nc = m - c0 < 16 ? m - c0 : 16;
            hipLaunchKernelGGL(lsq_normal_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, nullptr,
                               reinterpret_cast<const cx<double>*>(d_x), h_win ? d_w : (const double*)nullptr, d_t, n,
                               a, d_res);
            e = hipGetLastError();
        }
    }
    double* res = sc.pin + 4 * n;
    if (e == hipSuccess) e = hipMemcpyAsync(res, d_res, nres * sizeof(double), hipMemcpyDeviceToHost, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return hip_fail(e, "lsq_normal", __FILE__, __LINE__);
    memcpy(h_G, res, 2 * (size_t)m * m * sizeof(double));
    memcpy(h_r, res + 2 * (size_t)m * m, 2 * (size_t)m * sizeof(double));
    *h_yy = res[nres - 1];
    return MFFT_OK;
}

// address of `scratch` on the current device, resolved once per device
static int static_scratch(double** out)
{
    static std::mutex mu;
    static double* ptr[64] = {};
    int device = 0;
    MFFT_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 64) { set_error("device index out of range"); return MFFT_E_UNSUPPORTED; }
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[device]) {
        void* p = nullptr;
        MFFT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { set_error("hipGetSymbolAddress returned a null address"); return MFFT_E_HIP; }
        ptr[device] = static_cast<double*>(p);
    }
    *out = ptr[device];
    return MFFT_OK;
}

// The batched normal equations on device arrays: per-sample-block partial sums into `partial`, then
// the fixed-order final sums into d_out.  Allocates nothing; `partial` holds `partial_doubles` doubles.
static int lsq_normal_launch(const void* d_x, const double* d_win, const double* d_times, int64_t n, int64_t batch,
                             const double* d_mu, const double* d_E, int m, double* d_out, double* partial,
                             size_t partial_doubles, hipStream_t s, const char* who)
{
    const int nchunk = (m + kLsqCols - 1) / kLsqCols;
    // the grid's z extent first: it also bounds the products below (batch <= 65535, m <= 64)
    if (batch > 65535 / nchunk) {
        set_error(std::string(who) + ": at most " + std::to_string(65535 / nchunk) + " signals per call at this column count; split the batch");
        return MFFT_E_UNSUPPORTED;
    }
    // partial sums: batch * m * nchunk * nblk slots of kLsqSlot doubles
    const size_t per_blk = (size_t)batch * (size_t)m * (size_t)nchunk * kLsqSlot;
    if (per_blk > partial_doubles) {
        set_error(std::string(who) + ": batch * m * ceil(m/16) * 36 doubles exceed the " + std::to_string(partial_doubles * sizeof(double)) +
                  "-byte scratch (" + std::to_string(per_blk * sizeof(double)) + " bytes needed per sample block); split the batch");
        return MFFT_E_UNSUPPORTED;
    }
    int64_t nblk = (n + 255) / 256;
    if (nblk > 64) nblk = 64;
    if ((size_t)nblk > partial_doubles / per_blk) nblk = (int64_t)(partial_doubles / per_blk);
    LsqBatchArgs a;
    a.x = static_cast<const cx<double>*>(d_x);
    a.win = d_win; a.times = d_times; a.mu = d_mu; a.E = d_E;
    a.partial = partial;
    a.n = n; a.m = m; a.nchunk = nchunk; a.nblk = (int)nblk;
    // every partial slot that lsq_final_kernel reads is written by exactly one block of this launch,
    // so nothing in the scratch has to start at zero and no stale value of an earlier call is read
    hipLaunchKernelGGL(lsq_partial_kernel, dim3((unsigned)nblk, (unsigned)m, (unsigned)(batch * nchunk)), dim3(256), 0, s, a);
    MFFT_HIP(hipGetLastError());
    const int64_t total = (int64_t)per_blk;               // one wave per value
    int64_t fblocks = (total + 3) / 4;
    if (fblocks > 2048) fblocks = 2048;
    hipLaunchKernelGGL(lsq_final_kernel, dim3((unsigned)fblocks), dim3(256), 0, s, (const double*)partial, d_out,
                       (int)batch, m, nchunk, (int)nblk);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_lsq_normal_static(const void* d_x, const double* d_win, const double* d_times, int64_t n, int64_t batch,
                           const double* d_mu, const double* d_E, int m, double* d_out, void* stream)
{
    if (!d_x || !d_times || !d_E || !d_out || n < 1 || batch < 1) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (m < 1 || m > 64) { set_error("1 <= m <= 64 columns"); return MFFT_E_INVALID; }
    if (n > (int64_t(1) << 40)) { set_error("n above 2^40 samples"); return MFFT_E_INVALID; }
    double* d_scratch = nullptr;
    const int rc = static_scratch(&d_scratch);
    if (rc) return rc;
    return lsq_normal_launch(d_x, d_win, d_times, n, batch, d_mu, d_E, m, d_out, d_scratch, kScratchDoubles,
                             static_cast<hipStream_t>(stream), "mfft_lsq_normal_static");
}

int mfft_local_maxima(const void* d_S, int64_t n, int dtype, int64_t* d_idx, int64_t capacity,
                      int64_t* d_count, void* stream)
{
    if ((!d_S && n >= 3) || (!d_idx && capacity > 0) || !d_count || n < 0 || capacity < 0) {
        set_error("bad argument"); return MFFT_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    MFFT_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    if (n < 3) return MFFT_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == MFFT_C64)
        hipLaunchKernelGGL(local_maxima_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const float*>(d_S), n, d_idx, capacity,
                           reinterpret_cast<unsigned long long*>(d_count));
    else if (dtype == MFFT_C128)
        hipLaunchKernelGGL(local_maxima_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const double*>(d_S), n, d_idx, capacity,
                           reinterpret_cast<unsigned long long*>(d_count));
    else { set_error("bad dtype"); return MFFT_E_INVALID; }
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_exec_host_peaks(mfft_plan plan, const void* h_in, void* h_out, double scale, double mu, double t0,
                         double dt, double* h_stats, int64_t* h_idx, int64_t capacity, int64_t* h_count)
{
    if (!plan || !h_in || !h_out || !h_idx || !h_count || capacity < 1) { set_error("null argument"); return MFFT_E_INVALID; }
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!(p->flags & MFFT_POWER) || p->batch != 1 || p->kind == Plan::COLUMNS) { set_error("peak scan needs a one-signal MFFT_POWER plan"); return MFFT_E_INVALID; }
    if ((p->flags & MFFT_STATS) && !h_stats) { set_error("plan has MFFT_STATS but h_stats is null"); return MFFT_E_INVALID; }
    if (p->lin.batch_stride || p->lin.seg || p->lout.batch_stride || p->lout.seg) {
        set_error("host entry points serve dense layouts only");
        return MFFT_E_UNSUPPORTED;
    }
    DeviceGuard g(p->device);
    if (!g.ok) { set_error("cannot select the plan's device"); return MFFT_E_HIP; }
    std::lock_guard<std::mutex> lock(p->mu);
    const size_t rs = real_size(p->dtype);
    const size_t in_bytes = (size_t)p->n_in * rs * ((p->flags & MFFT_REAL_IN) ? 1 : 2);
    const size_t out_bytes = (size_t)p->n * rs;
    if (!p->d_in) MFFT_HIP(hipMalloc(&p->d_in, in_bytes));
    if (!p->d_out) MFFT_HIP(hipMalloc(&p->d_out, out_bytes));
    // one result block: {max, sum, centroid} | count | indices, mirrored in pinned host memory so
    // that the statistics, the count and the first indices come back in a single copy
    constexpr int64_t kHead = 4, kFirst = 8192;
    if (p->peak_cap < capacity) {
        if (p->d_peaks) (void)hipFree(p->d_peaks);
        if (p->h_peaks) (void)hipHostFree(p->h_peaks);
        p->d_peaks = nullptr; p->h_peaks = nullptr; p->peak_cap = 0;
        MFFT_HIP(hipMalloc((void**)&p->d_peaks, sizeof(int64_t) * (size_t)(capacity + kHead)));
        MFFT_HIP(hipHostMalloc((void**)&p->h_peaks, sizeof(int64_t) * (size_t)(capacity + kHead), hipHostMallocDefault));
        p->peak_cap = capacity;
    }
    double* d_stats = reinterpret_cast<double*>(p->d_peaks);
    MFFT_HIP(hipMemcpyAsync(p->d_in, h_in, in_bytes, hipMemcpyHostToDevice, nullptr));
    MFFT_HIP(hipMemsetAsync(p->d_peaks, 0, sizeof(int64_t) * kHead, nullptr));
    int rc = plan_exec_ordered(p, p->d_in, p->d_out, scale, mu, t0, dt, d_stats, nullptr);
    if (rc) return rc;
    rc = mfft_local_maxima(p->d_out, p->n, p->dtype, p->d_peaks + kHead, capacity, p->d_peaks + 3, nullptr);
    if (rc) return rc;
    const int64_t first = capacity < kFirst ? capacity : kFirst;
    MFFT_HIP(hipMemcpyAsync(p->h_peaks, p->d_peaks, sizeof(int64_t) * (size_t)(kHead + first), hipMemcpyDeviceToHost, nullptr));
    MFFT_HIP(hipMemcpy(h_out, p->d_out, out_bytes, hipMemcpyDeviceToHost));      // synchronises the stream
    MFFT_HIP(hipStreamSynchronize(nullptr));
    if (h_stats) memcpy(h_stats, p->h_peaks, 3 * sizeof(double));
    *h_count = p->h_peaks[3];
    const int64_t take = *h_count < capacity ? *h_count : capacity;
    if (take > first)
        MFFT_HIP(hipMemcpy(p->h_peaks + kHead + first, p->d_peaks + kHead + first, sizeof(int64_t) * (size_t)(take - first),
                           hipMemcpyDeviceToHost));
    if (take > 0) memcpy(h_idx, p->h_peaks + kHead, sizeof(int64_t) * (size_t)take);
    return MFFT_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// Is this host pointer page-locked memory the device can write directly (mfft_host_alloc, hipHostMalloc)?
static bool host_pinned(const void* h)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, h) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// The samples of a line search: handed over by the caller (h_x), or generated on the device from the
// eigen-data of `nblocks` small Hermitian blocks (the batch's return amplitudes f(t), final2.py:85-97).
struct LineSource {
    const void* h_x = nullptr;
    const AmpArgs* blocks = nullptr;      // one per signal of the batch
    // centroid demodulation (streamlit_app_combined.py:203-209): a rough pass through this plan -- the same transform
    // without MFFT_DEMOD -- gives mu = sum(omega S) / sum(S), which the main pass then demodulates by; mu goes to *h_mu
    mfft_plan rough = nullptr;
    double* h_mu = nullptr;
};

static int lines_run(mfft_plan plan, const LineSource& src, double scale, double mu, double t0, double dt,
                     const double* h_times, const mfft_peak_params* prm, int normalize, void* h_S, double* h_stats,
                     int64_t* h_idx, double* h_omega_hat, double* h_height, int32_t* h_count, double* h_gram)
{
    const void* h_x = src.h_x;
    if (!plan || (!h_x && !src.blocks) || !prm || !h_idx || !h_omega_hat || !h_count || !h_stats) { set_error("null argument"); return MFFT_E_INVALID; }
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (src.blocks) {
        if (!h_times) { set_error("the amplitude generator needs the time grid"); return MFFT_E_INVALID; }
        if (p->dtype != MFFT_C128 || (p->flags & MFFT_REAL_IN)) { set_error("the amplitude generator writes complex128 samples"); return MFFT_E_UNSUPPORTED; }
    }
    const uint32_t need = MFFT_POWER | MFFT_STATS;
    if ((p->flags & need) != need || p->kind == Plan::COLUMNS) { set_error("the line search needs an MFFT_POWER | MFFT_STATS plan"); return MFFT_E_INVALID; }
    if (p->lin.batch_stride || p->lin.seg || p->lout.batch_stride || p->lout.seg) { set_error("host entry points serve dense layouts only"); return MFFT_E_UNSUPPORTED; }
    if (prm->want < 1 || prm->want > 64) { set_error("1 <= want <= 64 lines"); return MFFT_E_INVALID; }
    if (prm->mode != 0 && prm->mode != 1) { set_error("peak mode must be 0 or 1"); return MFFT_E_INVALID; }
    if (!(prm->dt > 0.0) || !(prm->guard >= 0.0)) { set_error("bad peak parameters"); return MFFT_E_INVALID; }
    if (h_gram) {
        if (!h_times) { set_error("the line fit needs the time grid"); return MFFT_E_INVALID; }
        if (p->dtype != MFFT_C128 || (p->flags & (MFFT_REAL_IN | MFFT_DEMOD))) {
            set_error("the fused line fit takes complex128 samples without demodulation");
            return MFFT_E_UNSUPPORTED;
        }
    }
    if (p->batch > 65535) { set_error("at most 65535 spectra per call"); return MFFT_E_UNSUPPORTED; }
    Plan* rough = reinterpret_cast<Plan*>(src.rough);
    if (rough) {
        if (!src.h_mu || rough == p) { set_error("null argument"); return MFFT_E_INVALID; }
        if (!(p->flags & MFFT_DEMOD) || rough->flags != (p->flags & ~(uint32_t)MFFT_DEMOD) || rough->n != p->n || rough->n_in != p->n_in ||
            rough->batch != 1 || p->batch != 1 || rough->dtype != p->dtype || rough->device != p->device || rough->kind == Plan::COLUMNS ||
            rough->lin.batch_stride || rough->lin.seg || rough->lout.batch_stride || rough->lout.seg) {
            set_error("the rough pass takes the same one-signal plan without MFFT_DEMOD");
            return MFFT_E_INVALID;
        }
    }
    DeviceGuard g(p->device);
    if (!g.ok) { set_error("cannot select the plan's device"); return MFFT_E_HIP; }
    std::lock_guard<std::mutex> lock(p->mu);
    p->resident = false;
    const size_t rs = real_size(p->dtype);
    const int64_t B = p->batch, n = p->n, n_in = p->n_in;
    const int want = prm->want;
    const size_t in_bytes = (size_t)B * (size_t)n_in * rs * ((p->flags & MFFT_REAL_IN) ? 1 : 2);
    const size_t out_bytes = (size_t)B * (size_t)n * rs;
    if (!p->d_out) MFFT_HIP(hipMalloc(&p->d_out, out_bytes));
    // One device block and its pinned mirror, so that a call is one upload, one download of the small
    // results and one download of the spectra:
    //   up:    samples | time grid
    //   work:  candidate lists | candidate counts | line positions E
    //   down:  statistics | picked bins | omega_hat | heights | lines found | Gram blocks
    const int64_t cap = n / 2 + 2;
    const size_t nres = 2 * (size_t)want * want + 2 * (size_t)want + 1;
    const size_t o_x = 0, o_t = align256(o_x + in_bytes), up_bytes = o_t + sizeof(double) * (size_t)n_in;
    const size_t o_cidx = align256(up_bytes), o_cval = align256(o_cidx + sizeof(int64_t) * (size_t)(B * cap));
    const size_t o_count = align256(o_cval + sizeof(double) * (size_t)(B * cap));
    const size_t o_E = align256(o_count + sizeof(unsigned int) * (size_t)B);
    const size_t o_down = align256(o_E + sizeof(double) * (size_t)(B * want));
    const size_t o_stats = o_down, o_idx = o_stats + sizeof(double) * 3 * (size_t)B;
    const size_t o_om = o_idx + sizeof(int64_t) * (size_t)(B * want), o_h = o_om + sizeof(double) * (size_t)(B * want);
    const size_t o_n = o_h + sizeof(double) * (size_t)(B * want);
    const size_t o_gram = (o_n + sizeof(int) * (size_t)B + 7) & ~size_t(7);
    const size_t total = o_gram + sizeof(double) * (size_t)B * nres;
    const size_t down_bytes = total - o_down;
    if (!p->d_lines || p->lines_want != want) {
        if (p->d_lines) (void)hipFree(p->d_lines);
        if (p->h_lines) (void)hipHostFree(p->h_lines);
        if (p->d_lsq_part) (void)hipFree(p->d_lsq_part);       // sized by the line count
        p->d_lsq_part = nullptr; p->lsq_part_doubles = 0;
        p->d_lines = nullptr; p->h_lines = nullptr; p->lines_want = 0;
        MFFT_HIP(hipMalloc((void**)&p->d_lines, total));
        MFFT_HIP(hipHostMalloc((void**)&p->h_lines, up_bytes + down_bytes, hipHostMallocDefault));
        p->lines_want = want;
        p->lines_dirty = true;
    }
    // the whole call runs on a stream of the plan: calls on other plans (other sessions) overlap on the
    // device instead of queueing on the null stream
    if (!p->s_lines) MFFT_HIP(hipStreamCreateWithFlags(&p->s_lines, hipStreamNonBlocking));
    hipStream_t sm = p->s_lines;
    char* d = p->d_lines;
    char* h_up = p->h_lines;
    char* h_down = p->h_lines + up_bytes;
    int64_t* d_cidx = reinterpret_cast<int64_t*>(d + o_cidx);
    double* d_cval = reinterpret_cast<double*>(d + o_cval);
    unsigned int* d_cnt = reinterpret_cast<unsigned int*>(d + o_count);
    double* d_stats = reinterpret_cast<double*>(d + o_stats);
    int64_t* d_idx = reinterpret_cast<int64_t*>(d + o_idx);
    double* d_om = reinterpret_cast<double*>(d + o_om);
    double* d_h = reinterpret_cast<double*>(d + o_h);
    int* d_n = reinterpret_cast<int*>(d + o_n);
    double* d_gram = reinterpret_cast<double*>(d + o_gram);
    double* d_E = reinterpret_cast<double*>(d + o_E);
    const double* d_t = reinterpret_cast<const double*>(d + o_t);

    if (h_x) memcpy(h_up + o_x, h_x, in_bytes);
    if (h_gram || src.blocks) memcpy(h_up + o_t, h_times, sizeof(double) * (size_t)n_in);
    // no return may leave this stream (or the side stream below) working on the plan's buffers, or writing
    // into its pinned ones, behind the caller's back
    struct StreamSync {
        hipStream_t s; bool armed;
        ~StreamSync() { if (armed) (void)hipStreamSynchronize(s); }
    } main_sync{sm, true};
    if (src.blocks) {
        // only the time grid goes up; every block's f(t) is written where the transform reads it
        MFFT_HIP(hipMemcpyAsync(d + o_t, h_up + o_t, sizeof(double) * (size_t)n_in, hipMemcpyHostToDevice, sm));
        int64_t gblocks = (n_in + 255) / 256;
        if (gblocks > 4096) gblocks = 4096;
        for (int64_t b = 0; b < B; ++b)
            hipLaunchKernelGGL(return_amplitude_kernel, dim3((unsigned)gblocks), dim3(256), 0, sm,
                               reinterpret_cast<cx<double>*>(d + o_x) + b * n_in, n_in, d_t, 0.0, 0.0, src.blocks[b]);
        MFFT_HIP(hipGetLastError());
    } else {
        MFFT_HIP(hipMemcpyAsync(d, h_up, h_gram ? up_bytes : in_bytes, hipMemcpyHostToDevice, sm));
    }
    int rc = MFFT_OK;
    if (rough) {
        // rough spectrum -> its sum and first moment -> mu, on the same resident samples
        std::lock_guard<std::mutex> rough_lock(rough->mu);
        if (!rough->d_out) MFFT_HIP(hipMalloc(&rough->d_out, out_bytes));
        rc = plan_exec_ordered(rough, d + o_x, rough->d_out, scale, 0.0, 0.0, 1.0, d_stats, sm);
        if (rc) return rc;
        MFFT_HIP(hipMemcpyAsync(h_down + (o_stats - o_down), d_stats, sizeof(double) * 3, hipMemcpyDeviceToHost, sm));
        MFFT_HIP(hipStreamSynchronize(sm));
        const double* st = reinterpret_cast<const double*>(h_down + (o_stats - o_down));
        const double two_pi = 2.0 * 3.141592653589793;
        mu = st[1] > 0.0 ? (two_pi / ((double)n * prm->dt)) * st[2] / st[1] : 0.0;     // the reference's own expression, in its order
        *src.h_mu = mu;
    }
    rc = plan_exec_ordered(p, d + o_x, p->d_out, scale, mu, t0, dt, d_stats, sm);
    if (rc) return rc;
    // The spectra leave on a stream of their own as soon as the transform is done -- normalised into a
    // copy, because the line search reads the unnormalised ones meanwhile -- and land in pinned memory
    // (a download into a fresh pageable array runs at ~17 GB/s, one staging thread, page faults
    // included; from 1 MiB on the copy threads move them out of the pinned buffer ~2x faster).
    // Below 1 MiB the direct download is faster than the cross-stream hand-off (24016 bins: 0.18 against 0.22 ms per call).
    // A caller's page-locked array (mfft_host_alloc) takes the download itself: no staging buffer, no second copy.
    const bool direct = h_S != nullptr && host_pinned(h_S);
    bool side = h_S != nullptr && out_bytes >= (size_t(1) << 20) && (direct || out_bytes <= (size_t(256) << 20));
    if (side && !direct && !p->h_spec && hipHostMalloc((void**)&p->h_spec, out_bytes, hipHostMallocDefault) != hipSuccess) { p->h_spec = nullptr; side = false; }
    if (side && !p->s_spec && hipStreamCreateWithFlags(&p->s_spec, hipStreamNonBlocking) != hipSuccess) { p->s_spec = nullptr; side = false; }
    if (side && !p->ev_spec && hipEventCreateWithFlags(&p->ev_spec, hipEventDisableTiming) != hipSuccess) { p->ev_spec = nullptr; side = false; }
    if (side && normalize && !p->d_norm && hipMalloc(&p->d_norm, out_bytes) != hipSuccess) { p->d_norm = nullptr; side = false; }
    (void)hipGetLastError();
    const auto normalise = [&](void* dst, hipStream_t s) -> int {
        // S / (S.max() if S.max() > 0 else 1.0), per spectrum (final2.py:146)
        int64_t nblocks = (n + 255) / 256;
        if (nblocks > 4096) nblocks = 4096;
        if (p->dtype == MFFT_C64)
            hipLaunchKernelGGL(normalize_kernel<float>, dim3((unsigned)nblocks, (unsigned)B), dim3(256), 0, s,
                               static_cast<const float*>(p->d_out), static_cast<float*>(dst), n, (const double*)d_stats, (int64_t)3);
        else
            hipLaunchKernelGGL(normalize_kernel<double>, dim3((unsigned)nblocks, (unsigned)B), dim3(256), 0, s,
                               static_cast<const double*>(p->d_out), static_cast<double*>(dst), n, (const double*)d_stats, (int64_t)3);
        MFFT_HIP(hipGetLastError());
        return MFFT_OK;
    };
    StreamSync side_sync{p->s_spec, side};
    if (side) {
        MFFT_HIP(hipEventRecord(p->ev_spec, sm));
        MFFT_HIP(hipStreamWaitEvent(p->s_spec, p->ev_spec, 0));
        if (normalize) { rc = normalise(p->d_norm, p->s_spec); if (rc) return rc; }
        MFFT_HIP(hipMemcpyAsync(direct ? h_S : p->h_spec, normalize ? p->d_norm : p->d_out, out_bytes, hipMemcpyDeviceToHost, p->s_spec));
    }
    PeakArgs a;
    a.n = n; a.cap = cap;
    a.inv_nd = 1.0 / ((double)n * prm->dt);
    a.dc_thresh = prm->dc_guard; a.guard = prm->guard;
    a.include_dc = prm->include_dc; a.mode = prm->mode; a.want = want;
    // the pick kernel leaves the candidate counters at zero; they are cleared here only after a
    // (re)allocation or when an earlier call stopped between the two kernels
    if (p->lines_dirty) MFFT_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * (size_t)B, sm));
    p->lines_dirty = true;
    int64_t blocks = (n + kPeakChunk - 1) / kPeakChunk;
    if (blocks > 1024) blocks = 1024;
    const int pick_threads = n >= 16384 ? 1024 : 256;
    if (p->dtype == MFFT_C64) {
        hipLaunchKernelGGL(peak_candidates_kernel<float>, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, sm,
                           static_cast<const float*>(p->d_out), a, d_cidx, d_cval, d_cnt);
        hipLaunchKernelGGL(peak_pick_kernel<float>, dim3((unsigned)B), dim3(pick_threads), 0, sm,
                           static_cast<const float*>(p->d_out), a, d_cidx, d_cval, d_cnt, d_idx, d_om, d_h, d_n, h_gram ? d_E : nullptr);
    } else {
        hipLaunchKernelGGL(peak_candidates_kernel<double>, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, sm,
                           static_cast<const double*>(p->d_out), a, d_cidx, d_cval, d_cnt);
        hipLaunchKernelGGL(peak_pick_kernel<double>, dim3((unsigned)B), dim3(pick_threads), 0, sm,
                           static_cast<const double*>(p->d_out), a, d_cidx, d_cval, d_cnt, d_idx, d_om, d_h, d_n, h_gram ? d_E : nullptr);
    }
    MFFT_HIP(hipGetLastError());
    p->lines_dirty = false;
    if (h_gram) {
        // The reduction's partial sums live in a block of the plan (not in the scratch of
        // mfft_lsq_normal_static, which belongs to that entry point's callers): sized for the whole batch
        // up to 16 MiB, larger batches go through it in sub-batches.
        const int nchunk = (want + kLsqCols - 1) / kLsqCols;
        const size_t per_sig = (size_t)want * nchunk * kLsqSlot;          // doubles per signal and sample block
        if (!p->d_lsq_part) {
            size_t doubles = per_sig * (size_t)B * 64;
            const size_t cap = (size_t(16) << 20) / sizeof(double);
            if (doubles > cap) doubles = cap < per_sig ? per_sig : cap;
            MFFT_HIP(hipMalloc((void**)&p->d_lsq_part, doubles * sizeof(double)));
            p->lsq_part_doubles = doubles;
        }
        int64_t per = (int64_t)(p->lsq_part_doubles / per_sig);
        if (per > 65535 / nchunk) per = 65535 / nchunk;
        if (per < 1) { set_error("line fit does not fit the reduction scratch"); return MFFT_E_UNSUPPORTED; }
        for (int64_t b0 = 0; b0 < B; b0 += per) {
            const int64_t nb = B - b0 < per ? B - b0 : per;
            rc = lsq_normal_launch(reinterpret_cast<const cx<double>*>(d + o_x) + b0 * n_in,
                                   (p->flags & MFFT_HANN) ? static_cast<const double*>(p->d_win) : nullptr, d_t, n_in,
                                   nb, nullptr, d_E + b0 * want, want, d_gram + (size_t)b0 * nres, p->d_lsq_part,
                                   p->lsq_part_doubles, sm, "mfft_lines_host");
            if (rc) return rc;
        }
    }
    MFFT_HIP(hipMemcpyAsync(h_down, d + o_down, h_gram ? down_bytes : (o_gram - o_down), hipMemcpyDeviceToHost, sm));
    if (h_S && !side) {                                   // small spectra, or too large for a pinned copy
        if (normalize) { rc = normalise(p->d_out, sm); if (rc) return rc; }
        MFFT_HIP(hipMemcpyAsync(h_S, p->d_out, out_bytes, hipMemcpyDeviceToHost, sm));
    }
    uint64_t ticket = 0;
    const bool pooled = side && !direct;
    if (side) {
        const hipError_t e_side = hipStreamSynchronize(p->s_spec);
        side_sync.armed = false;
        if (e_side != hipSuccess) return hip_fail(e_side, "spectrum download", __FILE__, __LINE__);
        if (pooled) ticket = CopyPool::out().start(h_S, p->h_spec, out_bytes);
    }
    const hipError_t e_main = hipStreamSynchronize(sm);
    main_sync.armed = false;
    if (e_main != hipSuccess) { if (pooled) CopyPool::out().wait(ticket); return hip_fail(e_main, "line search", __FILE__, __LINE__); }
    memcpy(h_stats, h_down + (o_stats - o_down), sizeof(double) * 3 * (size_t)B);
    memcpy(h_idx, h_down + (o_idx - o_down), sizeof(int64_t) * (size_t)(B * want));
    memcpy(h_omega_hat, h_down + (o_om - o_down), sizeof(double) * (size_t)(B * want));
    if (h_height) memcpy(h_height, h_down + (o_h - o_down), sizeof(double) * (size_t)(B * want));
    memcpy(h_count, h_down + (o_n - o_down), sizeof(int) * (size_t)B);
    if (h_gram) memcpy(h_gram, h_down + (o_gram - o_down), sizeof(double) * (size_t)B * nres);
    if (pooled) CopyPool::out().wait(ticket);
    p->resident = src.blocks != nullptr && p->dtype == MFFT_C128 && !(p->flags & MFFT_REAL_IN);   // samples and time grid stay in d_lines
    return MFFT_OK;
}

int mfft_lines_host(mfft_plan plan, const void* h_x, double scale, double mu, double t0, double dt,
                    const double* h_times, const mfft_peak_params* prm, int normalize, void* h_S, double* h_stats,
                    int64_t* h_idx, double* h_omega_hat, double* h_height, int32_t* h_count, double* h_gram)
{
    if (!h_x) { set_error("null argument"); return MFFT_E_INVALID; }
    LineSource src;
    src.h_x = h_x;
    return lines_run(plan, src, scale, mu, t0, dt, h_times, prm, normalize, h_S, h_stats, h_idx, h_omega_hat, h_height, h_count, h_gram);
}

int mfft_lines_blocks_host(mfft_plan plan, const double* h_w, const double* h_c0, const double* h_vrow, int nk, double hbar,
                           double scale, const double* h_times, const mfft_peak_params* prm, int normalize, void* h_S,
                           double* h_stats, int64_t* h_idx, double* h_omega_hat, double* h_height, int32_t* h_count,
                           double* h_gram)
{
    if (!plan || !h_w || !h_c0 || !h_vrow) { set_error("null argument"); return MFFT_E_INVALID; }
    const int64_t B = reinterpret_cast<Plan*>(plan)->batch;
    std::vector<AmpArgs> blocks((size_t)B);
    for (int64_t b = 0; b < B; ++b) {
        const int rc = amp_args(&blocks[(size_t)b], h_w + b * nk, h_c0 + 2 * b * nk, h_vrow + 2 * b * nk, nk, hbar);
        if (rc) return rc;
    }
    LineSource src;
    src.blocks = blocks.data();
    return lines_run(plan, src, scale, 0.0, 0.0, 1.0, h_times, prm, normalize, h_S, h_stats, h_idx, h_omega_hat, h_height, h_count, h_gram);
}

int mfft_lines_extract_host(mfft_plan plan, mfft_plan rough_plan, const double* h_w, const double* h_c0, const double* h_vrow,
                            int nk, double hbar, double scale, double t0, double dt, const double* h_times,
                            const mfft_peak_params* prm, void* h_S, double* h_stats, int64_t* h_idx, double* h_omega_hat,
                            double* h_height, int32_t* h_count, double* h_mu)
{
    if (!plan || !h_w || !h_c0 || !h_vrow || !h_mu) { set_error("null argument"); return MFFT_E_INVALID; }
    if (reinterpret_cast<Plan*>(plan)->batch != 1) { set_error("one signal per call"); return MFFT_E_INVALID; }
    AmpArgs block;
    const int rc = amp_args(&block, h_w, h_c0, h_vrow, nk, hbar);
    if (rc) return rc;
    *h_mu = 0.0;
    LineSource src;
    src.blocks = &block;
    src.rough = rough_plan;
    src.h_mu = h_mu;
    return lines_run(plan, src, scale, 0.0, t0, dt, h_times, prm, 0, h_S, h_stats, h_idx, h_omega_hat, h_height, h_count, nullptr);
}

int mfft_lines_gram_host(mfft_plan plan, const double* h_E, int m, double mu, double* h_gram)
{
    if (!plan || !h_E || !h_gram) { set_error("null argument"); return MFFT_E_INVALID; }
    if (m < 1 || m > 64) { set_error("1 <= m <= 64 columns"); return MFFT_E_INVALID; }
    Plan* p = reinterpret_cast<Plan*>(plan);
    DeviceGuard g(p->device);
    if (!g.ok) { set_error("cannot select the plan's device"); return MFFT_E_HIP; }
    std::lock_guard<std::mutex> lock(p->mu);
    if (!p->resident || !p->d_lines || !p->s_lines || p->batch != 1) {
        set_error("no resident samples: mfft_lines_extract_host / mfft_lines_blocks_host of this plan must come first");
        return MFFT_E_INVALID;
    }
    constexpr size_t kCols = 64, kRes = 2 * kCols * kCols + 2 * kCols + 1;
    const int nchunk = (m + kLsqCols - 1) / kLsqCols;
    const size_t part = (size_t)64 * 4 * kLsqSlot * 64;                    // 64 columns x 4 chunks x 64 sample blocks
    const size_t head = kCols + 1 + kRes;
    if (!p->d_fit) {
        MFFT_HIP(hipMalloc((void**)&p->d_fit, (head + part) * sizeof(double)));
        MFFT_HIP(hipHostMalloc((void**)&p->h_fit, head * sizeof(double), hipHostMallocDefault));
    }
    const size_t nres = 2 * (size_t)m * m + 2 * (size_t)m + 1;
    const size_t in_bytes = (size_t)p->n_in * 2 * sizeof(double);
    hipStream_t sm = p->s_lines;
    for (int i = 0; i < m; ++i) p->h_fit[i] = h_E[i];
    p->h_fit[kCols] = mu;
    MFFT_HIP(hipMemcpyAsync(p->d_fit, p->h_fit, (kCols + 1) * sizeof(double), hipMemcpyHostToDevice, sm));
    const int rc = lsq_normal_launch(p->d_lines, (p->flags & MFFT_HANN) ? static_cast<const double*>(p->d_win) : nullptr,
                                     reinterpret_cast<const double*>(p->d_lines + align256(in_bytes)), p->n_in, 1, p->d_fit + kCols,
                                     p->d_fit, m, p->d_fit + kCols + 1, p->d_fit + head, part, sm, "mfft_lines_gram_host");
    (void)nchunk;
    if (rc) { (void)hipStreamSynchronize(sm); return rc; }
    hipError_t e = hipMemcpyAsync(p->h_fit + kCols + 1, p->d_fit + kCols + 1, nres * sizeof(double), hipMemcpyDeviceToHost, sm);
    if (e == hipSuccess) e = hipStreamSynchronize(sm);
    if (e != hipSuccess) return hip_fail(e, "mfft_lines_gram_host", __FILE__, __LINE__);
    memcpy(h_gram, p->h_fit + kCols + 1, nres * sizeof(double));
    return MFFT_OK;
}

int mfft_normalize_by_max(const void* d_in, void* d_out, int64_t count, const double* d_max, int dtype,
                          void* stream)
{
    if (!d_in || !d_out || !d_max || count < 0) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (count == 0) return MFFT_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == MFFT_C64)
        hipLaunchKernelGGL(normalize_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const float*>(d_in), static_cast<float*>(d_out), count, d_max, (int64_t)0);
    else if (dtype == MFFT_C128)
        hipLaunchKernelGGL(normalize_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const double*>(d_in), static_cast<double*>(d_out), count, d_max, (int64_t)0);
    else { set_error("bad dtype"); return MFFT_E_INVALID; }
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_host_alloc(void** out, int64_t bytes)
{
    if (!out || bytes < 1) { set_error("bad argument"); return MFFT_E_INVALID; }
    *out = nullptr;
    MFFT_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return MFFT_OK;
}

int mfft_host_free(void* p)
{
    if (!p) return MFFT_OK;
    MFFT_HIP(hipHostFree(p));
    return MFFT_OK;
}

int mfft_transpose(const void* d_in, void* d_out, int64_t rows, int64_t cols, int64_t batch, int dtype,
                   void* stream)
{
    if (!d_in || !d_out || rows < 1 || cols < 1 || batch < 1 || d_in == d_out) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (batch > 65535 || (rows + 31) / 32 > 65535) { set_error("transpose grid too large"); return MFFT_E_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
    if (dtype == MFFT_C64)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, static_cast<const cx<float>*>(d_in),
                           static_cast<cx<float>*>(d_out), rows, cols);
    else if (dtype == MFFT_C128)
        hipLaunchKernelGGL(transpose_kernel<double>, grid, dim3(256), 0, s, static_cast<const cx<double>*>(d_in),
                           static_cast<cx<double>*>(d_out), rows, cols);
    else { set_error("bad dtype"); return MFFT_E_INVALID; }
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_swap_axes_strided(const void* d_in, void* d_out, int64_t a, int64_t b, int64_t w, int64_t in_stride_a,
                           int64_t in_stride_b, int dtype, void* stream)
{
    if (!d_in || !d_out || a < 1 || b < 1 || w < 1 || d_in == d_out || in_stride_a < 0 || in_stride_b < 0) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (dtype != MFFT_C64 && dtype != MFFT_C128) { set_error("bad dtype"); return MFFT_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = a * b;
    const int64_t bytes_per_row = w * 2 * (int64_t)real_size(dtype);
    const int block = bytes_per_row >= 4096 ? 256 : 64;      // 16 bytes per lane per trip
    const int64_t cap = (int64_t)device_cu_count() * 32;
    const unsigned grid = (unsigned)(rows < cap ? rows : cap);
    if (dtype == MFFT_C64)
        hipLaunchKernelGGL(swap_axes_kernel<float>, dim3(grid), dim3(block), 0, s, static_cast<const cx<float>*>(d_in),
                           static_cast<cx<float>*>(d_out), a, b, w, in_stride_a, in_stride_b);
    else
        hipLaunchKernelGGL(swap_axes_kernel<double>, dim3(grid), dim3(block), 0, s, static_cast<const cx<double>*>(d_in),
                           static_cast<cx<double>*>(d_out), a, b, w, in_stride_a, in_stride_b);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

int mfft_swap_axes(const void* d_in, void* d_out, int64_t a, int64_t b, int64_t w, int dtype, void* stream)
{
    if (a < 1 || b < 1 || w < 1) { set_error("bad argument"); return MFFT_E_INVALID; }
    return mfft_swap_axes_strided(d_in, d_out, a, b, w, b * w, w, dtype, stream);
}

int mfft_twiddle_tile(void* d_x, int64_t rows, int64_t cols, int64_t r0, int64_t c0, int64_t n_total,
                      int dtype, void* stream)
{
    if (!d_x || rows < 1 || cols < 1 || r0 < 0 || c0 < 0 || n_total < 1) { set_error("bad argument"); return MFFT_E_INVALID; }
    if (n_total > (int64_t(1) << 31)) { set_error("n_total above 2^31"); return MFFT_E_UNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype != MFFT_C64 && dtype != MFFT_C128) { set_error("bad dtype"); return MFFT_E_INVALID; }
    // Two-level table, k = hi * 2^lo_bits + lo, kept per (device, n_total): a job uses one or two
    // moduli, so the few most recent tables stay on the device and a call is one kernel launch.
    using Table = TwiddleTable;
    std::mutex& tab_mu = g_tab_mu;
    Table (&tabs)[8] = g_tabs;
    uint64_t& tick = g_tab_tick;
    int device = 0;
    MFFT_HIP(hipGetDevice(&device));
    const cx<double>* d_tab = nullptr;
    int lo_bits = 0;
    int64_t lo_n = 0;
    {
        std::lock_guard<std::mutex> lock(tab_mu);
        Table* hit = nullptr;
        Table* victim = &tabs[0];
        for (Table& t : tabs) {
            if (t.d && t.device == device && t.n == n_total) { hit = &t; break; }
            if (t.used < victim->used) victim = &t;
        }
        if (!hit) {
            const int total_bits = ilog2(n_total);
            const int lb = total_bits / 2;
            const int64_t ln = int64_t(1) << lb;
            const int64_t hn = (n_total + ln - 1) / ln;
            std::vector<cx<double>> h((size_t)(ln + hn));
            for (int64_t k = 0; k < ln; ++k) {
                long double c, d;
                unit_root(k, n_total, &c, &d);
                h[(size_t)k] = cx<double>{(double)c, (double)d};
            }
            for (int64_t k = 0; k < hn; ++k) {
                long double c, d;
                unit_root(k * ln, n_total, &c, &d);
                h[(size_t)(ln + k)] = cx<double>{(double)c, (double)d};
            }
            cx<double>* fresh = nullptr;
            MFFT_HIP(hipMalloc((void**)&fresh, h.size() * sizeof(cx<double>)));
            const hipError_t e = hipMemcpy(fresh, h.data(), h.size() * sizeof(cx<double>), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(fresh); return hip_fail(e, "hipMemcpy", __FILE__, __LINE__); }
            if (victim->d) {                      // hipFree waits for kernels that may still read it
                int prev = device;
                (void)hipSetDevice(victim->device);
                (void)hipFree(victim->d);
                (void)hipSetDevice(prev);
            }
            victim->device = device; victim->n = n_total; victim->d = fresh; victim->lo_bits = lb;
            hit = victim;
        }
        hit->used = ++tick;
        d_tab = hit->d;
        lo_bits = hit->lo_bits;
        lo_n = int64_t(1) << lo_bits;
    }
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == MFFT_C64)
        hipLaunchKernelGGL(twiddle_tile_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<cx<float>*>(d_x), rows, cols, r0, c0, n_total, d_tab, d_tab + lo_n, lo_bits);
    else
        hipLaunchKernelGGL(twiddle_tile_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<cx<double>*>(d_x), rows, cols, r0, c0, n_total, d_tab, d_tab + lo_n, lo_bits);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

}  // extern "C"
