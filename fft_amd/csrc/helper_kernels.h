// Element-wise and reduction kernels around the transforms: window/demodulation/chirp tables,
// Bluestein filter, max-normalisation, transposes and block copies of the multi-GPU routes, the
// normal equations of the line fit, the return-amplitude generator, the local-maxima scan and the
// one-point "transform".  Included by plan.hip only.
#pragma once
#include "common.h"

namespace mfft {


// pre[j] = win[j] * exp(+i*mu*(t0 + j*dt)) * extra[j]   (win / extra optional)
template <typename T>
__global__ void premul_kernel(cx<T>* pre, const T* win, const cx<T>* extra, int n, double mu, double t0,
                              double dt, int demod)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double re = 1.0, im = 0.0;
    if (demod) {
        double s, c;
        sincos(mu * (t0 + (double)j * dt), &s, &c);
        re = c;
        im = s;
    }
    if (win) { re *= (double)win[j]; im *= (double)win[j]; }
    if (extra) {
        const double er = (double)extra[j].x, ei = (double)extra[j].y;
        const double r2 = re * er - im * ei, i2 = re * ei + im * er;
        re = r2;
        im = i2;
    }
    pre[j] = cx<T>{(T)re, (T)im};
}

// chirp[j] = exp(-i*pi*j^2/n), j < n: j^2 is reduced mod 2n in integers (j < 2^31, so j^2 fits 64
// bits), then to the nearest quarter turn, and the remaining angle |x| <= 1/4 (in units of pi) goes
// through sincospi -- the phase error stays at the 1e-16 level for every j, as the long-double host
// tables have it, at a fraction of their cost (the apps' largest length, 1.28e8 points: 12 s of host
// sinl/cosl).
template <typename T>
__global__ void chirp_kernel(cx<T>* chirp, int64_t n)
{
    const uint64_t N = 2 * (uint64_t)n;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = ((uint64_t)j * (uint64_t)j) % N;
        const uint64_t k4 = 4 * k;
        const uint64_t q = (k4 + N / 2) / N;                          // nearest quarter turn
        const int64_t rem = (int64_t)k4 - (int64_t)(q * N);           // |rem| <= N/2
        double sa, ca;
        sincospi((double)rem / (double)(2 * N), &sa, &ca);            // angle = 2*pi*rem/(4N)
        double c, sn;
        switch (q & 3) {
        case 0: c = ca; sn = sa; break;
        case 1: c = -sa; sn = ca; break;
        case 2: c = -ca; sn = -sa; break;
        default: c = sa; sn = -ca; break;
        }
        chirp[j] = cx<T>{(T)c, (T)(-sn)};
    }
}

// Bluestein filter b[d] = conj(chirp[|d|]) for -n_in < d < n, wrapped into m points, zero elsewhere
template <typename T>
__global__ void chirp_filter_kernel(const cx<T>* chirp, cx<T>* b, int64_t n, int64_t n_in, int64_t m)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        cx<T> v{0, 0};
        if (i < n) v = cx<T>{chirp[i].x, -chirp[i].y};
        else if (m - i < n_in) v = cx<T>{chirp[m - i].x, -chirp[m - i].y};
        b[i] = v;
    }
}

// Spectrum blockIdx.y (count elements each) divided by its own guarded maximum, which sits
// max_stride doubles after the previous spectrum's.
template <typename T>
__global__ void normalize_kernel(const T* in, T* out, int64_t count, const double* d_max, int64_t max_stride = 0)
{
    const int64_t b = blockIdx.y;
    const double m = d_max[b * max_stride];
    const T den = (T)(m > 0.0 ? m : 1.0);
    in += b * count;
    out += b * count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i] / den;
}

// Per-spectrum statistics {max S, sum S, sum f_k S} (f_k the signed FFT-order bin) of dense power
// spectra, as a two-stage reduction without atomics: block (x, b) reduces its share of spectrum b
// into partial[(b * gridDim.x + x) * 3 ..], stats_final_kernel folds the partials in a fixed order.
// Used after multi-pass transforms, whose last pass would otherwise send three atomics per wave to
// the same three addresses (105 us of a 150 us chirp-z execution of three 2^19-point signals).
template <typename T>
__global__ void __launch_bounds__(256) stats_partial_kernel(const T* S, int64_t n_out, int64_t stride, double* partial)
{
    __shared__ double red[4][3];
    const int64_t b = blockIdx.y;
    const T* s = S + b * stride;
    double mx = 0.0, sm = 0.0, sk = 0.0;
    const int64_t half = (n_out + 1) / 2;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)s[k];
        const double f = k < half ? (double)k : (double)(k - n_out);
        mx = v > mx ? v : mx;
        sm += v;
        sk += f * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double omx = __shfl_xor(mx, o, 64);
        mx = omx > mx ? omx : mx;
        sm += __shfl_xor(sm, o, 64);
        sk += __shfl_xor(sk, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = mx; red[threadIdx.x >> 6][1] = sm; red[threadIdx.x >> 6][2] = sk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m2 = red[0][0], s2 = red[0][1], k2 = red[0][2];
        for (int w = 1; w < 4; ++w) { m2 = red[w][0] > m2 ? red[w][0] : m2; s2 += red[w][1]; k2 += red[w][2]; }
        double* out = partial + (b * gridDim.x + blockIdx.x) * 3;
        out[0] = m2; out[1] = s2; out[2] = k2;
    }
}

__global__ void __launch_bounds__(64) stats_final_kernel(const double* partial, int nblk, double* stats)
{
    const int64_t b = blockIdx.x;
    const double* p = partial + b * nblk * 3;
    double mx = 0.0, sm = 0.0, sk = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        mx = p[3 * i] > mx ? p[3 * i] : mx;
        sm += p[3 * i + 1];
        sk += p[3 * i + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double omx = __shfl_xor(mx, o, 64);
        mx = omx > mx ? omx : mx;
        sm += __shfl_xor(sm, o, 64);
        sk += __shfl_xor(sk, o, 64);
    }
    if (threadIdx.x == 0) { stats[3 * b] = mx; stats[3 * b + 1] = sm; stats[3 * b + 2] = sk; }
}

// out[b][c][r] = in[b][r][c], 32x32 tiles through LDS (33-wide rows: no bank conflicts)
template <typename T>
__global__ void transpose_kernel(const cx<T>* in, cx<T>* out, int64_t rows, int64_t cols)
{
    __shared__ cx<T> tile[32][33];
    const int64_t b = blockIdx.z;
    const cx<T>* src = in + b * rows * cols;
    cx<T>* dst = out + b * rows * cols;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int64_t r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = src[r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) dst[c * rows + r] = tile[tx][i];
    }
}

// dense[b][j] <-> seg[b*batch_stride + (j/seg)*seg_stride + j%seg] for `count` elements per signal of
// type E (to_dense = gather, else scatter): stages the rare combinations of a segmented layout
// with a runtime prologue / epilogue, which the LDS-resident general variants do not address
template <typename E>
__global__ void seg_copy_kernel(const E* src, E* dst, int64_t batch, int64_t count, int64_t batch_stride,
                                int64_t seg, int64_t seg_stride, int to_dense)
{
    const int64_t total = batch * count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / count, j = i - b * count;
        const int64_t at = b * batch_stride + seg_index(j, seg, seg_stride);
        if (to_dense) dst[i] = src[at];
        else dst[at] = src[i];
    }
}

// Spectrum of a real signal of n = 2M samples from the M-point transform Z of its packed form
// z[j] = x[2j] + i x[2j+1]:
//     X[k] = (Z[k] + conj Z[M-k]) / 2  -  i w^k (Z[k] - conj Z[M-k]) / 2,   w = exp(-2 pi i / n),  0 <= k <= M
//     X[n-k] = conj X[k]                                                      0 <  k <  M
// (indices of Z modulo M).  One thread per k: Z is read once forwards and once backwards, every bin of the
// full spectrum is written once -- as scale * X, or as |scale * X|^2 when POWER.  The root w^k is evaluated
// (exact integer reduction, tile_root) in double for both precisions.
template <typename T, bool POWER>
__global__ void __launch_bounds__(256) r2c_split_kernel(const cx<T>* Z, void* out, int64_t M, int log2n, int64_t z_stride,
                                                        int64_t out_stride, T scale)
{
    const int64_t b = blockIdx.y;
    const cx<T>* z = Z + b * z_stride;
    const int64_t n = 2 * M;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= M; k += (int64_t)gridDim.x * blockDim.x) {
        const cx<T> A = z[k & (M - 1)], B = z[(M - k) & (M - 1)];
        const cx<double> w = tile_root((uint64_t)k, log2n);
        const double er = 0.5 * ((double)A.x + (double)B.x), ei = 0.5 * ((double)A.y - (double)B.y);
        const double orr = 0.5 * ((double)A.y + (double)B.y), oi = -0.5 * ((double)A.x - (double)B.x);
        const double xr = (er + (w.x * orr - w.y * oi)) * (double)scale, xi = (ei + (w.x * oi + w.y * orr)) * (double)scale;
        if constexpr (POWER) {
            T* o = static_cast<T*>(out) + b * out_stride;
            const T sp = (T)(xr * xr + xi * xi);
            o[k] = sp;
            if (k > 0 && k < M) o[n - k] = sp;
        } else {
            cx<T>* o = static_cast<cx<T>*>(out) + b * out_stride;
            o[k] = cx<T>{(T)xr, (T)xi};
            if (k > 0 && k < M) o[n - k] = cx<T>{(T)xr, (T)(-xi)};
        }
    }
}

// out[b][a][:] = in[a*sa + b*sb + (0..W)] for complex elements (sa = B*W, sb = W: an (A, B, W) array with
// its two leading axes swapped): one row of W contiguous elements per loop trip, copied 16 bytes per
// lane where W and the strides allow
template <typename T>
__global__ void swap_axes_kernel(const cx<T>* in, cx<T>* out, int64_t A, int64_t B, int64_t W, int64_t sa, int64_t sb)
{
    constexpr int PER16 = 16 / (int)sizeof(cx<T>);            // elements per 16-byte unit
    const int64_t rows = A * B;
    const bool wide = W % PER16 == 0 && sa % PER16 == 0 && sb % PER16 == 0 &&
                      (reinterpret_cast<size_t>(in) & 15) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int64_t a = row / B, b = row - a * B;
        const cx<T>* src = in + a * sa + b * sb;
        cx<T>* dst = out + (b * A + a) * W;
        if (wide) {
            using V4 = float __attribute__((ext_vector_type(4)));
            const V4* s4 = reinterpret_cast<const V4*>(src);
            V4* d4 = reinterpret_cast<V4*>(dst);
            const int64_t units = W / PER16;
            for (int64_t u = threadIdx.x; u < units; u += blockDim.x)
                __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + u), d4 + u);
        } else {
            for (int64_t u = threadIdx.x; u < W; u += blockDim.x) dst[u] = src[u];
        }
    }
}

// x[r][c] *= exp(-2*pi*i*(r0+r)*(c0+c)/n): two-level table product, tables built on the host
template <typename T>
__global__ void twiddle_tile_kernel(cx<T>* x, int64_t rows, int64_t cols, int64_t r0, int64_t c0,
                                    int64_t n, const cx<double>* lo, const cx<double>* hi, int lo_bits)
{
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        // indices are reduced mod n first, and n <= 2^31, so the product fits 64 bits
        const uint64_t prod = ((uint64_t)(r0 + r) % (uint64_t)n) * ((uint64_t)(c0 + c) % (uint64_t)n);
        const int64_t k = (int64_t)(prod % (uint64_t)n);
        const cx<double> a = lo[k & ((int64_t(1) << lo_bits) - 1)], b = hi[k >> lo_bits];
        const double wr = a.x * b.x - a.y * b.y, wi = a.x * b.y + a.y * b.x;
        const cx<T> v = x[i];
        x[i] = cx<T>{(T)((double)v.x * wr - (double)v.y * wi), (T)((double)v.x * wi + (double)v.y * wr)};
    }
}

// Normal equations of the windowed line fit.  Columns b_a[j] = win[j] exp(-i E_a t_j), target
// y[j] = x[j] win[j] exp(+i mu t_j).  blockIdx.y = a selects the row: out holds
//   G[a][b] = sum_j conj(b_a[j]) b_b[j]   (2*m*m doubles, row-major, re/im interleaved)
//   r[a]    = sum_j conj(b_a[j]) y[j]     (2*m doubles)
//   yy      = sum_j |y[j]|^2              (1 double, accumulated by row 0)
struct LsqArgs {
    int m;            // all columns (rows of G): blockIdx.y
    int c0, nc;       // this launch fills columns [c0, c0 + nc), nc <= 16
    double mu;
    double E[64];
};

__global__ void __launch_bounds__(256) lsq_normal_kernel(const cx<double>* x, const double* win, const double* times,
                                                         int64_t n, LsqArgs a, double* out)
{
    const int row = blockIdx.y;
    const int m = a.m, nc = a.nc;
    double gr[16], gi[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) { gr[b] = 0.0; gi[b] = 0.0; }
    double rr = 0.0, ri = 0.0, yy = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const double t = times[j];
        const double w = win ? win[j] : 1.0;
        double sa, ca;
        sincos(a.E[row] * t, &sa, &ca);                // conj(b_a) = w (ca + i sa)
        const double ar = w * ca, ai = w * sa;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            if (b < nc) {
                double sb, cb;
                sincos(a.E[a.c0 + b] * t, &sb, &cb);   // b_b = w (cb - i sb)
                const double br = w * cb, bi = -w * sb;
                gr[b] += ar * br - ai * bi;
                gi[b] += ar * bi + ai * br;
            }
        }
        if (a.c0 == 0) {                               // right-hand side and target energy: first launch only
            cx<double> y = x[j];
            y.x *= w; y.y *= w;
            if (a.mu != 0.0) {
                double sm, cm;
                sincos(a.mu * t, &sm, &cm);
                y = cx<double>{y.x * cm - y.y * sm, y.x * sm + y.y * cm};
            }
            rr += ar * y.x - ai * y.y;
            ri += ar * y.y + ai * y.x;
            yy += y.x * y.x + y.y * y.y;
        }
    }
    // wave reduction, then one atomic per wave and value
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            if (b < nc) { gr[b] += __shfl_xor(gr[b], o, 64); gi[b] += __shfl_xor(gi[b], o, 64); }
        }
        rr += __shfl_xor(rr, o, 64);
        ri += __shfl_xor(ri, o, 64);
        yy += __shfl_xor(yy, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        double* G = out + 2 * ((size_t)row * m + a.c0);
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            if (b < nc) { atomicAdd(G + 2 * b, gr[b]); atomicAdd(G + 2 * b + 1, gi[b]); }
        }
        if (a.c0 == 0) {
            double* r = out + 2 * (size_t)m * m + 2 * row;
            atomicAdd(r, rr);
            atomicAdd(r + 1, ri);
            if (row == 0) atomicAdd(out + 2 * (size_t)m * m + 2 * m, yy);
        }
    }
}

// The same normal equations for `batch` signals at once, reduced without atomics: block
// (bx, row, sig*nchunk + chunk) sums its share of the samples for 16 columns of row `row` of
// signal `sig` and writes 36 doubles {16 x (re, im) of G, re r, im r, yy, pad} to
//   partial[(((sig*m + row)*nchunk + chunk)*nblk + bx)*36 ...]        (plain stores)
// and lsq_final_kernel adds the nblk partials of every value in a fixed order, so results do
// not depend on scheduling.  E holds m line positions per signal, mu one shift per signal (or
// nullptr), times and win are shared by all signals.
constexpr int kLsqCols = 16;
constexpr int kLsqSlot = 2 * kLsqCols + 4;
struct LsqBatchArgs {
    const cx<double>* x;      // batch * n samples
    const double* win;        // n, or nullptr
    const double* times;      // n
    const double* mu;         // batch, or nullptr
    const double* E;          // batch * m
    double* partial;
    int64_t n;
    int m, nchunk, nblk;
};

__global__ void __launch_bounds__(256) lsq_partial_kernel(const LsqBatchArgs a)
{
    __shared__ double red[4][kLsqSlot];
    const int row = blockIdx.y;
    const int sig = blockIdx.z / a.nchunk, chunk = blockIdx.z - sig * a.nchunk;
    const int c0 = chunk * kLsqCols;
    const int nc = a.m - c0 < kLsqCols ? a.m - c0 : kLsqCols;
    const double* E = a.E + (size_t)sig * a.m;
    const double e_row = E[row];
    double e_col[kLsqCols];
#pragma unroll
    for (int b = 0; b < kLsqCols; ++b) e_col[b] = E[c0 + (b < nc ? b : 0)];
    const double mu = a.mu ? a.mu[sig] : 0.0;
    const bool rhs = chunk == 0;
    const cx<double>* x = a.x + (size_t)sig * a.n;
    double gr[kLsqCols], gi[kLsqCols];
#pragma unroll
    for (int b = 0; b < kLsqCols; ++b) { gr[b] = 0.0; gi[b] = 0.0; }
    double rr = 0.0, ri = 0.0, yy = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n; j += (int64_t)gridDim.x * blockDim.x) {
        const double t = a.times[j];
        const double w = a.win ? a.win[j] : 1.0;
        double sa, ca;
        sincos(e_row * t, &sa, &ca);                  // conj(b_row) = w (ca + i sa)
        const double ar = w * ca, ai = w * sa;
#pragma unroll
        for (int b = 0; b < kLsqCols; ++b) {
            if (b < nc) {
                double sb, cb;
                sincos(e_col[b] * t, &sb, &cb);       // b_col = w (cb - i sb)
                const double br = w * cb, bi = -w * sb;
                gr[b] += ar * br - ai * bi;
                gi[b] += ar * bi + ai * br;
            }
        }
        if (rhs) {
            cx<double> y = x[j];
            y.x *= w; y.y *= w;
            if (mu != 0.0) {
                double sm, cm;
                sincos(mu * t, &sm, &cm);
                y = cx<double>{y.x * cm - y.y * sm, y.x * sm + y.y * cm};
            }
            rr += ar * y.x - ai * y.y;
            ri += ar * y.y + ai * y.x;
            yy += y.x * y.x + y.y * y.y;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int b = 0; b < kLsqCols; ++b) { gr[b] += __shfl_xor(gr[b], o, 64); gi[b] += __shfl_xor(gi[b], o, 64); }
        rr += __shfl_xor(rr, o, 64);
        ri += __shfl_xor(ri, o, 64);
        yy += __shfl_xor(yy, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int b = 0; b < kLsqCols; ++b) { red[wave][2 * b] = gr[b]; red[wave][2 * b + 1] = gi[b]; }
        red[wave][2 * kLsqCols] = rr;
        red[wave][2 * kLsqCols + 1] = ri;
        red[wave][2 * kLsqCols + 2] = yy;
        red[wave][2 * kLsqCols + 3] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x < kLsqSlot) {
        const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        a.partial[((((size_t)sig * a.m + row) * a.nchunk + chunk) * a.nblk + blockIdx.x) * kLsqSlot + threadIdx.x] = v;
    }
}

// out[sig] = {G (2*m*m, row-major, re/im interleaved), r (2*m), yy}: every value is the sum of its
// nblk <= 64 partials, one wave per value -- lane b holds block b's partial and a fixed butterfly adds
// them, so the result does not depend on scheduling (a single thread walking the nblk strided
// partials took 10.6 us for three lines of 8001 samples: a chain of dependent cache misses) -- and
// every value of `out` is written exactly once.
__global__ void __launch_bounds__(256) lsq_final_kernel(const double* partial, double* out, int batch, int m, int nchunk,
                                                        int nblk)
{
    const int64_t total = (int64_t)batch * m * nchunk * kLsqSlot;
    const size_t nres = 2 * (size_t)m * m + 2 * (size_t)m + 1;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < total; i += nwaves) {          // wave-uniform
        const int v = (int)(i % kLsqSlot);
        const int64_t q = i / kLsqSlot;                 // (sig*m + row)*nchunk + chunk
        const int chunk = (int)(q % nchunk);
        const int64_t sr = q / nchunk;
        const int row = (int)(sr % m);
        const int64_t sig = sr / m;
        const int c0 = chunk * kLsqCols;
        const int nc = m - c0 < kLsqCols ? m - c0 : kLsqCols;
        double* o = out + (size_t)sig * nres;
        double* dst = nullptr;
        if (v < 2 * nc) dst = o + 2 * ((size_t)row * m + c0) + v;
        else if (chunk == 0 && (v == 2 * kLsqCols || v == 2 * kLsqCols + 1)) dst = o + 2 * (size_t)m * m + 2 * row + (v - 2 * kLsqCols);
        else if (chunk == 0 && row == 0 && v == 2 * kLsqCols + 2) dst = o + nres - 1;
        if (!dst) continue;
        const double* p = partial + (size_t)q * nblk * kLsqSlot + v;
        double sum = lane < nblk ? p[(size_t)lane * kLsqSlot] : 0.0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s, 64);
        if (lane == 0) *dst = sum;
    }
}

// f[j] = sum_k vrow[k] * (exp(-i w[k] t_j / hbar) * c0[k]); the small per-block constants
// travel in the kernel argument buffer
struct AmpArgs {
    int nk;
    double hbar;
    double w[32];
    double c0[64];
    double vrow[64];
};

__global__ void return_amplitude_kernel(cx<double>* out, int64_t n, const double* times, double t0, double dt,
                                        const AmpArgs a)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const double t = times ? times[j] : t0 + (double)j * dt;
        double fr = 0.0, fi = 0.0;
        for (int k = 0; k < a.nk; ++k) {
            double sn, cs;
            sincos((a.w[k] * t) / a.hbar, &sn, &cs);
            // tmp = exp(-i phi) * c0[k]
            const double cr = a.c0[2 * k], ci = a.c0[2 * k + 1];
            const double tr = cs * cr + sn * ci, ti = cs * ci - sn * cr;
            // f += vrow[k] * tmp
            const double vr = a.vrow[2 * k], vi = a.vrow[2 * k + 1];
            fr += vr * tr - vi * ti;
            fi += vr * ti + vi * tr;
        }
        out[j] = cx<double>{fr, fi};
    }
}

// indices of interior local maxima (left strict, right weak), unordered, with a total count
template <typename T>
__global__ void local_maxima_kernel(const T* S, int64_t n, int64_t* idx, int64_t capacity,
                                    unsigned long long* count)
{
    for (int64_t i = 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n - 1;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T c = S[i];
        if (c > S[i - 1] && c >= S[i + 1]) {
            const unsigned long long slot = atomicAdd(count, 1ULL);
            if ((int64_t)slot < capacity) idx[slot] = i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Peak pipeline of the two estimators on the device copy of the spectra (final2.py:151-171,
// streamlit_app_combined.py:218-228): candidates = interior local maxima (left strict, right weak),
// plus bin 0 / minus the DC guard band in mode 0; then the guard-spaced greedy pick in order of
// decreasing height; then the three-point log-parabola refinement of every picked line.
// omega_k = two_pi * (f_k * inv_nd) with f_k the signed FFT-order bin and inv_nd = 1 / (n * dt):
// the same two multiplications numpy makes, so the guard comparisons see the reference's doubles.
// ---------------------------------------------------------------------------------------------
struct PeakArgs {
    int64_t n;            // bins per spectrum
    int64_t cap;          // candidate slots per spectrum (n/2 + 2 always suffices)
    double inv_nd;        // 1.0 / (n * dt)
    double dc_thresh;     // mode 0 without include_dc: keep candidates with |omega| > dc_thresh
    double guard;         // picked lines are farther apart than this (mode 1: in |omega|)
    int include_dc;       // mode 0: bin 0 is a candidate when S[0] > 0, and no DC guard applies
    int mode;             // 0: distance |w_k - w_j|, 1: distance ||w_k| - |w_j||, no DC handling
    int want;             // lines to pick per spectrum, <= 64
};

__device__ __forceinline__ double peak_omega(int64_t k, int64_t n, double inv_nd)
{
    // two separately rounded products, as numpy evaluates them: no fused multiply-add may absorb
    // them into a caller's subtraction (omega_k - omega_k has to be exactly zero)
#pragma clang fp contract(off)
    const int64_t f = k < (n + 1) / 2 ? k : k - n;
    const double fr = (double)f * inv_nd;
    return 6.283185307179586 * fr;
}

// cand_idx / cand_val: [batch][cap] (unordered), count: [batch] (zeroed by the caller).  A workgroup
// scans one contiguous chunk of a spectrum, collects its candidates in LDS and reserves their
// slots with ONE global atomic (a counter per spectrum hit once per candidate serialised the scan:
// 138 us for three 256032-bin spectra against 6 us).
constexpr int kPeakChunk = 2048;                 // bins per workgroup: at most 1024 local maxima + bin 0
template <typename T>
__global__ void __launch_bounds__(256) peak_candidates_kernel(const T* S, PeakArgs a, int64_t* cand_idx, double* cand_val,
                                                             unsigned int* count)
{
    __shared__ int64_t l_idx[kPeakChunk / 2 + 2];
    __shared__ double l_val[kPeakChunk / 2 + 2];
    __shared__ unsigned int l_n, l_base;
    const int64_t b = blockIdx.y;
    const T* s = S + b * a.n;
    for (int64_t c0 = (int64_t)blockIdx.x * kPeakChunk; c0 < a.n; c0 += (int64_t)gridDim.x * kPeakChunk) {
        if (threadIdx.x == 0) l_n = 0;
        __syncthreads();
        const int64_t c1 = c0 + kPeakChunk < a.n ? c0 + kPeakChunk : a.n;
        for (int64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
            const T c = s[i];
            bool take = i >= 1 && i <= a.n - 2 && c > s[i - 1] && c >= s[i + 1];
            if (a.mode == 0) {
                if (a.include_dc) take = take || (i == 0 && c > (T)0);
                else take = take && fabs(peak_omega(i, a.n, a.inv_nd)) > a.dc_thresh;
            }
            if (take) {
                const unsigned int slot = atomicAdd(&l_n, 1u);
                l_idx[slot] = i;
                l_val[slot] = (double)c;
            }
        }
        __syncthreads();
        const unsigned int found = l_n;
        if (threadIdx.x == 0 && found > 0) l_base = atomicAdd(count + b, found);
        __syncthreads();
        const unsigned int base = l_base;
        for (unsigned int k = threadIdx.x; k < found; k += blockDim.x) {
            if ((int64_t)(base + k) < a.cap) {
                cand_idx[b * a.cap + base + k] = l_idx[k];
                cand_val[b * a.cap + base + k] = l_val[k];
            }
        }
        __syncthreads();
    }
}

// One workgroup per spectrum.  Round r takes the tallest remaining candidate (ties: the larger bin
// index, as a reversed stable sort would) and strikes every candidate inside its guard band, so a
// round compares with the newest line only.  out_idx / out_omega / out_height: [batch][want], padded
// with -1 / 0; out_count: [batch]; out_E (optional): -omega_hat, the E = -omega convention of
// final2.py:172, zero beyond the picked lines.  cand_val may be consumed (struck entries become -1 when the
// list is too long for the register form) and the candidate counter is left at zero for the next scan.
template <typename T>
__global__ void __launch_bounds__(1024) peak_pick_kernel(const T* S, PeakArgs a, const int64_t* cand_idx, double* cand_val,
                                                       unsigned int* count, int64_t* out_idx, double* out_omega,
                                                       double* out_height, int* out_count, double* out_E)
{
    __shared__ double red_val[16];
    __shared__ int64_t red_idx[16];
    __shared__ int64_t chosen[64];
    __shared__ int n_chosen;
    const int64_t b = blockIdx.x;
    const T* s = S + b * a.n;
    const int64_t* ci = cand_idx + b * a.cap;
    double* cv = cand_val + b * a.cap;
    const int64_t nc = (int64_t)count[b] < a.cap ? (int64_t)count[b] : a.cap;
    if (threadIdx.x == 0) n_chosen = 0;
    __syncthreads();
    // Usual case (a zero-padded spectrum has about one local maximum per input sample): every thread keeps
    // its share of the candidates -- height, bin, omega -- in registers, so a round is a register scan, one
    // block reduction and a register strike instead of two dependent sweeps over global memory
    // (three rounds over 8000 candidates: 26 -> 19 us under the profiler).  Longer lists take the global-memory form below.
    constexpr int KREG = 8;
    const bool in_regs = nc <= (int64_t)KREG * blockDim.x;
    double rv[KREG], rw[KREG];
    int ri[KREG];
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < KREG; ++q) {
            const int64_t j = threadIdx.x + (int64_t)q * blockDim.x;
            const bool have = j < nc;
            rv[q] = have ? cv[j] : -1.0;
            ri[q] = have ? (int)ci[j] : -1;
            rw[q] = have ? peak_omega(ci[j], a.n, a.inv_nd) : 0.0;
        }
    }
    for (int r = 0; r < a.want; ++r) {
        double best = -1.0;
        int64_t best_i = -1;
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < KREG; ++q) {
                const double v = rv[q];
                const int64_t k = ri[q];
                if (v > best || (v == best && v >= 0.0 && k > best_i)) { best = v; best_i = k; }
            }
        } else {
            for (int64_t j = threadIdx.x; j < nc; j += blockDim.x) {
                const double v = cv[j];
                const int64_t k = ci[j];
                if (v > best || (v == best && v >= 0.0 && k > best_i)) { best = v; best_i = k; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int64_t oi = __shfl_xor(best_i, o, 64);
            if (ov > best || (ov == best && oi > best_i)) { best = ov; best_i = oi; }
        }
        if ((threadIdx.x & 63) == 0) { red_val[threadIdx.x >> 6] = best; red_idx[threadIdx.x >> 6] = best_i; }
        __syncthreads();
        best = red_val[0]; best_i = red_idx[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (red_val[w] > best || (red_val[w] == best && red_idx[w] > best_i)) { best = red_val[w]; best_i = red_idx[w]; }
        __syncthreads();                                   // everyone has read red_* before the next round writes
        if (best_i < 0 || best < 0.0) break;               // uniform: no candidate left
        if (threadIdx.x == 0) { chosen[r] = best_i; n_chosen = r + 1; }
        const double wk = peak_omega(best_i, a.n, a.inv_nd);
        if (in_regs) {
#pragma unroll
            for (int q = 0; q < KREG; ++q) {
                const double d = a.mode == 0 ? fabs(rw[q] - wk) : fabs(fabs(rw[q]) - fabs(wk));
                if (rv[q] >= 0.0 && (ri[q] == best_i || !(d > a.guard))) rv[q] = -1.0;
            }
        } else {
            for (int64_t j = threadIdx.x; j < nc; j += blockDim.x) {
                if (cv[j] < 0.0) continue;
                const double wj = peak_omega(ci[j], a.n, a.inv_nd);
                const double d = a.mode == 0 ? fabs(wj - wk) : fabs(fabs(wj) - fabs(wk));
                if (ci[j] == best_i || !(d > a.guard)) cv[j] = -1.0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    const int got = n_chosen;
    if (threadIdx.x == 0) { out_count[b] = got; count[b] = 0; }
    if ((int)threadIdx.x < a.want) {
        const int r = threadIdx.x;
        int64_t i = -1;
        double wh = 0.0, h = 0.0;
        if (r < got) {
            i = chosen[r];
            const double x1 = peak_omega(i, a.n, a.inv_nd);
            h = (double)s[i];
            wh = x1;
            if (i > 0 && i < a.n - 1) {                    // three-point log-parabola (final2.py:102-113)
                const double y0 = (double)s[i - 1], y1 = h, y2 = (double)s[i + 1];
                const double l0 = log(y0 > 1e-300 ? y0 : 1e-300), l1 = log(y1 > 1e-300 ? y1 : 1e-300),
                             l2 = log(y2 > 1e-300 ? y2 : 1e-300);
                const double denom = 2.0 * (l0 - 2.0 * l1 + l2);
                if (denom != 0.0) {
                    const double delta = (l0 - l2) / denom;
                    const double dx = 0.5 * (peak_omega(i + 1, a.n, a.inv_nd) - peak_omega(i - 1, a.n, a.inv_nd));
                    wh = x1 + delta * dx;
                }
            }
        }
        out_idx[b * a.want + r] = i;
        out_omega[b * a.want + r] = wh;
        out_height[b * a.want + r] = h;
        if (out_E) out_E[b * a.want + r] = r < got ? -wh : 0.0;
    }
}

// out[b] = f(in[b]) for 1-point "transforms" (np.fft.fft(x, n=1)): the prologue and
// epilogue of the general path applied to sample 0.
template <typename T>
__global__ void n1_kernel(const void* in, void* out, int64_t batch, int64_t in_stride, int real_in,
                          const T* win, const cx<T>* pre, T scale, int conj, int power, double* stats)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    cx<T> x{0, 0};
    if (real_in) x.x = static_cast<const T*>(in)[b * in_stride];
    else x = static_cast<const cx<T>*>(in)[b * in_stride];
    if (conj) x.y = -x.y;
    if (win) { x.x *= win[0]; x.y *= win[0]; }
    if (pre) x = cmul(x, pre[0]);
    x.x *= scale;
    x.y *= scale;
    if (conj) x.y = -x.y;
    if (power) {
        const T s = x.x * x.x + x.y * x.y;
        static_cast<T*>(out)[b] = s;
        if (stats) { stats[3 * b] = (double)s; stats[3 * b + 1] = (double)s; stats[3 * b + 2] = 0.0; }
    } else {
        static_cast<cx<T>*>(out)[b] = x;
    }
}

}  // namespace mfft
