// Last global pass of the HALF-length transform of a packed real signal, with the split into the signal's own
// spectrum fused into its store (np.fft.fft of a real x, final2.py:142-143 with a real f_t).
//
// x has n = 2M real samples; z[j] = x[2j] + i x[2j+1] has the M-point transform Z, and
//     X[k] = E + w^k O,   X[M-k] = conj(E - w^k O),   X[n-k] = conj X[k],   X[M+k] = conj X[M-k],
//     E = (Z[k] + conj Z[M-k]) / 2,   O = -i (Z[k] - conj Z[M-k]) / 2,   w = exp(-2 pi i / n),
// so bin k needs bin M-k beside it.  In the last pass (radix G, NS = M / G super-butterflies per signal) bin
// k = j + q NS is output q of column j, and M - k = (NS - j) + (G-1-q) NS is output G-1-q of column NS - j: a
// workgroup therefore transforms TWO column sets per trip -- A = [j0, j0 + C) and its mirror B = [NS-j0-C+1, NS-j0] --
// one after the other through the same LDS image, turns B through LDS once more so that every thread finds the
// partner of each A output it holds, and stores the four bins of every pair.  That replaces a separate split
// kernel (one more read of M and write of n elements) and the M-element intermediate.
//
// Columns 0 and NS/2 are their own mirrors (output q pairs with output (G-q) mod G resp. G-1-q of the same column).
// The first trip of a signal carries them -- column 0 in A, column NS/2 in the slot of B that would be column NS --
// and writes their 2 G raw outputs to a side buffer, from which r2c_special_kernel forms their bins.
#pragma once
#include "tile_kernel.h"

namespace mfft {

template <typename T, int LOG2G, int LOG2C>
__global__ void __launch_bounds__(((TileCfg<T, LOG2G>::TX) << LOG2C))
fft_tile_r2c_kernel(const TArgs<T> a)
{
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

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C_* lds = reinterpret_cast<C_*>(smem_raw);
    C_* twl = lds + LDS_ELEMS;
    const int c = threadIdx.x % C;
    const int t = threadIdx.x / C;
    for (int i = threadIdx.x; i < tw_total_elems<RD>(); i += BLOCK) twl[i] = a.tw[i];
    __syncthreads();

    const auto image = [&](int pos, int col) -> C_& {
        if constexpr (PITCH == C) return lds[col * COLPITCH + lds_pad<PADSHIFT>(pos)];
        else return lds[tile_addr<PADSHIFT, PITCH>(pos, col)];
    };
    const int64_t NS = a.NS;                         // = L in the last pass
    const int64_t M = NS << LOG2G, n = 2 * M;
    const int log2m = a.log2_ns + LOG2G, log2n = log2m + 1;
    const int log2_pairs = a.log2_tiles - 1;
    const int64_t total = a.batch << log2_pairs;
    const int64_t pstep = (int64_t)TX * a.L;

    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t b = w >> log2_pairs;
        const uint32_t u = (uint32_t)(w & ((int64_t(1) << log2_pairs) - 1));
        const bool first = u == 0;                               // uniform: the trip that carries columns 0 and NS/2
        const uint32_t j0 = u * C;
        const uint32_t jA = j0 + (uint32_t)c;
        uint32_t jB = (uint32_t)NS - j0 - (uint32_t)C + 1u + (uint32_t)c;
        if (first && c == C - 1) jB = (uint32_t)(NS >> 1);
        const C_* __restrict__ sig = reinterpret_cast<const C_*>(a.in) + b * a.in_stride;

        // column jl of this signal: load, inter-pass twiddle, G-point DFT -> v[e] = output t + TX*e
        const auto column_dft = [&](uint32_t jl, C_ (&v)[P]) {
            const C_* __restrict__ p = sig + jl + (int64_t)t * a.L;
            if (a.nt & 1) {
#pragma unroll
                for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, true>(p); p += pstep; }
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) { v[e] = tile_ld<T, false>(p); p += pstep; }
            }
            cx<double> wt = tile_root((uint64_t)jl * (uint32_t)t, log2m);
            const cx<double> step = tile_root((uint64_t)jl * (uint32_t)TX, log2m);
            if constexpr (sizeof(T) == 4 && P % 4 == 0) {
                const cx<double> s2 = cmul(step, step), s3 = cmul(s2, step), s4 = cmul(s2, s2);
                const C_ S1{(T)step.x, (T)step.y}, S2{(T)s2.x, (T)s2.y}, S3{(T)s3.x, (T)s3.y};
#pragma unroll
                for (int k = 0; k < P / 4; ++k) {
                    const C_ B{(T)wt.x, (T)wt.y};
                    constexpr bool VEC = MFFT_TILE_VEC != 0;
                    v[4 * k] = cmul<T, VEC>(v[4 * k], B);
                    v[4 * k + 1] = cmul<T, VEC>(v[4 * k + 1], cmul<T, VEC>(B, S1));
                    v[4 * k + 2] = cmul<T, VEC>(v[4 * k + 2], cmul<T, VEC>(B, S2));
                    v[4 * k + 3] = cmul<T, VEC>(v[4 * k + 3], cmul<T, VEC>(B, S3));
                    wt = cmul(wt, s4);
                }
            } else {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    const double xr = (double)v[e].x, xi = (double)v[e].y;
                    v[e] = C_{(T)(xr * wt.x - xi * wt.y), (T)(xr * wt.y + xi * wt.x)};
                    wt = cmul(wt, step);
                }
            }
            pass_compute<T, RD, 0, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
            if constexpr (NPASS > 1) {
                tile_exchange<T, RD, 0, P, TX, PADSHIFT, PITCH>(v, t, c, lds);
                pass_compute<T, RD, 1, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
            }
            if constexpr (NPASS > 2) {
                tile_exchange<T, RD, 1, P, TX, PADSHIFT, PITCH>(v, t, c, lds);
                pass_compute<T, RD, 2, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
            }
            if constexpr (NPASS > 3) {
                tile_exchange<T, RD, 2, P, TX, PADSHIFT, PITCH>(v, t, c, lds);
                pass_compute<T, RD, 3, P, TX, TW_LDS, (MFFT_TILE_VEC != 0)>(v, t, a.tw, nullptr, twl);
            }
        };
        C_ va[P], vb[P];
        column_dft(jB, vb);
        column_dft(jA, va);

        if (first) {
            // the self-mirrored columns' raw outputs: side[b][0][q] = column 0, side[b][1][q] = column NS/2
            C_* __restrict__ sd = a.side + b * (2 * G);
            if (c == 0) {
#pragma unroll
                for (int e = 0; e < P; ++e) sd[t + TX * e] = va[e];
            }
            if (c == C - 1) {
#pragma unroll
                for (int e = 0; e < P; ++e) sd[G + t + TX * e] = vb[e];
            }
        }
        // B through LDS: the partner of A's output q in column c is B's output G-1-q in column C-1-c
        __syncthreads();
#pragma unroll
        for (int e = 0; e < P; ++e) image(t + TX * e, c) = vb[e];
        __syncthreads();
        if (!(first && c == 0)) {
            // w^k for k = jA + (t + TX e) NS: base and step evaluated, stepped in double
            cx<double> wk = tile_root((uint64_t)jA + (uint64_t)t * (uint64_t)NS, log2n);
            const cx<double> wstep = tile_root((uint64_t)TX * (uint64_t)NS, log2n);
            const double sc = (double)a.scale;
            const int64_t k0 = (int64_t)jA + (int64_t)t * NS, kstep = (int64_t)TX * NS;
            const int64_t ob = b * a.out_stride;
#pragma unroll
            for (int e = 0; e < P; ++e) {
                const C_ zm = image(G - 1 - (t + TX * e), C - 1 - c);
                const double ar = (double)va[e].x, ai = (double)va[e].y, br = (double)zm.x, bi = (double)zm.y;
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);           // E
                const double orr = 0.5 * (ai + bi), oi = -0.5 * (ar - br);         // O = -i (Z[k] - conj Z[M-k]) / 2
                const double pr = wk.x * orr - wk.y * oi, pi = wk.x * oi + wk.y * orr;   // w^k O
                const double xr = (er + pr) * sc, xi = (ei + pi) * sc;             // X[k]
                const double yr = (er - pr) * sc, yi = (ei - pi) * sc;             // X[M+k] = conj X[M-k]
                const int64_t k = k0 + e * kstep;
                if (a.r2c_power) {
                    T* __restrict__ o = reinterpret_cast<T*>(a.out) + ob;
                    const T sx = (T)(xr * xr + xi * xi), sy = (T)(yr * yr + yi * yi);
                    o[k] = sx; o[n - k] = sx; o[M - k] = sy; o[M + k] = sy;
                } else {
                    C_* __restrict__ o = reinterpret_cast<C_*>(a.out) + ob;
                    o[k] = C_{(T)xr, (T)xi}; o[n - k] = C_{(T)xr, (T)(-xi)};
                    o[M - k] = C_{(T)yr, (T)(-yi)}; o[M + k] = C_{(T)yr, (T)yi};
                }
                wk = cmul(wk, wstep);
            }
        }
        // (the next trip's first exchange starts with a barrier: the image is free by then)
    }
}

// Bins of the two self-mirrored columns from their raw outputs: k = q NS (q = 0 .. G: X[0] .. X[M]) from column
// 0, k = NS/2 + q NS (q < G) from column NS/2; X[n-k] beside each.  One thread per k.
template <typename T, bool POWER>
__global__ void __launch_bounds__(256) r2c_special_kernel(const cx<T>* side, void* out, int64_t NS, int G, int log2n,
                                                          int64_t out_stride, T scale)
{
    const int64_t b = blockIdx.y;
    const cx<T>* sd = side + b * (2 * (int64_t)G);
    const int64_t M = NS * G, n = 2 * M;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= 2 * G; i += gridDim.x * blockDim.x) {
        cx<T> A, B;
        int64_t k;
        if (i <= G) { k = (int64_t)i * NS; A = sd[i % G]; B = sd[(G - i) % G]; }
        else { const int q = i - G - 1; k = (NS >> 1) + (int64_t)q * NS; A = sd[G + q]; B = sd[G + (G - 1 - q)]; }
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

template <typename T, int LOG2G, int LOG2C>
int launch_tile_r2c_one(const TArgs<T>& a, hipStream_t s)
{
    constexpr int BLOCK = TileCfg<T, LOG2G>::TX << LOG2C;
    constexpr size_t lds = tile_lds_bytes<T, LOG2G, LOG2C>();
    auto kern = fft_tile_r2c_kernel<T, LOG2G, LOG2C>;
    static thread_local int attr_dev = -1, per_cu = 1;
    int dev = 0;
    MFFT_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        MFFT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_cu = blocks_per_cu(reinterpret_cast<const void*>(kern), BLOCK, lds);
        attr_dev = dev;
    }
    if (a.log2_tiles < 1 || a.side == nullptr || a.log2_w != 0) { set_error("fused real-input pass needs at least two tiles per signal"); return MFFT_E_INVALID; }
    const int64_t total = a.batch << (a.log2_tiles - 1);
    if (total <= 0) return MFFT_OK;
    if (a.batch > 65535) { set_error("at most 65535 real signals per call above one LDS residency"); return MFFT_E_UNSUPPORTED; }
    hipLaunchKernelGGL(kern, dim3(pick_grid(per_cu, total)), dim3(BLOCK), lds, s, a);
    MFFT_HIP(hipGetLastError());
    const int G = 1 << LOG2G;
    const dim3 grid((unsigned)((2 * G + 1 + 255) / 256), (unsigned)a.batch);
    const int log2n = a.log2_ns + LOG2G + 1;
    if (a.r2c_power)
        hipLaunchKernelGGL((r2c_special_kernel<T, true>), grid, dim3(256), 0, s, (const cx<T>*)a.side, a.out, a.NS, G, log2n, a.out_stride, a.scale);
    else
        hipLaunchKernelGGL((r2c_special_kernel<T, false>), grid, dim3(256), 0, s, (const cx<T>*)a.side, a.out, a.NS, G, log2n, a.out_stride, a.scale);
    MFFT_HIP(hipGetLastError());
    return MFFT_OK;
}

template <typename T, int LOG2G>
int launch_tile_r2c_width(const TArgs<T>& a, hipStream_t s)
{
    constexpr int LC = TileCols<T, LOG2G, false>::LOG2C;
    if constexpr (sizeof(T) == 8 && LOG2G == 10 && LC != 2) {
        if (a.narrow & 2) return launch_tile_r2c_one<T, LOG2G, 2>(a, s);
    }
    return launch_tile_r2c_one<T, LOG2G, LC>(a, s);
}

template <typename T>
int launch_tile_r2c_any(int log2g, const TArgs<T>& a, hipStream_t s)
{
    switch (log2g) {
    case 4: return launch_tile_r2c_width<T, 4>(a, s);
    case 5: return launch_tile_r2c_width<T, 5>(a, s);
    case 6: return launch_tile_r2c_width<T, 6>(a, s);
    case 7: return launch_tile_r2c_width<T, 7>(a, s);
    case 8: return launch_tile_r2c_width<T, 8>(a, s);
    case 9: return launch_tile_r2c_width<T, 9>(a, s);
    case 10: return launch_tile_r2c_width<T, 10>(a, s);
    case 11: if constexpr (sizeof(T) == 8) return launch_tile_r2c_width<T, 11>(a, s); else break;
    }
    set_error("no fused real-input pass at this tile radix");
    return MFFT_E_UNSUPPORTED;
}

}  // namespace mfft
