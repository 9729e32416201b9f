// Butterflies and Stockham index algebra shared by every kernel of libmfft.so.
//
// Everything here is a pure function of registers: no memory access, no thread
// ids.  The templates compile for gfx950 device code (hipcc) and for the host.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MFFT_HD __host__ __device__ __forceinline__ constexpr
#else
#define MFFT_HD inline constexpr
#endif

// Template parameter V of the butterflies ("packed form"): complex64 device code written on two-lane
// vectors.  Written on scalars, "+/- i" turns and complex products reach the vectoriser as an add
// lane beside a subtract lane and cost lane moves on top of the packed instructions; as vector
// operands whose lanes are swapped and negated they become source modifiers of one packed
// instruction (16384-point kernel: 1128 -> 643 vector instructions per signal and thread).  Which
// kernels use it is a measured choice of their own (lds_kernel.h: vec_default).
#if defined(__HIP_DEVICE_COMPILE__)
#define MFFT_VEC_MATH(T, V) ((V) && sizeof(T) == 4)
#else
#define MFFT_VEC_MATH(T, V) false
#endif

namespace mfft {

template <typename T> struct cx { T x, y; };

#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> using vec2_t = T __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ vec2_t<T> to_v(cx<T> a) { return vec2_t<T>{a.x, a.y}; }
template <typename T> __device__ __forceinline__ cx<T> to_c(vec2_t<T> a) { return cx<T>{a.x, a.y}; }
// (a.y, -a.x) = -i a
template <typename T> __device__ __forceinline__ vec2_t<T> v_mi(vec2_t<T> a) { return __builtin_shufflevector(a, -a, 1, 2); }
#endif

template <typename T> MFFT_HD cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> MFFT_HD cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
#if defined(__HIP_DEVICE_COMPILE__)
// Packed complex product: (a.x b.x, a.x b.y), then + (a.y (-b.y), a.y b.x).  Lane selection and the
// sign are source modifiers of the two packed instructions (left to the compiler, a third one is
// spent on a lane move).
__device__ __forceinline__ vec2_t<float> cmul_pk(vec2_t<float> A, vec2_t<float> B)
{
    vec2_t<float> t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(A), "v"(B), "v"(t));
    return r;
}
#endif
template <typename T, bool V = false> MFFT_HD cx<T> cmul(cx<T> a, cx<T> b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MFFT_VEC_MATH(T, V)) return to_c<T>(cmul_pk(to_v(a), to_v(b)));
#endif
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// a * conj(b)
template <typename T> MFFT_HD cx<T> cmulc(cx<T> a, cx<T> b)
{
    return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
// multiply by -i (forward quarter turn) / +i
template <typename T> MFFT_HD cx<T> mul_mi(cx<T> a) { return {a.y, -a.x}; }
template <typename T> MFFT_HD cx<T> mul_pi(cx<T> a) { return {-a.y, a.x}; }

// ---------------------------------------------------------------------------
// Forward DFT of R points held in registers, natural order in and out:
//   v[q] <- sum_i v[i] * exp(-2*pi*i * i*q / R)
// R = 2, 4, 8, 16.  Built as decimation in time from radix-4 kernels with the
// small constant twiddles folded in (W8^1 = (1-i)/sqrt2 etc.).
// ---------------------------------------------------------------------------

template <typename T> MFFT_HD void bfly2(cx<T>& a, cx<T>& b)
{
    cx<T> t = a - b;
    a = a + b;
    b = t;
}

template <typename T, bool V = false> MFFT_HD void dft4(cx<T>& a0, cx<T>& a1, cx<T>& a2, cx<T>& a3)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MFFT_VEC_MATH(T, V)) {
        using v2 = vec2_t<T>;
        const v2 A0 = to_v(a0), A1 = to_v(a1), A2 = to_v(a2), A3 = to_v(a3);
        const v2 s02 = A0 + A2, d02 = A0 - A2, s13 = A1 + A3, d13 = v_mi<T>(A1 - A3);
        a0 = to_c<T>(s02 + s13); a1 = to_c<T>(d02 + d13); a2 = to_c<T>(s02 - s13); a3 = to_c<T>(d02 - d13);
        return;
    }
#endif
    cx<T> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = mul_mi(a1 - a3);
    a0 = s02 + s13;
    a1 = d02 + d13;
    a2 = s02 - s13;
    a3 = d02 - d13;
}

template <typename T> struct consts {
    static constexpr T s2 = (T)0.70710678118654752440084436210484903928L;   // sqrt(1/2)
    static constexpr T c16 = (T)0.92387953251128675612818318939678828682L;  // cos(pi/8)
    static constexpr T s16 = (T)0.38268343236508977172845998403039886676L;  // sin(pi/8)
};

// a * W8^1 = a * (1 - i)/sqrt2 ; a * W8^3 = a * (-1 - i)/sqrt2
template <typename T, bool V = false> MFFT_HD cx<T> mul_w8_1(cx<T> a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MFFT_VEC_MATH(T, V)) { const vec2_t<T> A = to_v(a); return to_c<T>((A + v_mi<T>(A)) * consts<T>::s2); }
#endif
    return {(a.x + a.y) * consts<T>::s2, (a.y - a.x) * consts<T>::s2};
}
template <typename T, bool V = false> MFFT_HD cx<T> mul_w8_3(cx<T> a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MFFT_VEC_MATH(T, V)) { const vec2_t<T> A = to_v(a); return to_c<T>((v_mi<T>(A) - A) * consts<T>::s2); }
#endif
    return {(a.y - a.x) * consts<T>::s2, -(a.x + a.y) * consts<T>::s2};
}

template <typename T, bool V = false> MFFT_HD void dft8(cx<T>* v)
{
    // two radix-4 on even / odd inputs, then radix-2 combine with W8^k
    dft4<T, V>(v[0], v[2], v[4], v[6]);
    dft4<T, V>(v[1], v[3], v[5], v[7]);
    cx<T> t1 = mul_w8_1<T, V>(v[3]), t2 = mul_mi(v[5]), t3 = mul_w8_3<T, V>(v[7]);
    cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + t1; v[5] = e1 - t1;
    v[2] = e2 + t2; v[6] = e2 - t2;
    v[3] = e3 + t3; v[7] = e3 - t3;
}

// a * W16^k for the constant k that appear in the 4x4 decomposition
template <typename T, int K, bool V = false> MFFT_HD cx<T> mul_w16(cx<T> a)
{
    constexpr T c = consts<T>::c16, s = consts<T>::s16;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MFFT_VEC_MATH(T, V) && (K == 1 || K == 3 || K == 9)) {
        const vec2_t<T> A = to_v(a), M = v_mi<T>(A);
        if constexpr (K == 1) return to_c<T>(A * c + M * s);
        else if constexpr (K == 3) return to_c<T>(A * s + M * c);
        else return to_c<T>(-(A * c + M * s));
    }
#endif
    if constexpr (K == 0) return a;
    else if constexpr (K == 1) return {a.x * c + a.y * s, a.y * c - a.x * s};       // (c - i s)
    else if constexpr (K == 2) return mul_w8_1<T, V>(a);
    else if constexpr (K == 3) return {a.x * s + a.y * c, a.y * s - a.x * c};       // (s - i c)
    else if constexpr (K == 4) return mul_mi(a);
    else if constexpr (K == 6) return mul_w8_3<T, V>(a);
    else /* K == 9 */ return {-(a.x * c + a.y * s), a.x * s - a.y * c};             // (-c + i s) = -W16^1
}

template <typename T, bool V = false> MFFT_HD void dft16(cx<T>* v)
{
    // index i = i0 + 4*i1 : radix-4 over i1 for each i0, twiddle W16^(i0*k1), radix-4 over i0
    dft4<T, V>(v[0], v[4], v[8], v[12]);
    dft4<T, V>(v[1], v[5], v[9], v[13]);
    dft4<T, V>(v[2], v[6], v[10], v[14]);
    dft4<T, V>(v[3], v[7], v[11], v[15]);
    // after the first step v[i0 + 4*k1] holds partial k1; apply W16^(i0*k1)
    v[5] = mul_w16<T, 1, V>(v[5]);  v[9] = mul_w16<T, 2, V>(v[9]);   v[13] = mul_w16<T, 3, V>(v[13]);
    v[6] = mul_w16<T, 2, V>(v[6]);  v[10] = mul_w16<T, 4, V>(v[10]); v[14] = mul_w16<T, 6, V>(v[14]);
    v[7] = mul_w16<T, 3, V>(v[7]);  v[11] = mul_w16<T, 6, V>(v[11]); v[15] = mul_w16<T, 9, V>(v[15]);
    // radix-4 over i0 for each k1: inputs v[0..3 + 4*k1], output q = k1 + 4*k0
    dft4<T, V>(v[0], v[1], v[2], v[3]);
    dft4<T, V>(v[4], v[5], v[6], v[7]);
    dft4<T, V>(v[8], v[9], v[10], v[11]);
    dft4<T, V>(v[12], v[13], v[14], v[15]);
    // now v[k0 + 4*k1] holds output q = k1 + 4*k0 : transpose the 4x4 index
    cx<T> t;
    t = v[1];  v[1] = v[4];   v[4] = t;
    t = v[2];  v[2] = v[8];   v[8] = t;
    t = v[3];  v[3] = v[12];  v[12] = t;
    t = v[6];  v[6] = v[9];   v[9] = t;
    t = v[7];  v[7] = v[13];  v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// 32 points: two 16-point transforms of the even and odd inputs, combined with W32^k
template <typename T, bool V = false> MFFT_HD void dft32(cx<T>* v)
{
    constexpr T c[16] = {(T)1.0L, (T)0.98078528040323044912618223613423903697L, (T)0.92387953251128675612818318939678828682L,
                         (T)0.83146961230254523707878837761790575673L, (T)0.70710678118654752440084436210484903928L,
                         (T)0.55557023301960222474283081394853287438L, (T)0.38268343236508977172845998403039886676L,
                         (T)0.19509032201612826784828486847702224093L, (T)0.0L, (T)-0.19509032201612826784828486847702224093L,
                         (T)-0.38268343236508977172845998403039886676L, (T)-0.55557023301960222474283081394853287438L,
                         (T)-0.70710678118654752440084436210484903928L, (T)-0.83146961230254523707878837761790575673L,
                         (T)-0.92387953251128675612818318939678828682L, (T)-0.98078528040323044912618223613423903697L};
    cx<T> e[16] = {}, o[16] = {};
    for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft16<T, V>(e);
    dft16<T, V>(o);
    for (int k = 0; k < 16; ++k) {
        // W32^k = cos(2 pi k/32) - i sin(2 pi k/32), sin(2 pi k/32) = cos(2 pi (8-k)/32)
        const T cs = c[k], sn = k <= 8 ? c[8 - k] : c[k - 8];
        const cx<T> t{o[k].x * cs + o[k].y * sn, o[k].y * cs - o[k].x * sn};
        v[k] = e[k] + t;
        v[k + 16] = e[k] - t;
    }
}

// Radix-R DFT of the R registers v[0], v[S], v[2S], ... (stride S), natural order.
template <typename T, int R, int S, bool V = false> MFFT_HD void dft_strided(cx<T>* v)
{
    if constexpr (R == 1) {
        (void)v;
    } else if constexpr (R == 2) {
        bfly2(v[0], v[S]);
    } else if constexpr (R == 4) {
        dft4<T, V>(v[0], v[S], v[2 * S], v[3 * S]);
    } else {
        cx<T> w[R] = {};
#pragma unroll
        for (int i = 0; i < R; ++i) w[i] = v[i * S];
        if constexpr (R == 8) dft8<T, V>(w); else if constexpr (R == 16) dft16<T, V>(w); else dft32<T, V>(w);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i * S] = w[i];
    }
}

// ---------------------------------------------------------------------------
// Stockham autosort index algebra for one pass of radix R over a length-N
// transform whose already-combined sub-transform length is NS (product of the
// radices of the earlier passes).  Butterfly j in [0, N/R):
//   reads   x[j + i*(N/R)]                       i = 0..R-1
//   twiddle x_i *= exp(-2*pi*i * (j mod NS) * i / (NS*R))
//   writes  y[(j / NS)*NS*R + (j mod NS) + q*NS] q = 0..R-1
// ---------------------------------------------------------------------------
MFFT_HD int stockham_out_base(int j, int ns, int r)
{
    return (j / ns) * ns * r + (j % ns);
}

// LDS padding: one extra element after every 2^SHIFT elements (SHIFT >= 31 disables).
template <int SHIFT> MFFT_HD int lds_pad(int pos)
{
    if constexpr (SHIFT >= 31) return pos;
    else return pos + (pos >> SHIFT);
}

}  // namespace mfft
