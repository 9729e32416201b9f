/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the spectral hot path.
 *
 * Nothing under fft_amd/ (the product) may call, link or import this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker.
 *
 * What it restates.  The reference computes every transform with
 *     np.fft.fft(xw, n=Nfft)          /root/reference/final2.py:143
 *                                     /root/reference/streamlit_app_combined.py:204, :213
 * whose arithmetic lives in numpy's bundled pocketfft (numpy.fft._pocketfft_umath;
 * the reference pins no version, the container has numpy 2.2.6).  pocketfft's
 * published algorithm is: factor n; if every prime factor is small run a
 * mixed-radix Cooley-Tukey transform with twiddles exp(-2*pi*i*k/n) computed
 * once per length to full precision; otherwise (a large prime factor) run
 * Bluestein's chirp-z algorithm on a padded smooth length.  Input shorter than
 * n is zero-padded, the transform is forward (exponent -2*pi*i*j*k/n) and
 * unnormalised.  This file restates exactly that in plain C (double precision,
 * long-double twiddle generation), plus an O(n^2) long-double direct DFT used to
 * pin both this restatement and numpy at small n.
 *
 * Parity status: the reference has no tests or golden vectors of its own, so the
 * oracle is pinned by fixtures generated from the reference's functions in the
 * build container (tests/golden/make_golden.py) and by the direct DFT below.
 *
 * Build: gcc -O2 -fPIC -shared -o liboracle.so pocket_dft.c -lm   (see Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } cpx;

static const long double TWO_PI_L = 6.283185307179586476925286766559005768L;

/* exp(-2*pi*i*k/n).  The angle is reduced to [-pi/4, pi/4] around the nearest
 * quarter turn in exact integer arithmetic, so the rounding error does not grow
 * with k and the quarter-turn values are exact. */
static cpx unit_root(int64_t k, int64_t n)
{
    k %= n; if (k < 0) k += n;
    int64_t q = (4 * k + n / 2) / n;        /* nearest quarter turn, 0..4 */
    int64_t rem4 = 4 * k - q * n;           /* theta = q*pi/2 + 2*pi*rem4/(4n) */
    long double a = TWO_PI_L * (long double)rem4 / (4.0L * (long double)n);
    long double ca = cosl(a), sa = sinl(a), c, s;
    switch (q & 3) {
    case 0:  c = ca;  s = sa;  break;
    case 1:  c = -sa; s = ca;  break;
    case 2:  c = -ca; s = -sa; break;
    default: c = sa;  s = -ca; break;
    }
    cpx w = { (double)c, (double)(-s) };    /* forward transform: exp(-i*theta) */
    return w;
}

static inline cpx cmul(cpx a, cpx b)
{
    cpx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}

/* ---- mixed-radix decimation-in-time, recursive, out of place ------------- */

typedef struct {
    int64_t n;       /* table length */
    cpx *w;          /* w[k] = exp(-2*pi*i*k/n) */
    cpx *scratch;    /* max radix entries */
} plan_t;

static int64_t smallest_factor(int64_t n)
{
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    for (int64_t p = 3; p * p <= n; p += 2)
        if (n % p == 0) return p;
    return n;
}

static int64_t largest_prime_factor(int64_t n)
{
    int64_t best = 1;
    while (n % 2 == 0) { best = 2; n /= 2; }
    for (int64_t p = 3; p * p <= n; p += 2)
        while (n % p == 0) { best = p; n /= p; }
    if (n > 1) best = n;
    return best;
}

static void mixed_radix(const plan_t *pl, int64_t n, const cpx *in, int64_t stride, cpx *out)
{
    if (n == 1) { out[0] = in[0]; return; }
    const int64_t p = smallest_factor(n), m = n / p;
    for (int64_t r = 0; r < p; ++r)
        mixed_radix(pl, m, in + r * stride, stride * p, out + r * m);
    const int64_t tw_step = pl->n / n;      /* W_n^j == w[j * tw_step] */
    const int64_t rt_step = pl->n / p;      /* W_p^j == w[j * rt_step] */
    cpx *t = pl->scratch;
    for (int64_t k = 0; k < m; ++k) {
        for (int64_t r = 0; r < p; ++r)
            t[r] = cmul(out[r * m + k], pl->w[(r * k) * tw_step]);
        if (p == 2) {
            cpx a = t[0], b = t[1];
            out[k].re = a.re + b.re; out[k].im = a.im + b.im;
            out[k + m].re = a.re - b.re; out[k + m].im = a.im - b.im;
        } else if (p == 4) {
            cpx a = { t[0].re + t[2].re, t[0].im + t[2].im };
            cpx b = { t[0].re - t[2].re, t[0].im - t[2].im };
            cpx c = { t[1].re + t[3].re, t[1].im + t[3].im };
            cpx d = { t[1].re - t[3].re, t[1].im - t[3].im };
            /* forward: multiply d by -i */
            out[k].re = a.re + c.re;         out[k].im = a.im + c.im;
            out[k + m].re = b.re + d.im;     out[k + m].im = b.im - d.re;
            out[k + 2 * m].re = a.re - c.re; out[k + 2 * m].im = a.im - c.im;
            out[k + 3 * m].re = b.re - d.im; out[k + 3 * m].im = b.im + d.re;
        } else {
            for (int64_t q = 0; q < p; ++q) {
                double sr = t[0].re, si = t[0].im;
                for (int64_t r = 1; r < p; ++r) {
                    cpx w = pl->w[((r * q) % p) * rt_step];
                    sr += t[r].re * w.re - t[r].im * w.im;
                    si += t[r].re * w.im + t[r].im * w.re;
                }
                out[k + q * m].re = sr; out[k + q * m].im = si;
            }
        }
    }
}

static int plan_init(plan_t *pl, int64_t n)
{
    pl->n = n;
    pl->w = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    int64_t lpf = largest_prime_factor(n);
    pl->scratch = (cpx *)malloc(sizeof(cpx) * (size_t)(lpf < 4 ? 4 : lpf));
    if (!pl->w || !pl->scratch) return -1;
    for (int64_t k = 0; k < n; ++k) pl->w[k] = unit_root(k, n);
    return 0;
}

static void plan_free(plan_t *pl) { free(pl->w); free(pl->scratch); }

/* smooth transform of exactly n points (n's prime factors all "small") */
static int smooth_fft(const cpx *in, cpx *out, int64_t n)
{
    plan_t pl;
    if (plan_init(&pl, n)) return -1;
    mixed_radix(&pl, n, in, 1, out);
    plan_free(&pl);
    return 0;
}

/* ---- Bluestein chirp-z for lengths with a large prime factor ------------- */

static int bluestein_fft(const cpx *in, cpx *out, int64_t n)
{
    int64_t m = 1;
    while (m < 2 * n - 1) m *= 2;
    cpx *chirp = (cpx *)malloc(sizeof(cpx) * (size_t)n);      /* exp(-i*pi*j^2/n) */
    cpx *a = (cpx *)calloc((size_t)m, sizeof(cpx));
    cpx *b = (cpx *)calloc((size_t)m, sizeof(cpx));
    cpx *A = (cpx *)malloc(sizeof(cpx) * (size_t)m);
    cpx *B = (cpx *)malloc(sizeof(cpx) * (size_t)m);
    if (!chirp || !a || !b || !A || !B) return -1;
    for (int64_t j = 0; j < n; ++j) {
        /* j^2 mod 2n, exactly */
        int64_t jj = (int64_t)(((__int128)j * j) % (2 * n));
        chirp[j] = unit_root(jj, 2 * n);
    }
    for (int64_t j = 0; j < n; ++j) a[j] = cmul(in[j], chirp[j]);
    b[0].re = chirp[0].re; b[0].im = -chirp[0].im;
    for (int64_t j = 1; j < n; ++j) {
        cpx c = { chirp[j].re, -chirp[j].im };
        b[j] = c; b[m - j] = c;
    }
    plan_t pl;
    if (plan_init(&pl, m)) return -1;
    mixed_radix(&pl, m, a, 1, A);
    mixed_radix(&pl, m, b, 1, B);
    for (int64_t k = 0; k < m; ++k) {
        cpx c = cmul(A[k], B[k]);
        a[k].re = c.re; a[k].im = -c.im;         /* conj -> inverse via forward */
    }
    mixed_radix(&pl, m, a, 1, A);
    const double inv = 1.0 / (double)m;
    for (int64_t k = 0; k < n; ++k) {
        cpx c = { A[k].re * inv, -A[k].im * inv };
        out[k] = cmul(c, chirp[k]);
    }
    plan_free(&pl);
    free(chirp); free(a); free(b); free(A); free(B);
    return 0;
}

/* ---- exported entry points ------------------------------------------------ */

/* Forward DFT of n points of the n_in-point input zero-padded (or truncated) to n,
 * as np.fft.fft(x, n=n) defines it.  in/out are interleaved (re, im) doubles. */
int odft_forward_c128(const double *in, int64_t n_in, double *out, int64_t n)
{
    if (n < 1 || n_in < 0) return -2;
    cpx *x = (cpx *)calloc((size_t)n, sizeof(cpx));
    if (!x) return -1;
    int64_t ncopy = n_in < n ? n_in : n;
    memcpy(x, in, sizeof(cpx) * (size_t)ncopy);
    int rc;
    if (largest_prime_factor(n) <= 256) rc = smooth_fft(x, (cpx *)out, n);
    else rc = bluestein_fft(x, (cpx *)out, n);
    free(x);
    return rc;
}

/* The same transform forced through the Bluestein branch (any n), for tests. */
int odft_forward_c128_bluestein(const double *in, int64_t n_in, double *out, int64_t n)
{
    if (n < 1 || n_in < 0) return -2;
    cpx *x = (cpx *)calloc((size_t)n, sizeof(cpx));
    if (!x) return -1;
    int64_t ncopy = n_in < n ? n_in : n;
    memcpy(x, in, sizeof(cpx) * (size_t)ncopy);
    int rc = bluestein_fft(x, (cpx *)out, n);
    free(x);
    return rc;
}

/* O(n * n_in) DFT by definition, long-double accumulation, exact (j*k mod n)
 * index reduction.  The known-answer anchor at small n. */
int odft_direct_ld(const double *in, int64_t n_in, double *out, int64_t n)
{
    if (n < 1 || n_in < 0) return -2;
    int64_t nn = n_in < n ? n_in : n;
    long double *c = (long double *)malloc(sizeof(long double) * (size_t)n);
    long double *s = (long double *)malloc(sizeof(long double) * (size_t)n);
    if (!c || !s) return -1;
    for (int64_t k = 0; k < n; ++k) {
        long double a = TWO_PI_L * (long double)k / (long double)n;
        c[k] = cosl(a); s[k] = -sinl(a);
    }
    for (int64_t k = 0; k < n; ++k) {
        long double sr = 0, si = 0;
        for (int64_t j = 0; j < nn; ++j) {
            int64_t idx = (int64_t)(((__int128)j * k) % n);
            long double xr = in[2 * j], xi = in[2 * j + 1];
            sr += xr * c[idx] - xi * s[idx];
            si += xr * s[idx] + xi * c[idx];
        }
        out[2 * k] = (double)sr; out[2 * k + 1] = (double)si;
    }
    free(c); free(s);
    return 0;
}

/* |F|^2 scaled: S[k] = (scale*re)^2 + (scale*im)^2  (final2.py:143,145) */
void opower_c128(const double *F, double scale, double *S, int64_t n)
{
    for (int64_t k = 0; k < n; ++k) {
        double r = scale * F[2 * k], i = scale * F[2 * k + 1];
        S[k] = r * r + i * i;
    }
}
