/*
 * mfft -- MI355X (gfx950) FFT engine for the spectral hot path: C ABI.
 *
 * The reference has no FFI of its own: its seam is the Python call
 *     F = dt * np.fft.fft(xw, n=Nfft)            final2.py:143
 *                                                streamlit_app_combined.py:204, :213
 * surrounded by the window / power / normalise expressions of
 *     fft_from_timeseries                        final2.py:139-146
 *     fft_extract_E_p                            streamlit_app_combined.py:199-215
 * Each entry point below names the reference expression it replaces.  The Python
 * side (fft_amd/_lib.py) binds these with ctypes; nothing in the signatures is a
 * torch or numpy type -- plain pointers, sizes and scalars only.
 *
 * Conventions: all functions return 0 (MFFT_OK) or a negative MFFT_E_* code and
 * never throw.  mfft_last_error() returns a thread-local detail string.  Plans
 * are bound to the HIP device that was current when they were created.  A plan
 * may be executed from several host threads and on several streams: executions
 * of the SAME plan that share device state of the plan (pass workspaces, the
 * chirp-z convolution buffer, the demodulation table, the staging copies of a
 * windowed / padded transform over a segmented layout) are ordered by the
 * library, on the host and -- through a completion event -- on the device,
 * whichever entry point issues them (mfft_exec on the caller's stream, the host
 * entry points on streams of their own); distinct plans are independent and
 * overlap on the device.  "d_" pointers are device memory, "h_" pointers host
 * memory.
 */
#ifndef MFFT_H
#define MFFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfft_plan_s* mfft_plan;

/* element types: the arithmetic type of the transform */
#define MFFT_C64   0   /* complex64  (float  re, im) -- fp32 butterflies */
#define MFFT_C128  1   /* complex128 (double re, im) -- fp64 butterflies */

/* plan flags */
#define MFFT_HANN     0x01u  /* x[j] *= hann(n_in)[j]      np.hanning(len(f_t)); f_t*win   final2.py:139-140 */
#define MFFT_POWER    0x02u  /* out = |scale*F|^2 (real)   np.abs(dt*F)**2                 final2.py:143,145 */
#define MFFT_DEMOD    0x04u  /* x[j] *= exp(+i*mu*(t0+j*dt))                                streamlit_app_combined.py:209 */
#define MFFT_REAL_IN  0x08u  /* input is real (float/double), promoted like numpy does      BASELINE config 1 */
#define MFFT_STATS    0x10u  /* also return per-transform max(S), sum(S), sum(f_k*S)        final2.py:146, streamlit_app_combined.py:207 */
#define MFFT_INVERSE  0x20u  /* exponent +2*pi*i, unnormalised (used internally by the chirp-z path) */

/* error codes */
#define MFFT_OK              0
#define MFFT_E_INVALID      -1   /* bad argument (the Python layer raises ValueError) */
#define MFFT_E_UNSUPPORTED  -2   /* valid request this build cannot serve */
#define MFFT_E_NOMEM        -3   /* host or device allocation failed */
#define MFFT_E_HIP          -4   /* a HIP runtime call failed; see mfft_last_error() */
#define MFFT_E_NODEVICE     -5   /* no usable gfx950 device */

/* ---- library ------------------------------------------------------------- */

int         mfft_version(void);            /* 10000*major + 100*minor + patch */
int         mfft_device_count(void);       /* number of HIP devices, or MFFT_E_NODEVICE */
int         mfft_device_memory(int64_t* free_bytes, int64_t* total_bytes);   /* of the current device (plan caches size themselves by it) */
int         mfft_release_cached(void);     /* frees what helper entry points keep between calls: the staging block of
                                              mfft_lsq_normal_host (per device) and the tables of mfft_twiddle_tile */
const char* mfft_strerror(int code);
const char* mfft_last_error(void);

/* ---- 1-D plans -------------------------------------------------------------
 * Replaces np.fft.fft(x, n=n, axis=-1) on a (batch, n_in) array, fused with the
 * expressions selected by `flags`:
 *
 *   for each of `batch` signals x[0..n_in):
 *     x[j] *= hann(n_in)[j]                      if MFFT_HANN
 *     x[j] *= exp(+i*mu*(t0 + j*dt))             if MFFT_DEMOD
 *     X[k]  = sum_{j<n_in} x[j]*exp(-2*pi*i*j*k/n),  k < n     (zero-padded, n_in <= n)
 *     out[k] = scale*X[k]                        (complex)      dt*np.fft.fft(...)
 *        or   = (scale*Re X)^2 + (scale*Im X)^2  (real)         if MFFT_POWER
 *
 * n may be any positive length: powers of two run as Stockham radix-16/8/4/2
 * passes (one LDS-resident kernel up to 16384 points c64 / 8192 c128, a
 * multi-kernel four-step decomposition above that); other lengths run through a
 * chirp-z (Bluestein) wrapper around the power-of-two kernels.
 */
int mfft_plan_1d(mfft_plan* out, int64_t n, int64_t n_in, int64_t batch, int dtype, uint32_t flags);

/* The same plan over strided / segmented device buffers (mfft_exec only).  A layout
 * says where element j of signal b lives:
 *     b*batch_stride + (seg ? (j/seg)*seg_stride + j%seg : j)        (in elements)
 * batch_stride 0 selects the dense default (n_in for input, n for output); seg 0
 * means the signal is contiguous.  No counterpart in the reference (it never leaves
 * one process): this is how the multi-GPU decompositions -- np.fft.fft2-style pencils
 * and the four-step split of one long np.fft.fft (BASELINE configs 4 and 5) -- pack
 * and unpack their all-to-all exchanges inside the transform's own load and store. */
typedef struct mfft_layout {
    int64_t batch_stride;
    int64_t seg;
    int64_t seg_stride;
} mfft_layout;
int mfft_plan_1d_layout(mfft_plan* out, int64_t n, int64_t n_in, int64_t batch, int dtype, uint32_t flags,
                        const mfft_layout* in_layout, const mfft_layout* out_layout);

/* Column-batched plan: np.fft.fft(a, axis=-2) on a (batch, n, inner) array whose last axis is
 * contiguous -- every butterfly is shared by `inner` adjacent columns, so the transform runs
 * on the strided columns where they lie (coalesced tiles of 256-byte rows) and no transpose
 * pass exists.  This is the column step of np.fft.fft2 (BASELINE config 4) and the strided
 * steps of the four-step split of one long np.fft.fft (config 5).  n: a power of two >= 16;
 * inner: a power of two >= 32 (complex64) / 16 (complex128); other shapes return
 * MFFT_E_UNSUPPORTED (callers fall back to mfft_transpose + mfft_plan_1d).  Plain complex in
 * and out, executed with mfft_exec (scale must be 1) or mfft_exec_host.
 * twiddle_n > 0 (a power of two, at most 2^40) additionally multiplies output element (k, c) by
 * exp(-2*pi*i * k*(twiddle_c0 + c) / twiddle_n): the inter-step twiddle of the four-step
 * algorithm, folded into the last pass' store. */
int mfft_plan_1d_columns(mfft_plan* out, int64_t n, int64_t inner, int64_t batch, int dtype,
                         int64_t twiddle_n, int64_t twiddle_c0);
/* The same with the rows of the RESULT out_ld >= inner elements apart (0 or inner: dense): the transform
 * of a block of `inner` columns written straight into its place in a wider (n x out_ld) array, as
 * np.fft.fft(a[:, c0:c0+inner], axis=0) assigned to out[:, c0:c0+inner] -- the column groups of the
 * 2-D decomposition.  mfft_exec only; n * out_ld <= 2^32; excludes the fused twiddle. */
int mfft_plan_1d_columns_ld(mfft_plan* out, int64_t n, int64_t inner, int64_t batch, int dtype,
                            int64_t twiddle_n, int64_t twiddle_c0, int64_t out_ld);
int mfft_plan_destroy(mfft_plan plan);

/* Bytes of device workspace the plan holds, and the number of kernel launches
 * one execution issues (for reporting). */
int mfft_plan_workspace_bytes(mfft_plan plan, int64_t* bytes);
int mfft_plan_launches(mfft_plan plan, int* launches);

/* Execute on device buffers.  d_in: batch*n_in elements (complex, or real if
 * MFFT_REAL_IN), contiguous per signal.  d_out: batch*n complex, or batch*n real
 * if MFFT_POWER.  d_stats: NULL, or (with MFFT_STATS) 3*batch doubles receiving
 * per signal {max_k S[k], sum_k S[k], sum_k f_k*S[k]} where S is the power
 * spectrum and f_k = k for k < ceil(n/2), k-n after (np.fft.fftfreq order, in
 * bins).  mu, t0, dt are read only with MFFT_DEMOD.  stream: a hipStream_t, or
 * NULL for the default stream.  The call is asynchronous.  d_out may be d_in for
 * a dense complex -> complex transform of full-length input (n_in == n, no
 * MFFT_REAL_IN, no MFFT_POWER); any other aliasing of the two is undefined.
 * Stream capture: when `stream` is recording into a graph (hipStreamBeginCapture, torch.cuda.graph) the
 * plan's kernels are recorded like any others -- nothing is allocated and no event from outside the capture
 * is waited for; a plan that allocates a buffer on first use (a second pass workspace, staging copies)
 * returns MFFT_E_UNSUPPORTED until it has been executed once outside a capture.  Replays of such a graph are
 * ordered against other executions of the same plan by the caller, not by the library. */
int mfft_exec(mfft_plan plan, const void* d_in, void* d_out, double scale,
              double mu, double t0, double dt, double* d_stats, void* stream);

/* Same transform on host buffers (numpy arrays): copies in, runs, copies out and
 * synchronises.  h_stats: NULL or 3*batch doubles.  With MFFT_STATS, h_out may be
 * NULL: only the statistics come back (the centroid pass of
 * streamlit_app_combined.py:204-207 needs sum(omega*S)/sum(S), not S itself). */
int mfft_exec_host(mfft_plan plan, const void* h_in, void* h_out, double scale,
                   double mu, double t0, double dt, double* h_stats);

/* ---- peak candidates ------------------------------------------------------------
 * Interior local maxima of a real spectrum S[0..n): every i in [1, n-2] with
 *     S[i-1] < S[i]  and  S[i] >= S[i+1]           (left strict, right weak)
 * i.e. np.where((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:]))[0] + 1   -- final2.py:151,
 * streamlit_app_combined.py:218.  Indices are written to d_idx in no particular order
 * (at most `capacity` of them); *d_count receives the TOTAL number found, which may
 * exceed capacity (the caller then rescans with a larger buffer).  The call zeroes
 * *d_count itself, on `stream`.  dtype selects float (MFFT_C64) or double (MFFT_C128)
 * spectra. */
int mfft_local_maxima(const void* d_S, int64_t n, int dtype, int64_t* d_idx, int64_t capacity,
                      int64_t* d_count, void* stream);

/* mfft_exec_host for a one-signal MFFT_POWER plan, followed by mfft_local_maxima on the
 * device copy of the spectrum: h_out receives S (n reals), h_idx up to `capacity` indices
 * (unordered), *h_count the total found.  Saves the three host passes over S that the
 * reference's peak scan makes. */
int mfft_exec_host_peaks(mfft_plan plan, const void* h_in, void* h_out, double scale, double mu,
                         double t0, double dt, double* h_stats, int64_t* h_idx, int64_t capacity,
                         int64_t* h_count);

/* ---- spectral lines of a batch of signals, one call ------------------------------------
 * What both estimators do between the power spectrum and the k x k solve, for every signal of
 * the plan's batch, on the device copy of the spectra:
 *   candidates   interior local maxima of S (rule above); mode 0 adds bin 0 when include_dc and
 *                S[0] > 0, otherwise drops |omega| <= dc_guard                       final2.py:151-156
 *   pick         walk the candidates by decreasing S, keep one when it is farther than `guard`
 *                from every line kept so far, stop at `want` lines                   final2.py:161-169
 *                (mode 1 measures the distance between |omega| values:  streamlit_app_combined.py:221-227)
 *   refine       three-point log-parabola around each kept bin                       final2.py:102-113, :171
 *   fit          (h_gram != NULL, mode 0) normal equations of the windowed least squares with
 *                columns E = -omega_hat, laid out per signal as mfft_lsq_normal_host  final2.py:174-175
 * omega_k = 2*pi * (f_k * (1 / (n*dt))), f_k the signed FFT-order bin: numpy's own doubles.  Equal
 * heights are taken in order of decreasing bin index.
 * plan: MFFT_POWER | MFFT_STATS (complex128 without MFFT_REAL_IN / MFFT_DEMOD when h_gram is asked
 * for).  h_x: batch*n_in samples; scale, mu, t0, dt as mfft_exec_host; h_times: n_in doubles (fit
 * only).  Outputs: h_S batch*n reals (may be NULL; with `normalize` each spectrum is divided on the
 * device by its maximum, or by 1 when that is not positive: final2.py:146), h_stats 3*batch (of the
 * unnormalised spectra), h_idx / h_omega_hat / h_height
 * batch*want in pick order (idx -1 beyond h_count[b]; h_height may be NULL), h_gram batch *
 * (2*want*want + 2*want + 1) doubles (columns beyond h_count[b] carry E = 0 and are to be ignored).
 * Synchronous; one upload and one download per array.  The device work runs on a stream of the plan, so
 * calls on different plans from different host threads (one Streamlit session each) overlap. */
typedef struct mfft_peak_params {
    double dt;          /* sample spacing of the time series (sets the omega axis) */
    double dc_guard;    /* mode 0, include_dc == 0: candidates need |omega| > dc_guard */
    double guard;       /* minimum spacing of kept lines */
    int include_dc;
    int mode;           /* 0 = final2.py, 1 = streamlit_app_combined.py */
    int want;           /* lines to keep per signal, 1..64 */
} mfft_peak_params;
int mfft_lines_host(mfft_plan plan, const void* h_x, double scale, double mu, double t0, double dt,
                    const double* h_times, const mfft_peak_params* prm, int normalize, void* h_S, double* h_stats,
                    int64_t* h_idx, double* h_omega_hat, double* h_height, int32_t* h_count, double* h_gram);

/* The same with the samples GENERATED on the device instead of uploaded: signal b of the batch is the return
 * amplitude f_b(t_j) of a small Hermitian block (formula and arguments as mfft_return_amplitude_host below;
 * h_w batch*nk eigenvalues, h_c0 / h_vrow batch*nk complex, the same nk <= 32 for every block), written
 * straight into the transform's input.  This is the body of the loop over blocks m,
 *     f_t = return_amplitude_block_general(K, J, m, site, times); fft_from_timeseries(f_t, times, ...)
 * (final2.py:752-758, :830-835, :853-859), as one call in which only the time grid goes up and the results
 * come down.  plan: complex128, MFFT_POWER | MFFT_STATS, no MFFT_REAL_IN / MFFT_DEMOD; h_times: the n_in
 * sample times (always needed).  Everything else as mfft_lines_host. */
int mfft_lines_blocks_host(mfft_plan plan, const double* h_w, const double* h_c0, const double* h_vrow, int nk,
                           double hbar, double scale, const double* h_times, const mfft_peak_params* prm,
                           int normalize, void* h_S, double* h_stats, int64_t* h_idx, double* h_omega_hat,
                           double* h_height, int32_t* h_count, double* h_gram);

/* ---- weights of the spectral lines ----------------------------------------------
 * Normal equations of the windowed least-squares fit both estimators end with
 * (final2.py:174-175: lstsq(exp(-1j*outer(times, E)) * win[:, None], f_t*win);
 * streamlit_app_combined.py:233-246: the same inside the 2^k sign search, against the
 * demodulated xw).  With columns b_a[j] = win[j]*exp(-i*E[a]*times[j]), a < m <= 64, and
 * target y[j] = x[j]*win[j]*exp(+i*mu*times[j]):
 *     G[a][b] = sum_j conj(b_a[j])*b_b[j]     h_G: 2*m*m doubles, row-major, re/im interleaved
 *     r[a]    = sum_j conj(b_a[j])*y[j]       h_r: 2*m doubles
 *     yy      = sum_j |y[j]|^2                *h_yy
 * so that the least-squares coefficients solve G c = r and the residual norm is
 * sqrt(yy - 2 Re(c^H r) + c^H G c).  The sign search evaluates every +-|omega| column once
 * (m = 2k) and reads each of its 2^k fits off sub-blocks of G.  h_x: n complex128 samples,
 * h_win: n doubles or NULL (all ones), h_times: n doubles.  Synchronous. */
int mfft_lsq_normal_host(const void* h_x, const double* h_win, const double* h_times, int64_t n,
                         double mu, const double* h_E, int m, double* h_G, double* h_r, double* h_yy);

/* The same normal equations for `batch` signals in one call, on DEVICE arrays, by an entry
 * point that allocates nothing.  No counterpart in the reference: it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation -- and for the batched estimator, which keeps every block's f(t) on the
 * device.  d_x: batch*n complex128 samples; d_win: n doubles or NULL; d_times: n doubles (shared
 * by all signals); d_mu: batch doubles or NULL (no demodulation); d_E: batch*m line positions;
 * d_out: batch * (2*m*m + 2*m + 1) doubles, per signal {G, r, yy} laid out as above.  Every value
 * of d_out is written (it need not be cleared).  Asynchronous on `stream`.
 * The per-block partial sums live in a fixed 2 MiB device array inside the library (one per GPU,
 * held until the process ends), so a call is refused with MFFT_E_UNSUPPORTED when
 * batch * m * ceil(m/16) * 36 doubles exceed it (m = 6: 1213 signals; m = 64: 28); split larger
 * batches (at most 65535 / ceil(m/16) signals per call in any case).  All calls of THIS entry point on
 * one GPU share that array: two of them that overlap on one GPU (two streams, two threads) race --
 * serialise them.  Nothing else in the library touches the array (mfft_lines_host's fit keeps its
 * partial sums in a block of its plan), so a call may overlap any other entry point.  The first call on
 * each device resolves the array's address and should be made outside graph capture.  Results are deterministic (fixed summation
 * order, no atomics) and agree with mfft_lsq_normal_host to rounding (1e-13 relative). */
int mfft_lsq_normal_static(const void* d_x, const double* d_win, const double* d_times, int64_t n,
                           int64_t batch, const double* d_mu, const double* d_E, int m, double* d_out,
                           void* stream);

/* ---- input generator -----------------------------------------------------------
 * Return amplitude of a small Hermitian block on a time grid,
 *     f[j] = sum_k vrow[k] * ( exp(-i * w[k] * t_j / hbar) * c0[k] ),   k < nk <= 32
 * which is what evolve_state_in_H + kets[:, site] compute one time step at a time in a
 * Python loop (final2.py:85-97; streamlit_app_combined.py:86-93, :166-169): w are the
 * eigenvalues, c0 = V^H psi0 and vrow = V[site, :].  t_j = h_times[j] when h_times is not
 * NULL, else t0 + j*dt.  h_* are host arrays (c0, vrow: interleaved re, im doubles); h_out
 * receives n complex128 values.  Evaluated in double precision on the device. */
int mfft_return_amplitude_host(void* h_out, int64_t n, const double* h_times, double t0, double dt,
                               const double* h_w, const double* h_c0, const double* h_vrow, int nk,
                               double hbar);
/* Same, writing n complex128 values to device memory (uniform grid only), asynchronous. */
int mfft_return_amplitude(void* d_out, int64_t n, double t0, double dt, const double* h_w,
                          const double* h_c0, const double* h_vrow, int nk, double hbar, void* stream);

/* ---- element-wise helpers used around the transform ------------------------ */

/* out[k] = in[k] / (m > 0 ? m : 1) with m = *d_max  -- S / (S.max() if S.max() > 0 else 1.0), final2.py:146.
 * dtype selects float (MFFT_C64) or double (MFFT_C128) real arrays. */
int mfft_normalize_by_max(const void* d_in, void* d_out, int64_t count, const double* d_max,
                          int dtype, void* stream);

/* out[b][c][r] = in[b][r][c]  for `batch` matrices of rows x cols complex elements
 * (local re-layout step of the pencil / four-step decompositions). */
/* fft_extract_E_p's device passes (streamlit_app_combined.py:190-231) on samples that never leave the device: the return
 * amplitude of ONE small Hermitian block is generated into the plan's input (as mfft_lines_blocks_host does), a rough
 * pass through `rough_plan` -- the same one-signal plan without MFFT_DEMOD -- gives the centroid
 *     mu = (2 pi / (n dt)) * sum_k f_k S0[k] / sum_k S0[k]      (0 when the sum is not positive;  :204-207)
 * and the main pass through `plan` (with MFFT_DEMOD) transforms the samples times exp(+i mu (t0 + j dt)) (:209-215) and
 * runs the line search on its spectrum.  rough_plan NULL: no demodulation (plan without MFFT_DEMOD, *h_mu = 0).  h_S receives
 * the unnormalised spectrum; the other outputs are mfft_lines_host's.  The samples and the time grid stay on the device
 * for mfft_lines_gram_host. */
int mfft_lines_extract_host(mfft_plan plan, mfft_plan rough_plan, const double* h_w, const double* h_c0, const double* h_vrow,
                            int nk, double hbar, double scale, double t0, double dt, const double* h_times,
                            const mfft_peak_params* prm, void* h_S, double* h_stats, int64_t* h_idx, double* h_omega_hat,
                            double* h_height, int32_t* h_count, double* h_mu);
/* Normal equations of the windowed line fit (streamlit_app_combined.py:233-246; mfft_lsq_normal_host's G, r, yy) for m
 * columns exp(-i E_a t) against the samples the last mfft_lines_extract_host / mfft_lines_blocks_host call of this
 * one-signal plan left on the device, demodulated by mu.  h_gram: 2 m m + 2 m + 1 doubles (G row-major re/im, r re/im, yy).
 * MFFT_E_INVALID when the plan holds no such samples. */
int mfft_lines_gram_host(mfft_plan plan, const double* h_E, int m, double mu, double* h_gram);

/* Page-locked host memory the device reads and writes directly.  No counterpart in the reference (its arrays are
 * numpy's): the host layer hands the spectra of fft_from_timeseries / fft_extract_E_p (final2.py:146, 188) back in
 * such blocks, so that mfft_lines_host downloads them in place instead of into a staging buffer and copying again.
 * Any host pointer works everywhere; a pinned one is merely faster. */
int mfft_host_alloc(void** out, int64_t bytes);
int mfft_host_free(void* p);

int mfft_transpose(const void* d_in, void* d_out, int64_t rows, int64_t cols, int64_t batch,
                   int dtype, void* stream);

/* out[j][i][:] = in[i][j][:]  for an (a, b, w) array of complex elements with the last axis
 * contiguous: the pack / unpack of an all-to-all exchange whose per-peer blocks are rows of w
 * elements (np.swapaxes(x.reshape(a, b, w), 0, 1) made contiguous). */
int mfft_swap_axes(const void* d_in, void* d_out, int64_t a, int64_t b, int64_t w, int dtype,
                   void* stream);
/* out[j][i][:] = in[i*in_stride_a + j*in_stride_b + (0..w)], i < a, j < b: the same block copy reading
 * rows that are not adjacent (in_stride_a = b*w, in_stride_b = w is mfft_swap_axes) -- one column group
 * of an exchange packed out of, or unpacked into, rows of the full width. */
int mfft_swap_axes_strided(const void* d_in, void* d_out, int64_t a, int64_t b, int64_t w,
                           int64_t in_stride_a, int64_t in_stride_b, int dtype, void* stream);

/* x[r][c] *= exp(-2*pi*i * (r0 + r)*(c0 + c) / n_total)  on a rows x cols complex tile
 * (the inter-step twiddle of the four-step algorithm). */
int mfft_twiddle_tile(void* d_x, int64_t rows, int64_t cols, int64_t r0, int64_t c0,
                      int64_t n_total, int dtype, void* stream);

/* ---- multi-GPU decompositions -----------------------------------------------------------
 * No counterpart in the reference (it never leaves one process): BASELINE configurations 4 and 5,
 * np.fft.fft2 of an n0 x n1 array held in row slabs and np.fft.fft of ONE signal of n1*n2 points
 * held in contiguous chunks, one rank per GPU.  Exchanges are groups of RCCL point-to-point
 * transfers over xGMI, cut into column groups and row chunks and overlapped with the transforms that
 * produce or consume them; packing and unpacking happen inside the transform kernels' own loads and stores.
 * RCCL (librccl.so.1) is loaded on first use; single-GPU callers never load it.
 *
 * Communicators.  mfft_comm_unique_id: rank 0 makes a 128-byte id and hands it to every rank by
 * any means (a file, MPI, torch.distributed); mfft_comm_init_rank: every rank, with its device
 * current, joins (ncclCommInitRank; collective).  mfft_comm_init_local: `nranks` ranks that share
 * the CURRENT device inside one process, exchanging by device copies -- a rehearsal of the
 * multi-rank paths on one GPU; each rank must then be driven by a host thread of its own, because
 * an exchange waits for all ranks to arrive. */
typedef struct mfft_comm_s* mfft_comm;
typedef struct mfft_dist_plan_s* mfft_dist_plan;
int mfft_comm_unique_id(void* id128);
int mfft_comm_init_rank(mfft_comm* out, int world, int rank, const void* id128);
int mfft_comm_init_local(mfft_comm* out /* nranks handles */, int nranks);
int mfft_comm_info(mfft_comm comm, int* world, int* rank);
int mfft_comm_destroy(mfft_comm comm);
/* The exchange primitive itself: block d_send[p] (`bytes` bytes, device memory) goes to rank p and
 * d_recv[p] receives what rank p sends here -- an all-to-all with caller-chosen block addresses,
 * one ncclSend / ncclRecv pair per peer in one group.  Enqueued on `stream`; collective. */
int mfft_comm_exchange(mfft_comm comm, const void* const* d_send, void* const* d_recv, int64_t bytes, void* stream);

/* 2-D forward DFT of an n0 x n1 array distributed by row slabs.  Input: this rank's n0/P
 * consecutive rows (row-major, (n0/P)*n1 elements).  Output, same size:
 *   restore_layout == 0 (one exchange):  all n0 rows of this rank's n1/P columns,
 *                                        out[k0*(n1/P) + c] = F[k0][rank*(n1/P) + c];
 *   restore_layout != 0 (two exchanges): the input's own layout, out[i*n1 + k1] = F[rank*(n0/P) + i][k1].
 * Both dimensions must divide by the number of ranks; shapes the column-batched passes do not
 * serve return MFFT_E_UNSUPPORTED.
 * Schedule.  col_groups (<= 0: 2): this rank's n1/P columns are cut into that many groups (fewer when a group
 * would be narrower than 64 complex128 / 128 complex64 columns); the groups are exchanged one after the
 * other, and a group that has arrived is transformed along the strided axis -- and, with restore_layout,
 * sent back -- while the next one travels.  chunks (<= 0: 4): row chunks of the group at either end of that
 * pipeline, so that the row transforms overlap the first group's exchange and the unpacking the last
 * group's.  Both are 1 with one rank; col_groups = 1 is the schedule that chunks by rows only. */
int mfft_plan_2d_dist(mfft_dist_plan* out, mfft_comm comm, int64_t n0, int64_t n1, int dtype,
                      int restore_layout, int chunks, int col_groups);

/* Forward DFT of ONE signal of n = n1*n2 points (four-step split) distributed in contiguous
 * chunks.  Input: this rank's n/P consecutive samples.  Output, n/P elements:
 *   natural_order != 0 (three exchanges): X[rank*n/P + m];
 *   natural_order == 0 (two exchanges):   out[i*n2 + k2] = X[(rank*(n1/P) + i) + n1*k2].
 * chunks, col_groups: as for mfft_plan_2d_dist, over this rank's n2/P columns of the n1 x n2 view: the
 * first exchange, the n1-point transforms and the second exchange form a pipeline over the column groups,
 * the n2-point transforms (and the natural-order exchange) one over the row chunks of the last group. */
int mfft_plan_1d_dist(mfft_dist_plan* out, mfft_comm comm, int64_t n1, int64_t n2, int dtype,
                      int natural_order, int chunks, int col_groups);

/* Run a distributed plan on device buffers (d_in != d_out; d_in is not modified), kernels on
 * `stream`, exchanges on a stream of the plan ordered against it by events.  Asynchronous;
 * collective: every rank of the communicator calls it with its own plan, in the same order. */
int mfft_dist_exec(mfft_dist_plan plan, const void* d_in, void* d_out, void* stream);
int mfft_dist_plan_info(mfft_dist_plan plan, int* chunks, int* col_groups, int64_t* workspace_bytes);
int mfft_dist_plan_destroy(mfft_dist_plan plan);

#ifdef __cplusplus
}
#endif
#endif /* MFFT_H */
