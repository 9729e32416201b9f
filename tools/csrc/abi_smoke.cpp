// Host-side exerciser of the C ABI for sanitizer builds (AddressSanitizer on host code only):
//   make -C fft_amd/csrc VARIANT=asan KNOBS="-Xarch_host -fsanitize=address -g"
//   g++ -std=c++17 -fsanitize=address -g -D__HIP_PLATFORM_AMD__ tools/csrc/abi_smoke.cpp -Iinclude -I/opt/rocm/include \
//       -Lfft_amd/lib -lmfft_asan -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,$PWD/fft_amd/lib -o tools/bin/abi_smoke
//   ASAN_OPTIONS=detect_leaks=0 tools/bin/abi_smoke        (on the GPU box)
// Walks every plan kind, the host entry points (small and pipelined transfers), the helper entry
// points and the error paths; checks results loosely (parity is the Python tests' job).
#include <complex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "mfft.h"

#define REQUIRE(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s  (%s)\n", __FILE__, __LINE__, #cond, mfft_last_error()); fflush(stdout); return 1; } } while (0)

using cd = std::complex<double>;
using cf = std::complex<float>;

static double tone_bin_error(const std::vector<double>& S, int64_t n, int64_t k)
{
    int64_t arg = 0;
    for (int64_t i = 1; i < n; ++i) if (S[(size_t)i] > S[(size_t)arg]) arg = i;
    return std::fabs((double)(arg - k));
}

int main()
{
    REQUIRE(mfft_device_count() >= 1);
    // ---- plan kinds through mfft_exec_host: 1 point, single kernel, tile passes, chirp-z --------
    for (int64_t n : {1, 2, 64, 1024, 16384, 65536, 1000, 24016}) {
        for (int dtype : {MFFT_C64, MFFT_C128}) {
            const int64_t n_in = n > 4 ? n - n / 3 : n, batch = 3;
            mfft_plan plan = nullptr;
            REQUIRE(mfft_plan_1d(&plan, n, n_in, batch, dtype, MFFT_HANN | MFFT_POWER | MFFT_STATS) == MFFT_OK);
            std::vector<double> stats(3 * batch);
            const int64_t k0 = n / 4;
            if (dtype == MFFT_C128) {
                std::vector<cd> x((size_t)(batch * n_in));
                for (int64_t b = 0; b < batch; ++b)
                    for (int64_t j = 0; j < n_in; ++j) x[(size_t)(b * n_in + j)] = std::polar(1.0, 2 * M_PI * (double)k0 * (double)j / (double)n);
                std::vector<double> S((size_t)(batch * n));
                REQUIRE(mfft_exec_host(plan, x.data(), S.data(), 0.5, 0.0, 0.0, 1.0, stats.data()) == MFFT_OK);
                if (n >= 64) REQUIRE(tone_bin_error(S, n, k0) <= 1.0);
                REQUIRE(mfft_exec_host(plan, x.data(), nullptr, 0.5, 0.0, 0.0, 1.0, stats.data()) == MFFT_OK);   // stats only
                if (n > 1) {
                    mfft_plan one = nullptr;
                    REQUIRE(mfft_plan_1d(&one, n, n_in, 1, dtype, MFFT_HANN | MFFT_POWER | MFFT_STATS) == MFFT_OK);
                    std::vector<int64_t> idx(16);
                    int64_t count = -1;
                    REQUIRE(mfft_exec_host_peaks(one, x.data(), S.data(), 0.5, 0.0, 0.0, 1.0, stats.data(), idx.data(), 16, &count) == MFFT_OK);
                    REQUIRE(count >= 0);
                    REQUIRE(mfft_plan_destroy(one) == MFFT_OK);
                }
            } else {
                std::vector<cf> x((size_t)(batch * n_in), cf(1.0f, -1.0f));
                std::vector<float> S((size_t)(batch * n));
                REQUIRE(mfft_exec_host(plan, x.data(), S.data(), 0.5, 0.0, 0.0, 1.0, stats.data()) == MFFT_OK);
            }
            int64_t ws = 0; int launches = 0;
            REQUIRE(mfft_plan_workspace_bytes(plan, &ws) == MFFT_OK && ws >= 0);
            REQUIRE(mfft_plan_launches(plan, &launches) == MFFT_OK && launches >= 1);
            REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
        }
    }
    // ---- pipelined host transfers (above 48 MiB), ragged last slab, twice --------------------------
    {
        const int64_t n = 1024, batch = 8192 + 77;
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d(&plan, n, n, batch, MFFT_C64, MFFT_DEMOD) == MFFT_OK);
        std::vector<cf> x((size_t)(batch * n), cf(0.25f, 0.5f)), y((size_t)(batch * n));
        for (int rep = 0; rep < 2; ++rep) {
            REQUIRE(mfft_exec_host(plan, x.data(), y.data(), 1.0, 0.1, 0.0, 0.01, nullptr) == MFFT_OK);
            REQUIRE(std::isfinite(y.back().real()));
        }
        REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
    }
    // ---- column-batched plan ------------------------------------------------------------------------
    {
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d_columns(&plan, 256, 64, 2, MFFT_C128, 256 * 1024, 64) == MFFT_OK);
        std::vector<cd> x(2 * 256 * 64, cd(1.0, 0.0)), y(x.size());
        REQUIRE(mfft_exec_host(plan, x.data(), y.data(), 1.0, 0.0, 0.0, 1.0, nullptr) == MFFT_OK);
        REQUIRE(std::fabs(y[0].real() - 256.0) < 1e-9);
        REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
        REQUIRE(mfft_plan_1d_columns(&plan, 48, 64, 1, MFFT_C128, 0, 0) == MFFT_E_UNSUPPORTED && plan == nullptr);
    }
    // ---- helper entry points on host data --------------------------------------------------------------
    {
        const int64_t n = 4001;
        std::vector<cd> x((size_t)n), f((size_t)n);
        std::vector<double> t((size_t)n), win((size_t)n, 1.0);
        const double E[3] = {-1.0, 0.4, 2.2};
        for (int64_t j = 0; j < n; ++j) { t[(size_t)j] = 0.01 * (double)j; x[(size_t)j] = std::polar(1.0, -0.4 * t[(size_t)j]); }
        double G[18], r[6], yy = 0.0;
        REQUIRE(mfft_lsq_normal_host(x.data(), win.data(), t.data(), n, 0.0, E, 3, G, r, &yy) == MFFT_OK);
        REQUIRE(std::fabs(yy - (double)n) < 1e-6 && std::fabs(G[0] - (double)n) < 1e-6);
        REQUIRE(mfft_lsq_normal_host(x.data(), nullptr, t.data(), n, 0.0, E, 65, G, r, &yy) == MFFT_E_INVALID);
        const double w[2] = {0.5, -1.5}, c0[4] = {0.6, 0.0, 0.8, 0.0}, vrow[4] = {0.6, 0.0, 0.8, 0.0};
        REQUIRE(mfft_return_amplitude_host(f.data(), n, t.data(), 0.0, 0.0, w, c0, vrow, 2, 1.0) == MFFT_OK);
        REQUIRE(std::abs(f[0] - cd(1.0, 0.0)) < 1e-12);
        REQUIRE(mfft_return_amplitude_host(f.data(), n, nullptr, 0.0, 0.01, w, c0, vrow, 2, 1.0) == MFFT_OK);
    }
    // ---- line search of a batch (peak pipeline + fit on the device) ------------------------------------------
    {
        const int64_t n0 = 2000, n = 16 * n0, batch = 3;
        const double dt = 0.01;
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d(&plan, n, n0, batch, MFFT_C128, MFFT_HANN | MFFT_POWER | MFFT_STATS) == MFFT_OK);
        std::vector<cd> x((size_t)(batch * n0));
        std::vector<double> t((size_t)n0);
        for (int64_t j = 0; j < n0; ++j) t[(size_t)j] = dt * (double)j;
        for (int64_t b = 0; b < batch; ++b)
            for (int64_t j = 0; j < n0; ++j)
                x[(size_t)(b * n0 + j)] = 0.7 * std::polar(1.0, -(1.0 + (double)b) * t[(size_t)j]) + 0.3 * std::polar(1.0, 2.5 * t[(size_t)j]);
        const double domega = 2 * M_PI / (t.back() - t.front());
        mfft_peak_params prm{dt, 1.5 * domega, 1.2 * domega, 0, 0, 4};
        const int want = prm.want;
        std::vector<double> S((size_t)(batch * n)), stats(3 * batch), om((size_t)(batch * want)), hgt((size_t)(batch * want)),
            gram((size_t)batch * (2 * want * want + 2 * want + 1));
        std::vector<int64_t> idx((size_t)(batch * want));
        std::vector<int32_t> cnt((size_t)batch);
        for (int rep = 0; rep < 2; ++rep)
            REQUIRE(mfft_lines_host(plan, x.data(), dt, 0.0, 0.0, dt, t.data(), &prm, 1, S.data(), stats.data(), idx.data(),
                                    om.data(), hgt.data(), cnt.data(), gram.data()) == MFFT_OK);
        for (int64_t b = 0; b < batch; ++b) {
            REQUIRE(cnt[(size_t)b] >= 2);
            REQUIRE(std::fabs(om[(size_t)(b * want)] + (1.0 + (double)b)) < 0.05);       // the tallest line: E = 1 + b, omega = -E
        }
        prm.want = 65;
        REQUIRE(mfft_lines_host(plan, x.data(), dt, 0.0, 0.0, dt, t.data(), &prm, 0, nullptr, stats.data(), idx.data(), om.data(),
                                nullptr, cnt.data(), nullptr) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
    }
    // ---- the entry point that allocates nothing, on device arrays ------------------------------------------------
    {
        const int64_t n = 3000, batch = 2;
        const int m = 5;
        std::vector<cd> x((size_t)(batch * n), cd(1.0, 0.5));
        std::vector<double> t((size_t)n), E((size_t)(batch * m), 0.3), out((size_t)batch * (2 * m * m + 2 * m + 1));
        for (int64_t j = 0; j < n; ++j) t[(size_t)j] = 0.01 * (double)j;
        void *d_x, *d_t, *d_E, *d_o;
        REQUIRE(hipMalloc(&d_x, x.size() * sizeof(cd)) == hipSuccess && hipMalloc(&d_t, t.size() * 8) == hipSuccess &&
                hipMalloc(&d_E, E.size() * 8) == hipSuccess && hipMalloc(&d_o, out.size() * 8) == hipSuccess);
        REQUIRE(hipMemcpy(d_x, x.data(), x.size() * sizeof(cd), hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(hipMemcpy(d_t, t.data(), t.size() * 8, hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(hipMemcpy(d_E, E.data(), E.size() * 8, hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(mfft_lsq_normal_static(d_x, nullptr, (const double*)d_t, n, batch, nullptr, (const double*)d_E, m, (double*)d_o, nullptr) == MFFT_OK);
        REQUIRE(hipMemcpy(out.data(), d_o, out.size() * 8, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(std::fabs(out[0] - (double)n) < 1e-6 && std::fabs(out.back() - 1.25 * (double)n) < 1e-6);
        REQUIRE(mfft_lsq_normal_static(d_x, nullptr, (const double*)d_t, n, 2000, nullptr, (const double*)d_E, 64, (double*)d_o, nullptr) == MFFT_E_UNSUPPORTED);
        (void)hipFree(d_x); (void)hipFree(d_t); (void)hipFree(d_E); (void)hipFree(d_o);
        int64_t fr = 0, tot = 0;
        REQUIRE(mfft_device_memory(&fr, &tot) == MFFT_OK && fr > 0 && fr <= tot);
    }
    // ---- device-buffer executions: multi-pass, windowed + zero-padded tiles, packed real input, recorded into a graph ---
    {
        struct Case { int64_t n, n_in, batch; int dtype; uint32_t flags; };
        const Case cases[] = {{1 << 20, 1 << 20, 2, MFFT_C128, 0}, {1 << 17, 100001, 3, MFFT_C64, MFFT_HANN},
                              {1 << 18, 1 << 18, 2, MFFT_C128, MFFT_REAL_IN}, {1 << 17, (1 << 17) - 2, 2, MFFT_C64, MFFT_REAL_IN | MFFT_HANN | MFFT_POWER | MFFT_STATS},
                              {24016, 1501, 2, MFFT_C128, MFFT_HANN | MFFT_POWER | MFFT_STATS}, {1 << 22, 1 << 22, 1, MFFT_C64, MFFT_POWER}};
        hipStream_t st = nullptr;
        REQUIRE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
        for (const Case& c : cases) {
            const size_t rs = c.dtype == MFFT_C64 ? 4 : 8;
            const size_t in_bytes = (size_t)(c.batch * c.n_in) * rs * ((c.flags & MFFT_REAL_IN) ? 1 : 2);
            const size_t out_bytes = (size_t)(c.batch * c.n) * rs * ((c.flags & MFFT_POWER) ? 1 : 2);
            void *d_in = nullptr, *d_out = nullptr, *d_stats = nullptr;
            REQUIRE(hipMalloc(&d_in, in_bytes) == hipSuccess && hipMalloc(&d_out, out_bytes) == hipSuccess && hipMalloc(&d_stats, 24 * (size_t)c.batch) == hipSuccess);
            REQUIRE(hipMemset(d_in, 0, in_bytes) == hipSuccess);
            mfft_plan plan = nullptr;
            REQUIRE(mfft_plan_1d(&plan, c.n, c.n_in, c.batch, c.dtype, c.flags) == MFFT_OK);
            double* stats = (c.flags & MFFT_STATS) ? (double*)d_stats : nullptr;
            for (int rep = 0; rep < 2; ++rep) REQUIRE(mfft_exec(plan, d_in, d_out, 0.5, 0.0, 0.0, 1.0, stats, st) == MFFT_OK);
            REQUIRE(hipStreamSynchronize(st) == hipSuccess);
            hipGraph_t graph = nullptr;
            hipGraphExec_t gexec = nullptr;
            REQUIRE(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess);
            const int rc = mfft_exec(plan, d_in, d_out, 0.5, 0.0, 0.0, 1.0, stats, st);
            REQUIRE(hipStreamEndCapture(st, &graph) == hipSuccess);
            REQUIRE(rc == MFFT_OK && graph != nullptr);
            REQUIRE(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess);
            for (int rep = 0; rep < 2; ++rep) REQUIRE(hipGraphLaunch(gexec, st) == hipSuccess);
            REQUIRE(hipStreamSynchronize(st) == hipSuccess);
            (void)hipGraphExecDestroy(gexec); (void)hipGraphDestroy(graph);
            REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
            (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_stats);
        }
        (void)hipStreamDestroy(st);
    }
    // ---- layouts, block copies, column groups, device-side helpers ---------------------------------------------------
    {
        const int64_t n = 4096, rows = 8, seg = 512;
        mfft_layout lin{0, 0, 0}, lout{rows * seg, seg, rows * seg};          // bins of row b land at [k / seg][b][k % seg]
        lout.batch_stride = seg;
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d_layout(&plan, n, n, rows, MFFT_C64, 0, &lin, &lout) == MFFT_OK);
        void *d_a = nullptr, *d_b = nullptr, *d_cnt = nullptr, *d_idx = nullptr, *d_max = nullptr;
        const size_t bytes = (size_t)(rows * n) * 8;
        REQUIRE(hipMalloc(&d_a, bytes) == hipSuccess && hipMalloc(&d_b, bytes) == hipSuccess && hipMalloc(&d_cnt, 8) == hipSuccess &&
                hipMalloc(&d_idx, 8 * 1024) == hipSuccess && hipMalloc(&d_max, 8) == hipSuccess);
        REQUIRE(hipMemset(d_a, 0, bytes) == hipSuccess && hipMemset(d_cnt, 0, 8) == hipSuccess);
        REQUIRE(mfft_exec(plan, d_a, d_b, 1.0, 0.0, 0.0, 1.0, nullptr, nullptr) == MFFT_OK);
        REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
        REQUIRE(mfft_swap_axes(d_a, d_b, 8, 8, n / 8, MFFT_C64, nullptr) == MFFT_OK);
        REQUIRE(mfft_swap_axes_strided(d_a, d_b, 4, 8, 128, n, 128, MFFT_C64, nullptr) == MFFT_OK);
        REQUIRE(mfft_transpose(d_a, d_b, rows, n, 1, MFFT_C64, nullptr) == MFFT_OK);
        REQUIRE(mfft_twiddle_tile(d_a, rows, n, 3, 5, rows * n * 4, MFFT_C64, nullptr) == MFFT_OK);
        REQUIRE(mfft_local_maxima(d_a, 2 * rows * n, MFFT_C64, (int64_t*)d_idx, 1024, (int64_t*)d_cnt, nullptr) == MFFT_OK);
        const double one = 2.0;
        REQUIRE(hipMemcpy(d_max, &one, 8, hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(mfft_normalize_by_max(d_a, d_b, 2 * rows * n, (const double*)d_max, MFFT_C64, nullptr) == MFFT_OK);
        mfft_plan cols = nullptr;                                              // a column group written into a wider array
        REQUIRE(mfft_plan_1d_columns_ld(&cols, 256, 64, 1, MFFT_C64, 0, 0, 128) == MFFT_OK);
        REQUIRE(mfft_exec(cols, d_a, d_b, 1.0, 0.0, 0.0, 1.0, nullptr, nullptr) == MFFT_OK);
        REQUIRE(mfft_plan_destroy(cols) == MFFT_OK);
        const double w[2] = {0.5, -1.5}, c0[4] = {0.6, 0.0, 0.8, 0.0}, vrow[4] = {0.6, 0.0, 0.8, 0.0};
        REQUIRE(mfft_return_amplitude(d_b, 1000, 0.0, 0.01, w, c0, vrow, 2, 1.0, nullptr) == MFFT_OK);
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_cnt); (void)hipFree(d_idx); (void)hipFree(d_max);
    }
    // ---- the block pipeline (samples generated on the device) into a page-locked spectrum ------------------------------
    {
        const int64_t n0 = 4001, n = 32 * n0, batch = 3;
        const int nk = 3, want = 3;
        const double dt = 0.01;
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d(&plan, n, n0, batch, MFFT_C128, MFFT_HANN | MFFT_POWER | MFFT_STATS) == MFFT_OK);
        std::vector<double> t((size_t)n0), w((size_t)(batch * nk)), c0((size_t)(2 * batch * nk), 0.0), vrow((size_t)(2 * batch * nk), 0.0);
        for (int64_t j = 0; j < n0; ++j) t[(size_t)j] = dt * (double)j;
        for (int64_t b = 0; b < batch; ++b)
            for (int k = 0; k < nk; ++k) {
                w[(size_t)(b * nk + k)] = -2.0 + 1.7 * k + 0.1 * (double)b;
                c0[(size_t)(2 * (b * nk + k))] = vrow[(size_t)(2 * (b * nk + k))] = std::sqrt((k + 1) / 6.0);
            }
        const double domega = 2 * M_PI / (t.back() - t.front());
        mfft_peak_params prm{dt, 1.5 * domega, 1.2 * domega, 0, 0, want};
        void* S = nullptr;
        REQUIRE(mfft_host_alloc(&S, (int64_t)(batch * n) * 8) == MFFT_OK && S != nullptr);
        std::vector<double> S2((size_t)(batch * n)), stats(3 * batch), om((size_t)(batch * want)), hgt((size_t)(batch * want)),
            gram((size_t)batch * (2 * want * want + 2 * want + 1));
        std::vector<int64_t> idx((size_t)(batch * want));
        std::vector<int32_t> cnt((size_t)batch);
        REQUIRE(mfft_lines_blocks_host(plan, w.data(), c0.data(), vrow.data(), nk, 1.0, dt, t.data(), &prm, 1, S, stats.data(), idx.data(),
                                       om.data(), hgt.data(), cnt.data(), gram.data()) == MFFT_OK);
        REQUIRE(mfft_lines_blocks_host(plan, w.data(), c0.data(), vrow.data(), nk, 1.0, dt, t.data(), &prm, 1, S2.data(), stats.data(), idx.data(),
                                       om.data(), hgt.data(), cnt.data(), gram.data()) == MFFT_OK);
        for (int64_t b = 0; b < batch; ++b) REQUIRE(cnt[(size_t)b] == want);
        double worst = 0.0;
        for (size_t i = 0; i < S2.size(); ++i) worst = std::fmax(worst, std::fabs(S2[i] - static_cast<const double*>(S)[i]));
        REQUIRE(worst == 0.0);                                                  // the page-locked and the pageable destination agree
        REQUIRE(mfft_host_free(S) == MFFT_OK && mfft_host_free(nullptr) == MFFT_OK);
        // the other app's estimator: rough pass -> centroid -> demodulated pass -> peaks, then the fit on resident samples
        mfft_plan main1 = nullptr, rough1 = nullptr;
        REQUIRE(mfft_plan_1d(&main1, n, n0, 1, MFFT_C128, MFFT_HANN | MFFT_POWER | MFFT_STATS | MFFT_DEMOD) == MFFT_OK);
        REQUIRE(mfft_plan_1d(&rough1, n, n0, 1, MFFT_C128, MFFT_HANN | MFFT_POWER | MFFT_STATS) == MFFT_OK);
        const double Ecols[4] = {1.7, 0.3, -1.7, -0.3};
        std::vector<double> g4(2 * 16 + 8 + 1);
        double mu1 = -1.0;
        REQUIRE(mfft_lines_gram_host(main1, Ecols, 4, 0.0, g4.data()) == MFFT_E_INVALID);          // nothing resident yet
        REQUIRE(mfft_lines_extract_host(main1, plan, w.data(), c0.data(), vrow.data(), nk, 1.0, dt, 0.0, dt, t.data(), &prm, S2.data(),
                                        stats.data(), idx.data(), om.data(), hgt.data(), cnt.data(), &mu1) == MFFT_E_INVALID);   // a batch plan as rough pass
        for (int rep = 0; rep < 2; ++rep)
            REQUIRE(mfft_lines_extract_host(main1, rough1, w.data(), c0.data(), vrow.data(), nk, 1.0, dt, 0.0, dt, t.data(), &prm, S2.data(),
                                            stats.data(), idx.data(), om.data(), hgt.data(), cnt.data(), &mu1) == MFFT_OK);
        REQUIRE(std::isfinite(mu1) && mu1 != 0.0 && cnt[0] >= 1);
        REQUIRE(mfft_lines_gram_host(main1, Ecols, 4, mu1, g4.data()) == MFFT_OK);
        REQUIRE(g4.back() > 0.0 && std::fabs(g4[0] - g4[2 * 5]) < 1e-9 * g4[0]);                    // yy > 0, equal diagonal entries
        REQUIRE(mfft_lines_gram_host(main1, Ecols, 65, mu1, g4.data()) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_destroy(main1) == MFFT_OK && mfft_plan_destroy(rough1) == MFFT_OK);
        REQUIRE(mfft_host_alloc(nullptr, 8) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_destroy(plan) == MFFT_OK);
    }
    // ---- distributed plans: four ranks sharing the device, a host thread each -----------------------------------
    {
        const int P = 4;
        const int64_t n0 = 256, n1 = 128;
        mfft_comm comms[P];
        REQUIRE(mfft_comm_init_local(comms, P) == MFFT_OK);
        std::vector<int> rcs((size_t)P, -100);
        std::vector<cf> result((size_t)(n0 * n1));
        std::vector<std::thread> th;
        for (int r = 0; r < P; ++r)
            th.emplace_back([&, r] {
                const int64_t local = n0 / P * n1;
                std::vector<cf> mine((size_t)local, cf(1.0f, 0.0f));          // constant input: F[0][0] = n0*n1, zero elsewhere
                void *d_in = nullptr, *d_out = nullptr;
                mfft_dist_plan p2 = nullptr, p1 = nullptr;
                int rc = hipMalloc(&d_in, (size_t)local * 8) == hipSuccess && hipMalloc(&d_out, (size_t)local * 8) == hipSuccess ? 0 : -99;
                if (!rc) rc = hipMemcpy(d_in, mine.data(), (size_t)local * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : -98;
                if (!rc) rc = mfft_plan_2d_dist(&p2, comms[r], n0, n1, MFFT_C64, 1, 0, 0);
                int chunks = 0, groups = 0; int64_t ws = 0;
                if (!rc) rc = mfft_dist_plan_info(p2, &chunks, &groups, &ws);
                if (!rc && (chunks < 1 || groups < 1 || ws <= 0)) rc = -95;
                if (!rc) rc = mfft_dist_exec(p2, d_in, d_out, nullptr);
                if (!rc) rc = mfft_plan_1d_dist(&p1, comms[r], n0, n1, MFFT_C64, 1, 2, 0);
                if (!rc) rc = mfft_dist_exec(p1, d_in, d_out, nullptr);       // result overwritten below: exercised for its host side
                if (!rc) rc = mfft_dist_exec(p2, d_in, d_out, nullptr);
                if (!rc) rc = hipDeviceSynchronize() == hipSuccess ? 0 : -97;
                if (!rc) rc = hipMemcpy(result.data() + (size_t)r * (size_t)local, d_out, (size_t)local * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -96;
                (void)mfft_dist_plan_destroy(p2); (void)mfft_dist_plan_destroy(p1);
                (void)hipFree(d_in); (void)hipFree(d_out);
                rcs[(size_t)r] = rc;
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < P; ++r) REQUIRE(rcs[(size_t)r] == MFFT_OK);
        REQUIRE(std::fabs(result[0].real() - (float)(n0 * n1)) < 1e-2f && std::abs(result[1]) < 1e-2f && std::abs(result.back()) < 1e-2f);
        int world = 0, rank = -1;
        REQUIRE(mfft_comm_info(comms[2], &world, &rank) == MFFT_OK && world == P && rank == 2);
        for (int r = 0; r < P; ++r) REQUIRE(mfft_comm_destroy(comms[r]) == MFFT_OK);
        mfft_dist_plan bad = nullptr;
        REQUIRE(mfft_plan_2d_dist(&bad, nullptr, 64, 64, MFFT_C64, 0, 0, 0) == MFFT_E_INVALID);
    }
    // ---- error paths ---------------------------------------------------------------------------------------
    {
        mfft_plan plan = nullptr;
        REQUIRE(mfft_plan_1d(&plan, 0, 0, 1, MFFT_C64, 0) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(&plan, 8, 9, 1, MFFT_C64, 0) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(&plan, 8, 8, 0, MFFT_C64, 0) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(&plan, 8, 8, 1, 7, 0) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(&plan, 8, 8, 1, MFFT_C64, 0x4000u) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(&plan, 8, 8, 1, MFFT_C64, MFFT_STATS) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_1d(nullptr, 8, 8, 1, MFFT_C64, 0) == MFFT_E_INVALID);
        REQUIRE(mfft_exec_host(nullptr, nullptr, nullptr, 1, 0, 0, 1, nullptr) == MFFT_E_INVALID);
        REQUIRE(mfft_plan_destroy(nullptr) == MFFT_OK);
        REQUIRE(mfft_strerror(MFFT_E_HIP) != nullptr && mfft_version() > 0);
    }
    printf("abi_smoke ok\n");
    fflush(stdout);
    return 0;
}
