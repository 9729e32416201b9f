// Host-side exerciser of the C ABI for sanitizer builds (AddressSanitizer on host code only):
//   make -C fft_amd/csrc VARIANT=asan KNOBS="-Xarch_host -fsanitize=address -g"
//   g++ -std=c++17 -fsanitize=address -g tools/csrc/abi_smoke.cpp -Iinclude -Lfft_amd/lib -lmfft_asan \
//       -Wl,-rpath,$PWD/fft_amd/lib -o tools/csrc/abi_smoke && tools/csrc/abi_smoke
// Walks every plan kind, the host entry points (small and pipelined transfers), the helper entry
// points and the error paths; checks results loosely (parity is the Python tests' job).
#include <complex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
