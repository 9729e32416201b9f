// How fast can a pageable host buffer reach the GPU?  plain hipMemcpy vs hipHostRegister + async copy
// vs staged copies through a pinned buffer filled by 1..8 host threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void par_memcpy(char* dst, const char* src, size_t bytes, int threads)
{
    std::vector<std::thread> th;
    size_t per = (bytes + threads - 1) / threads;
    for (int i = 0; i < threads; ++i) {
        size_t lo = i * per, hi = lo + per > bytes ? bytes : lo + per;
        if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}
int main()
{
    const size_t bytes = 512ull << 20;
    char* h = nullptr;
    if (posix_memalign((void**)&h, 4096, bytes)) return 1;
    memset(h, 1, bytes);
    char* hout = (char*)malloc(bytes);          // untouched: first-touch page faults on D2H
    void* d; CK(hipMalloc(&d, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); double t1 = now();
        printf("hipMemcpy H2D pageable: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    { double t0 = now(); CK(hipMemcpy(hout, d, bytes, hipMemcpyDeviceToHost)); double t1 = now();
      printf("hipMemcpy D2H pageable, untouched destination: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9); }
    { double t0 = now(); CK(hipMemcpy(hout, d, bytes, hipMemcpyDeviceToHost)); double t1 = now();
      printf("hipMemcpy D2H pageable, touched destination: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9); }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        hipError_t er = hipHostRegister(h, bytes, hipHostRegisterDefault);
        if (er != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(er)); (void)hipGetLastError(); break; }
        double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipHostUnregister(h)); double t3 = now();
        printf("register %.1f ms, async H2D %.1f ms (%.1f GB/s), unregister %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    char* pin; CK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault));
    for (int threads : {1, 2, 4, 8, 16}) {
        double t0 = now(); par_memcpy(pin, h, bytes, threads); double t1 = now();
        printf("memcpy pageable -> pinned, %2d threads: %.1f ms (%.1f GB/s)\n", threads, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    { double t0 = now(); CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t1 = now();
      printf("pinned H2D: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9); }
    { double t0 = now(); CK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t1 = now();
      printf("pinned D2H: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9); }
    char* fresh = (char*)malloc(bytes);
    for (int threads : {1, 8}) {
        double t0 = now(); par_memcpy(fresh, pin, bytes, threads); double t1 = now();
        printf("memcpy pinned -> pageable (%s), %d threads: %.1f ms (%.1f GB/s)\n", threads == 1 ? "untouched" : "touched", threads,
               (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
