// Host copy threads of the pinned-slab pipeline (mfft_exec_host on large batches): one pool per
// direction, each copying one job at a time split over its threads.  Included by plan.hip only.
#pragma once
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <stdint.h>

namespace mfft {

void stream_copy(void* dst, const void* src, size_t bytes);   // hostcopy.cpp

class CopyPool {
public:
    // one pool per direction: uploads and downloads are copied concurrently
    static CopyPool& in()
    {
        static CopyPool* pool = new CopyPool;      // never destroyed: its threads outlive main()
        return *pool;
    }
    static CopyPool& out()
    {
        static CopyPool* pool = new CopyPool;
        return *pool;
    }
    // Starts a copy on the pool's threads and returns its ticket; wait(ticket) returns once that
    // copy is complete.  Jobs of concurrent callers run one after the other.
    uint64_t start(void* dst, const void* src, size_t bytes)
    {
        std::unique_lock<std::mutex> lk(mu_);
        idle_.wait(lk, [&] { return pending_ == 0; });
        dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes;
        parts_ = bytes < (size_t(1) << 20) ? 1 : nthreads_;
        next_ = 0; pending_ = parts_;
        const uint64_t ticket = ++generation_;
        work_.notify_all();
        return ticket;
    }
    void wait(uint64_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu_);
        idle_.wait(lk, [&] { return done_ >= ticket; });
    }
    void copy(void* dst, const void* src, size_t bytes) { wait(start(dst, src, bytes)); }

private:
    CopyPool()
    {
        int n = 8;                                  // measured on the GPU box: 4 -> 8 -> 12 threads = 27 -> 21 -> 21 ms per GiB moved
        if (const char* env = getenv("MFFT_COPY_THREADS")) n = atoi(env);
        nthreads_ = n < 1 ? 1 : (n > 32 ? 32 : n);
        if (const char* env = getenv("MFFT_COPY_STREAM")) stream_ = atoi(env) != 0;   // tuning aid
        for (int i = 0; i < nthreads_; ++i) std::thread([this] { loop(); }).detach();
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            work_.wait(lk, [&] { return next_ < parts_; });
            while (next_ < parts_) {
                const int part = next_++;
                const size_t per = (bytes_ + (size_t)parts_ - 1) / (size_t)parts_;
                const size_t lo = (size_t)part * per, hi = lo + per > bytes_ ? bytes_ : lo + per;
                char* d = dst_; const char* sr = src_;
                const uint64_t gen = generation_;
                lk.unlock();
                if (lo < hi) {
                    if (stream_) stream_copy(d + lo, sr + lo, hi - lo);
                    else memcpy(d + lo, sr + lo, hi - lo);
                }
                lk.lock();
                if (--pending_ == 0) { done_ = gen; idle_.notify_all(); }
            }
        }
    }
    std::mutex mu_;
    std::condition_variable work_, idle_;
    char* dst_ = nullptr; const char* src_ = nullptr;
    size_t bytes_ = 0;
    int next_ = 0, parts_ = 0, pending_ = 0, nthreads_ = 1;
    uint64_t generation_ = 0, done_ = 0;
    bool stream_ = true;
};

}  // namespace mfft
