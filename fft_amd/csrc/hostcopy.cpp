// Host-side block copy with streaming (non-temporal) stores, for the pinned-slab pipeline of
// mfft_exec_host.  A plain memcpy of a slab part (a few MiB per thread) stays below glibc's own
// streaming threshold, so every destination line is first read into the cache (read for
// ownership): three bytes of memory traffic per byte copied.  Both destinations here are written
// once and not read again by this core -- the pinned staging buffer is read by the DMA engine, the
// caller's output array by the caller, later -- so the stores bypass the cache: two bytes per byte.
#include <emmintrin.h>
#include <stdint.h>
#include <string.h>

namespace mfft {

void stream_copy(void* dst, const void* src, size_t bytes)
{
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    if (bytes < 4096) { memcpy(d, s, bytes); return; }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;   // align the stores
    if (head) { memcpy(d, s, head); d += head; s += head; bytes -= head; }
    size_t blocks = bytes / 64;
    while (blocks--) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 32));
        const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(d), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
        s += 64;
        d += 64;
    }
    _mm_sfence();                                  // streaming stores are weakly ordered
    const size_t tail = bytes & 63;
    if (tail) memcpy(d, s, tail);
}

}  // namespace mfft
