// Instantiates the lane-distributed kernels (transforms of at most 64 / 32 points).
#include "wave_kernel.h"

namespace mfft {
int launch_wave_c64(int log2n, bool win, bool power, bool real_in, const WaveArgs<float>& a, hipStream_t s) { return launch_wave_any<float>(log2n, win, power, real_in, a, s); }
int launch_wave_c128(int log2n, bool win, bool power, bool real_in, const WaveArgs<double>& a, hipStream_t s) { return launch_wave_any<double>(log2n, win, power, real_in, a, s); }
}  // namespace mfft
