// Instantiates the lane-distributed kernels (transforms of at most 64 / 32 points).
#include "wave_kernel.h"

namespace mfft {
int launch_wave_c64(int log2n, bool hann_power, const WaveArgs<float>& a, hipStream_t s) { return launch_wave_any<float>(log2n, hann_power, a, s); }
int launch_wave_c128(int log2n, bool hann_power, const WaveArgs<double>& a, hipStream_t s) { return launch_wave_any<double>(log2n, hann_power, a, s); }
}  // namespace mfft
