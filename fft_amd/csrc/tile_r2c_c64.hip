// Instantiates the fused real-input last passes for complex64.
#include "tile_r2c_kernel.h"

namespace mfft {
int launch_tile_r2c_c64(int log2g, const TArgs<float>& a, hipStream_t s) { return launch_tile_r2c_any<float>(log2g, a, s); }
}  // namespace mfft
