// Instantiates the fused real-input last passes for complex128.
#include "tile_r2c_kernel.h"

namespace mfft {
int launch_tile_r2c_c128(int log2g, const TArgs<double>& a, hipStream_t s) { return launch_tile_r2c_any<double>(log2g, a, s); }
}  // namespace mfft
