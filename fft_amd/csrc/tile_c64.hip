// Instantiates the global-pass (tile) kernels for complex64.
#include "tile_kernel.h"

namespace mfft {
int launch_tile_c64(int log2g, int mode, const TArgs<float>& a, hipStream_t s)
{
    return launch_tile_any<float>(log2g, mode, a, s);
}
int tile_log2c_c64(int log2g, int narrow) { return tile_log2c_for<float>(log2g, narrow); }
bool tile_geometry_c64(int log2g, LdsGeometry* g) { return tile_geometry_for<float>(log2g, g); }
bool tile_first_hann_c64() { return tile_first_hann<float, 9>(); }
}  // namespace mfft
