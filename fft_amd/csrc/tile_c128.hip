// Instantiates the global-pass (tile) kernels for complex128.
#include "tile_kernel.h"

namespace mfft {
int launch_tile_c128(int log2g, int mode, const TArgs<double>& a, hipStream_t s)
{
    return launch_tile_any<double>(log2g, mode, a, s);
}
int tile_log2c_c128(int log2g, int narrow) { return tile_log2c_for<double>(log2g, narrow); }
bool tile_geometry_c128(int log2g, LdsGeometry* g) { return tile_geometry_for<double>(log2g, g); }
bool tile_first_hann_c128() { return tile_first_hann<double, 9>(); }
}  // namespace mfft
