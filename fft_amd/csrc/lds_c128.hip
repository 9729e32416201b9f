// Instantiates the LDS-resident Stockham kernels for complex128 (fp64 butterflies).
#include "lds_kernel.h"

namespace mfft {
int launch_lds_c128(int log2n, int variant, const KArgs<double>& a, int grid, hipStream_t s)
{
    return launch_lds_any<double, kMaxLdsLog2C128>(log2n, variant, a, grid, s);
}
}  // namespace mfft
