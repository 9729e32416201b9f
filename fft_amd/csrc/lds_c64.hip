// Instantiates the LDS-resident Stockham kernels for complex64 (fp32 butterflies)
// and exports the size table (per precision) to the host code.
#include "lds_kernel.h"

namespace mfft {

int launch_lds_c64(int log2n, int variant, const KArgs<float>& a, int grid, hipStream_t s)
{
    return launch_lds_any<float, kMaxLdsLog2C64>(log2n, variant, a, grid, s);
}

template <typename T, int L> static void fill_geometry(int dtype, LdsGeometry* g)
{
    using SC = SizeFor<T, L>;
    g->n = 1 << L;
    g->p = SC::P;
    g->tx = BlockCfg<T, L>::TX;
    g->block = BlockCfg<T, L>::BLOCK;
    g->slots = g->block / g->tx;
    g->npass = SC::RD::npass;
    for (int i = 0; i < 4; ++i) g->radix[i] = SC::RD::r[i];
    g->split16 = SC::RD::split16 ? 1 : 0;
    g->lds_bytes = (SC::RD::npass > 1)
                       ? lds_bytes_for<T, L>()
                       : 0;
    (void)dtype;
    size_t tw = 0;
    for (int s = 1; s < SC::RD::npass; ++s) tw += (size_t)tw_pass_elems<typename SC::RD>(s);
    g->tw_elems = tw;
}

bool lds_geometry(int dtype, int log2n, LdsGeometry* g)
{
    const int maxlog = dtype == MFFT_C64 ? kMaxLdsLog2C64 : kMaxLdsLog2C128;
    if (log2n < 1 || log2n > maxlog) return false;
    switch (log2n) {
#define MFFT_G(L) case L: if (dtype == MFFT_C64) fill_geometry<float, L>(dtype, g); else fill_geometry<double, L>(dtype, g); return true;
        MFFT_G(1) MFFT_G(2) MFFT_G(3) MFFT_G(4) MFFT_G(5) MFFT_G(6) MFFT_G(7)
        MFFT_G(8) MFFT_G(9) MFFT_G(10) MFFT_G(11) MFFT_G(12) MFFT_G(13) MFFT_G(14)
#undef MFFT_G
    }
    return false;
}

}  // namespace mfft

#if MFFT_DIAG_STAMPS
// diagnostic builds only: the phase cycle counts the last LDS-DMA launch left behind (16 per workgroup)
extern "C" int mfft_diag_read(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mfft::g_diag_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif
