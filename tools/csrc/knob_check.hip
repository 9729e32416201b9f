// Compile check of the kernels' build knobs (tests/test_isa_invariants.py: hipcc -fsyntax-only with the
// non-default values of the MFFT_* tuning macros).  Instantiates every variant of a few representative
// sizes of each kernel family, so that the branches a default build never takes are still type-checked.
#include "../../fft_amd/csrc/tile_kernel.h"
#include "../../fft_amd/csrc/wave_kernel.h"

namespace mfft {
// host-side pieces the launchers refer to
void set_error(const std::string&) {}
int hip_fail(hipError_t, const char*, const char*, int) { return MFFT_E_HIP; }
int blocks_per_cu(const void*, int, size_t) { return 1; }
int pick_grid(int, int64_t) { return 1; }

template int launch_variant<float, 6>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<float, 10>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<float, 11>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<float, 12>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<float, 13>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<float, 14>(int, const KArgs<float>&, int, hipStream_t);
template int launch_variant<double, 9>(int, const KArgs<double>&, int, hipStream_t);
template int launch_variant<double, 11>(int, const KArgs<double>&, int, hipStream_t);
template int launch_variant<double, 13>(int, const KArgs<double>&, int, hipStream_t);
template int launch_tile_any<float>(int, int, const TArgs<float>&, hipStream_t);
template int launch_tile_any<double>(int, int, const TArgs<double>&, hipStream_t);
}  // namespace mfft
