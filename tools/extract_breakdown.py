import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fft_amd import spectral as sp
H = np.array([[0.3, 1.0, 0.0], [1.0, -0.2, 0.7], [0.0, 0.7, 0.5]])
times = np.arange(0.0, 15.0 + 0.01, 0.01)
for _ in range(20): sp.fft_extract_E_p(H, 0, times)
def med(f, n=200):
    v=[]
    for _ in range(n):
        t0=time.perf_counter(); f(); v.append(time.perf_counter()-t0)
    return sorted(v)[n//2]*1e6
print("fft_extract_E_p total", med(lambda: sp.fft_extract_E_p(H, 0, times)))
f_t = sp.return_amplitude(H, 0, times)
print("return_amplitude", med(lambda: sp.return_amplitude(H, 0, times)))
print("power_spectrum stats only", med(lambda: sp.power_spectrum(f_t, 16*len(f_t), 0.01, use_hann=True, want_spectrum=False)))
print("fft_from_timeseries same shape", med(lambda: sp.fft_from_timeseries(f_t, times, n_levels=3, pad_factor=16)))
