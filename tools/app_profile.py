#!/usr/bin/env python3
"""Where one fft_from_timeseries call spends its time: cProfile over repeated calls at the final2 defaults.
    python tools/app_profile.py [calls]"""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
from oracle import spectral_oracle as so   # input generator only

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n0, pad, dt = 8001, 32, 0.01
times = np.arange(n0) * dt
f_t = so.return_amplitude_block([1, 1, 1], [1, 1], 1, 0, times)
kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
for _ in range(20):
    fft_amd.fft_from_timeseries(f_t, times, **kw)
t0 = time.perf_counter()
for _ in range(calls):
    fft_amd.fft_from_timeseries(f_t, times, **kw)
print(f"plain loop: {(time.perf_counter() - t0) / calls * 1e3:.3f} ms per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    fft_amd.fft_from_timeseries(f_t, times, **kw)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
