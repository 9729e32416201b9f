import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fft_amd
from oracle import spectral_oracle as so
n0, pad, dt = 8001, 32, 0.01
times = np.arange(n0) * dt
f_t = so.return_amplitude_block([1, 1, 1], [1, 1], 1, 0, times)
kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
for _ in range(20): fft_amd.fft_from_timeseries(f_t, times, **kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fft_amd.fft_from_timeseries(f_t, times, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
