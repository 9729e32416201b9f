#!/usr/bin/env python3
"""cProfile of the drop-in estimators at the reference apps' default shapes (where the host
time of a call goes once the transform itself takes ~0.2 ms).
    python tools/app_profile.py"""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402
from oracle import spectral_oracle as so  # noqa: E402  (input generator only)

n0, pad, dt = 8001, 32, 0.01
times = np.arange(n0) * dt
f_t = so.return_amplitude_block([1, 1, 1], [1, 1], 1, 0, times)
kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
for _ in range(5):
    fft_amd.fft_from_timeseries(f_t, times, **kw)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fft_amd.fft_from_timeseries(f_t, times, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
