#!/usr/bin/env python3
"""cProfile of the apps' call shapes (final2 defaults) on the GPU path: one single call and one
batched call of three blocks, 200 repetitions each; shows where the host time of a call goes.
    python tools/app_cprofile.py"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fft_amd
from oracle import spectral_oracle as so

n0, pad, dt = 8001, 32, 0.01
times = np.arange(n0) * dt
blocks = [so.return_amplitude_block([1, 1, 1], [1, 1], m, 0, times) for m in (0, 1, 2)]
kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
for label, call in (("single", lambda: fft_amd.fft_from_timeseries(blocks[1], times, **kw)),
                    ("batch of three", lambda: fft_amd.fft_from_timeseries_batch(blocks, times, **kw))):
    for _ in range(20):
        call()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        call()
    pr.disable()
    print("=====", label)
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
