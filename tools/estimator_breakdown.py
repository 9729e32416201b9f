#!/usr/bin/env python3
"""Where one fft_from_timeseries call spends its time: the Python prologue (validation, staging), the C call
(mfft_lines_host: upload, transform, peak pipeline, normal equations, downloads), and the Python tail (axis, weights,
pruning, sorting); then fft_extract_E_p and two of its four device round trips.
    python tools/estimator_breakdown.py [--n0 8001 --pad 32]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import spectral as sp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n0", type=int, default=8001)
    ap.add_argument("--pad", type=int, default=32)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    times = np.arange(args.n0) * 0.01
    rng = np.random.default_rng(0)
    E = np.array([-1.3, 0.2, 1.7])
    f = (np.array([0.5, 0.3, 0.2])[None, :] * np.exp(-1j * times[:, None] * E[None, :])).sum(axis=1)
    f = f + 1e-6 * (rng.standard_normal(args.n0) + 1j * rng.standard_normal(args.n0))
    for _ in range(20):
        sp.fft_from_timeseries(f, times, n_levels=3, pad_factor=args.pad)
    spans = {}
    orig_lines = sp.Plan.exec_host_lines if hasattr(sp, "Plan") else None
    from fft_amd import engine
    real = engine.Plan.exec_host_lines

    def timed_lines(self, *a, **k):
        t0 = time.perf_counter()
        r = real(self, *a, **k)
        spans.setdefault("exec_host_lines (ctypes + C)", []).append(time.perf_counter() - t0)
        return r
    engine.Plan.exec_host_lines = timed_lines
    real_tail = sp._tuples_from_lines

    def timed_tail(*a, **k):
        t0 = time.perf_counter()
        r = real_tail(*a, **k)
        spans.setdefault("python tail (_tuples_from_lines)", []).append(time.perf_counter() - t0)
        return r
    sp._tuples_from_lines = timed_tail
    from fft_amd import _lib
    lib = _lib.load()
    real_c = lib.mfft_lines_host

    def timed_c(*a):
        t0 = time.perf_counter()
        r = real_c(*a)
        spans.setdefault("mfft_lines_host (C only)", []).append(time.perf_counter() - t0)
        return r
    try:
        lib.mfft_lines_host = timed_c
    except Exception:
        pass
    tot = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        sp.fft_from_timeseries(f, times, n_levels=3, pad_factor=args.pad)
        tot.append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6  # noqa: E731
    print(f"N0={args.n0} Nfft={args.n0 * args.pad}: fft_from_timeseries median {med(tot):.0f} us")
    for k, v in spans.items():
        print(f"    {k:40s} {med(v):7.0f} us")
    # the combined app's estimator (streamlit_app_combined.py:190-251) and its device round trips
    H = np.array([[0.3, 1.0, 0.0], [1.0, -0.2, 0.7], [0.0, 0.7, 0.5]])

    def timed(fn):
        v = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            v.append(time.perf_counter() - t0)
        return med(v)
    f3 = sp.return_amplitude(H, 0, times)
    print(f"N0={args.n0} pad={args.pad}: fft_extract_E_p median {timed(lambda: sp.fft_extract_E_p(H, 0, times, pad=args.pad)):.0f} us")
    print(f"    {'return_amplitude (generator + download)':40s} {timed(lambda: sp.return_amplitude(H, 0, times)):7.0f} us")
    print(f"    {'rough pass (statistics only)':40s} "
          f"{timed(lambda: sp.power_spectrum(f3, args.pad * args.n0, 0.01, use_hann=True, want_spectrum=False)):7.0f} us")


if __name__ == "__main__":
    main()
