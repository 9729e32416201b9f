#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-facing entry points (host array in, host array out).
    python tools/host_path_bench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402


def best(f, reps=5):
    b = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        b = min(b, time.perf_counter() - t0)
    return b


rng = np.random.default_rng(0)
for batch, n, dt in ((65536, 1024, np.complex64), (4096, 16384, np.complex64), (16, 1 << 22, np.complex64),
                     (8192, 4096, np.complex128)):
    x = (rng.standard_normal((batch, n), dtype=np.float32) + 1j * rng.standard_normal((batch, n), dtype=np.float32)).astype(dt)
    fft_amd.fft(x[:8])
    t = best(lambda: fft_amd.fft(x))
    mb = x.nbytes / 2**20
    t_np = best(lambda: np.fft.fft(x[: max(1, batch // 16)]), reps=2) * 16
    print(f"fft {batch} x {n} {np.dtype(dt).name}: {t*1e3:8.1f} ms  ({2*mb/t/1024:5.1f} GiB/s in+out)   numpy ~{t_np*1e3:8.0f} ms  x{t_np/t:5.1f}", flush=True)
    plan = fft_amd.Plan(n, batch=batch, dtype=0 if dt == np.complex64 else 1, flags=fft_amd.HANN | fft_amd.POWER)
    t = best(lambda: plan.exec_host(x, scale=0.01))
    print(f"    hann+power plan.exec_host: {t*1e3:8.1f} ms  ({1.5*mb/t/1024:5.1f} GiB/s in+out)", flush=True)
    plan.destroy()
