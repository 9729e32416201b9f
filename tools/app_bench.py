#!/usr/bin/env python3
"""The reference apps' own call shapes on the GPU path vs the numpy path, same host.

    python tools/app_bench.py

final2 defaults: N0 = 8001 -> Nfft = 256032 (pad 32); combined defaults: N0 = 1501 -> 24016."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402
from oracle import spectral_oracle as so  # noqa: E402  (CPU path timed beside, checker only)


def best(f, reps=5):
    b = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        b = min(b, time.perf_counter() - t0)
    return b


def loop(calls):
    """`calls` executions of fft_from_timeseries at final2's defaults and nothing else: the workload of a kernel
    timeline (tools/trace_app.sh) or of `python -m cProfile -s tottime tools/app_bench.py --loop 300`."""
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


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--loop":
        torch.cuda.set_device(0)
        return loop(int(sys.argv[2]))
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    for n0, pad in ((8001, 32), (1501, 16), (8001, 64)):
        nfft = n0 * pad
        dt = 0.01
        times = np.arange(n0) * dt
        f_t = so.return_amplitude_block([1, 1, 1], [1, 1], 1, 0, times)
        # device-resident plan
        plan = fft_amd.Plan(nfft, n_in=n0, dtype=fft_amd.C128, flags=fft_amd.HANN | fft_amd.POWER | fft_amd.STATS)
        d_x = torch.from_numpy(f_t).cuda()
        d_s = torch.empty(nfft, dtype=torch.float64, device="cuda")
        d_st = torch.zeros(3, dtype=torch.float64, device="cuda")
        for _ in range(20):
            plan.exec(d_x, d_s, scale=dt, d_stats=d_st, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            plan.exec(d_x, d_s, scale=dt, d_stats=d_st, stream=stream)
        e1.record(); e1.synchronize()
        t_dev = e0.elapsed_time(e1) / 50 * 1e-3
        t_host = best(lambda: plan.exec_host(f_t, scale=dt))
        kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
        fft_amd.fft_from_timeseries(f_t, times, **kw)
        t_full = best(lambda: fft_amd.fft_from_timeseries(f_t, times, **kw))
        t_ref_fft = best(lambda: np.abs(dt * np.fft.fft(f_t * np.hanning(n0), n=nfft)) ** 2)
        t_ref_full = best(lambda: so.fft_from_timeseries(f_t, times, **kw), reps=3)
        # the app's loop over the blocks m (final2.py:752-758): three single calls vs one batched call
        blocks = [so.return_amplitude_block([1, 1, 1], [1, 1], m, 0, times) for m in (0, 1, 2)]
        fft_amd.fft_from_timeseries_batch(blocks, times, **kw)
        t_three = best(lambda: [fft_amd.fft_from_timeseries(f, times, **kw) for f in blocks])
        t_batch = best(lambda: fft_amd.fft_from_timeseries_batch(blocks, times, **kw))
        print(f"N0={n0} Nfft={nfft}: three blocks, single calls {t_three*1e3:7.2f} ms ({t_three/3*1e3:6.2f} per block) | "
              f"one batched call {t_batch*1e3:7.2f} ms ({t_batch/3*1e3:6.2f} per block) = {t_three/t_batch:4.2f}x", flush=True)
        # the whole loop body: f(t) of each block AND its estimate.  Generator on the device, f(t) back to the host,
        # then the batched estimator -- against the resident pipeline, which never brings f(t) down
        K, J = [1.0, 1.0, 1.0], [1.0, 1.0]
        gen_then_batch = lambda: fft_amd.fft_from_timeseries_batch(                      # noqa: E731
            [fft_amd.return_amplitude_block_general(K, J, m, 0, times) for m in (0, 1, 2)], times, **kw)
        resident = lambda: fft_amd.fft_from_timeseries_blocks(K, J, [0, 1, 2], 0, times, **kw)   # noqa: E731
        resident()
        t_gen, t_res = best(gen_then_batch), best(resident)
        t_ref_loop = best(lambda: [so.fft_from_timeseries(so.return_amplitude_block(K, J, m, 0, times), times, **kw)
                                   for m in (0, 1, 2)], reps=2)
        print(f"N0={n0} Nfft={nfft}: loop body (f(t) + estimate) for three blocks: generator + batched estimator "
              f"{t_gen*1e3:7.2f} ms ({t_gen/3*1e3:6.2f} per block) | resident pipeline {t_res*1e3:7.2f} ms "
              f"({t_res/3*1e3:6.2f} per block) | the reference's numpy loop {t_ref_loop*1e3:8.1f} ms", flush=True)
        print(f"N0={n0} Nfft={nfft} launches={plan.launches}: "
              f"device plan {t_dev*1e6:8.1f} us | exec_host {t_host*1e3:7.3f} ms | "
              f"fft_from_timeseries GPU path {t_full*1e3:7.2f} ms vs numpy path {t_ref_full*1e3:7.2f} ms "
              f"(numpy window+fft+power alone {t_ref_fft*1e3:7.2f} ms)", flush=True)
        plan.destroy()


if __name__ == "__main__":
    main()
