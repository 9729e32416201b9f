#!/usr/bin/env python3
"""Throughput sweep over transform lengths on one GPU (plain complex -> complex).

    python tools/sweep.py [--dtype 0|1] [--points 26] [--lo 4 --hi 24]

For each n = 2^k the batch is chosen so that n*batch = 2^points.  Reports time per
launch sequence, GFLOP/s (5 n log2 n) and algorithmic GB/s (read + write once)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--points", type=int, default=26)
    ap.add_argument("--lo", type=int, default=4)
    ap.add_argument("--hi", type=int, default=24)
    ap.add_argument("--sizes", type=int, nargs="*", default=None, help="explicit lengths (any integer)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    cdt = torch.complex64 if args.dtype == 0 else torch.complex128
    rdt = torch.float32 if args.dtype == 0 else torch.float64
    esz = 8 if args.dtype == 0 else 16
    total = 1 << args.points
    sizes = args.sizes or [1 << k for k in range(args.lo, args.hi + 1)]
    stream = torch.cuda.current_stream().cuda_stream
    # keep the device busy first so clocks settle
    warm = torch.randn(1 << 24, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        warm.mul_(1.0001)
    torch.cuda.synchronize()
    print(f"# dtype={'c64' if args.dtype == 0 else 'c128'} points=2^{args.points}")
    print(f"{'n':>10s} {'batch':>8s} {'launches':>8s} {'us':>10s} {'GFLOP/s':>10s} {'GB/s(alg)':>10s}")
    for n in sizes:
        batch = max(1, total // n)
        x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
        y = torch.empty((batch, n), dtype=cdt, device="cuda")
        plan = fft_amd.Plan(n, batch=batch, dtype=args.dtype)
        for _ in range(20):
            plan.exec(x, y, stream=stream)
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.exec(x, y, stream=stream)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e-3)
        import math
        flops = 5.0 * n * math.log2(max(n, 2)) * batch
        byts = 2.0 * esz * n * batch
        print(f"{n:>10d} {batch:>8d} {plan.launches:>8d} {best*1e6:>10.1f} {flops/best/1e9:>10.0f} {byts/best/1e9:>10.0f}",
              flush=True)
        plan.destroy()
        del x, y


if __name__ == "__main__":
    main()
