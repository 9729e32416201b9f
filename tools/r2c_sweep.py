#!/usr/bin/env python3
"""Real-input transforms above one LDS residency: the packed route (n/2 complex samples, the split into the n bins fused
into the last pass) against the same route with a separate split kernel (MFFT_R2C_SPLIT_KERNEL=1), against the route
that promotes the reals to complex (MFFT_NO_R2C=1; both read when the plan is built) and against the complex transform
of the same length, alternated in one process on device buffers.
    python tools/r2c_sweep.py [--dtype 1] [--log2n 15 16 ... 24] [--flags 0]      (flags: 3 = Hann + power)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", type=int, default=1)
    ap.add_argument("--log2n", type=int, nargs="*", default=list(range(15, 25)))
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--points", type=int, default=1 << 24, help="total points per launch (sets the batch)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    rdt = torch.float32 if args.dtype == 0 else torch.float64
    stream = torch.cuda.current_stream().cuda_stream
    power = bool(args.flags & fft_amd.POWER)
    for lg in args.log2n:
        n = 1 << lg
        batch = max(1, args.points // n)
        xr = torch.randn((batch, n), dtype=rdt, device="cuda")
        xc = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
        out = torch.empty((batch, n), dtype=rdt if power else xc.dtype, device="cuda")
        plans = {}
        os.environ.pop("MFFT_NO_R2C", None)
        plans["packed"] = (fft_amd.Plan(n, batch=batch, dtype=args.dtype, flags=args.flags | fft_amd.REAL_IN), xr)
        os.environ["MFFT_R2C_SPLIT_KERNEL"] = "1"
        plans["split"] = (fft_amd.Plan(n, batch=batch, dtype=args.dtype, flags=args.flags | fft_amd.REAL_IN), xr)
        os.environ.pop("MFFT_R2C_SPLIT_KERNEL")
        os.environ["MFFT_NO_R2C"] = "1"
        plans["promoted"] = (fft_amd.Plan(n, batch=batch, dtype=args.dtype, flags=args.flags | fft_amd.REAL_IN), xr)
        os.environ.pop("MFFT_NO_R2C")
        plans["complex"] = (fft_amd.Plan(n, batch=batch, dtype=args.dtype, flags=args.flags), xc)
        times = {k: [] for k in plans}
        for _ in range(9):
            for k, (p, x) in plans.items():
                p.exec(x, out, stream=stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    p.exec(x, out, stream=stream)
                e1.record(); e1.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print(f"2^{lg} x {batch} {'complex64' if args.dtype == 0 else 'complex128'} flags {args.flags}: real input packed "
              f"{med['packed']:9.1f} us (launches {plans['packed'][0].launches}) | separate split kernel {med['split']:9.1f} us | promoted {med['promoted']:9.1f} us "
              f"(launches {plans['promoted'][0].launches}) | complex input {med['complex']:9.1f} us   packed/promoted "
              f"{med['packed'] / med['promoted']:.2f}  packed/complex {med['packed'] / med['complex']:.2f}", flush=True)
        for p, _ in plans.values():
            p.destroy()


if __name__ == "__main__":
    main()
