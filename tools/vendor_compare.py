#!/usr/bin/env python3
"""Time libmfft beside the vendor library (rocFFT, reached through torch.fft) on one GPU.

    python tools/vendor_compare.py [--cases cfg2 cfg3 cfg4 big1d sweep]

A positioning measurement only: torch.fft is never on the product path.  Each line gives the
time per transform for both engines on the same device buffers (events on torch's current
stream, best of 5 groups) and the algorithmic GB/s (input read once + output written once)."""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402
from fft_amd import dist as fd  # noqa: E402


def best_of(fn, inner, groups=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / inner)
    return best


def report(name, t_ours, t_vendor, flops, byts):
    print(f"{name:44s} ours {t_ours*1e6:10.1f} us {flops/t_ours/1e9:8.0f} GFLOP/s {byts/t_ours/1e9:6.0f} GB/s | "
          f"rocFFT {t_vendor*1e6:10.1f} us {byts/t_vendor/1e9:6.0f} GB/s | x{t_vendor/t_ours:5.2f}", flush=True)


def settle():
    warm = torch.randn(1 << 24, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        warm.mul_(1.0001)
    torch.cuda.synchronize()


def batched(n, batch, dtype, power=False):
    cdt = torch.complex64 if dtype == 0 else torch.complex128
    rdt = torch.float32 if dtype == 0 else torch.float64
    esz = 8 if dtype == 0 else 16
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
    stream = torch.cuda.current_stream().cuda_stream
    inner = max(3, min(50, int(2e9 / (n * batch * esz))))
    if power:
        y = torch.empty((batch, n), dtype=rdt, device="cuda")
        win = torch.hann_window(n, periodic=False, dtype=rdt, device="cuda")
        plan = fft_amd.Plan(n, batch=batch, dtype=dtype, flags=fft_amd.HANN | fft_amd.POWER)
        t_o = best_of(lambda: plan.exec(x, y, scale=0.01, stream=stream), inner)

        def vendor():
            f = torch.fft.fft(x * win)
            return (f.real * f.real + f.imag * f.imag) * 1e-4
        t_v = best_of(vendor, inner)
        byts = (esz + esz // 2) * n * batch
        name = f"{n} x {batch} c{esz*8} hann+power"
    else:
        y = torch.empty((batch, n), dtype=cdt, device="cuda")
        plan = fft_amd.Plan(n, batch=batch, dtype=dtype)
        t_o = best_of(lambda: plan.exec(x, y, stream=stream), inner)
        t_v = best_of(lambda: torch.fft.fft(x, out=y), inner)
        byts = 2 * esz * n * batch
        name = f"{n} x {batch} c{esz*8}"
    report(name, t_o, t_v, 5.0 * n * math.log2(n) * batch, byts)
    plan.destroy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="*", default=["cfg2", "cfg3", "cfg4", "big1d", "sweep"])
    args = ap.parse_args()
    torch.cuda.set_device(0)
    settle()
    if "cfg2" in args.cases:
        batched(1024, 65536, 0, power=True)
        batched(1024, 65536, 0)
    if "cfg3" in args.cases:
        batched(1 << 24, 1, 0)
        batched(1 << 24, 4, 0)
    if "sweep" in args.cases:
        for lg in (8, 10, 12, 13, 14, 16, 18, 20, 22):
            batched(1 << lg, 1 << (26 - lg), 0)
        for lg in (10, 12, 13, 16, 20):
            batched(1 << lg, 1 << (25 - lg), 1)
    if "cfg4" in args.cases:
        n = 16384
        a = torch.view_as_complex(torch.randn((n * n, 2), dtype=torch.float32, device="cuda"))
        comm, ops = fd.Comm(), fd.HipOps()
        t_o = best_of(lambda: fd.fft2_slab(a, n, n, comm, ops, restore_layout=True), 3)
        a2 = a.view(n, n)
        t_v = best_of(lambda: torch.fft.fft2(a2), 3)
        report(f"fft2 {n} x {n} c64 (natural layout)", t_o, t_v, 5.0 * n * n * 2 * math.log2(n), 16.0 * n * n)
        ops.close()
        del a, a2
    if "big1d" in args.cases:
        for lg in (28,):
            n1, n2 = 1 << ((lg + 1) // 2), 1 << (lg // 2)
            x = torch.view_as_complex(torch.randn((n1 * n2, 2), dtype=torch.float64, device="cuda"))
            comm, ops = fd.Comm(), fd.HipOps()
            t_o = best_of(lambda: fd.fft1d_fourstep(x, n1, n2, comm, ops, natural_order=True), 3)
            t_v = best_of(lambda: torch.fft.fft(x), 3)
            report(f"fft1d 2^{lg} c128 four-step (natural order)", t_o, t_v, 5.0 * n1 * n2 * lg, 32.0 * n1 * n2)
            plan = fft_amd.Plan(n1 * n2, dtype=1)
            y = torch.empty_like(x)
            stream = torch.cuda.current_stream().cuda_stream
            t_p = best_of(lambda: plan.exec(x, y, stream=stream), 3)
            report(f"fft1d 2^{lg} c128 single plan", t_p, t_v, 5.0 * n1 * n2 * lg, 32.0 * n1 * n2)
            plan.destroy()
            ops.close()


if __name__ == "__main__":
    main()
