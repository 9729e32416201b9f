#!/usr/bin/env python3
"""Per-workgroup durations of one batched single-kernel launch, from the timestamps a diagnostic build records
(make -C fft_amd/csrc VARIANT=stamps KNOBS="-DMFFT_DIAG_STAMPS=1"; complex64, register-load kernels):

    python tools/wg_durations.py [--log2n 10] [--batch 65536] [--flags 3] [--over N] [--tail pct,chunk]

Each workgroup stores s_memtime at its start and end plus XCC_ID / HW_ID of its first wave.  The counters of
different XCDs are not aligned, so only durations (end - start of one workgroup) are compared: their spread over CUs
and XCDs under one striding resident set is what several sets, or a dispatched tail, can win back.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmfft_stamps.so")
import fft_amd as fa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--flags", type=int, default=3)
    ap.add_argument("--over", default="1", help="MFFT_LDS_GRID_OVER (0 = the library's choice)")
    ap.add_argument("--tail", default="-1,1", help="MFFT_LDS_TAIL (default: none)")
    args = ap.parse_args()
    lib = _lib.load()
    n = 1 << args.log2n
    x = torch.randn(args.batch, n, dtype=torch.complex64, device="cuda")
    y = torch.empty(args.batch, n, dtype=torch.float32 if args.flags & 2 else torch.complex64, device="cuda")
    os.environ["MFFT_LDS_GRID_OVER"] = args.over
    os.environ["MFFT_LDS_TAIL"] = args.tail
    with fa.Plan(n, batch=args.batch, dtype=0, flags=args.flags) as p:
        for _ in range(200):
            p.exec(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            p.exec(x, y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        buf = (ctypes.c_ulonglong * (16 * 2048))()
        lib.mfft_diag_read(buf, 16 * 2048)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.int64)
    a = a[a[:, 1] > 0]
    dur = (a[:, 1] - a[:, 0]).astype(np.float64)
    xcc, hw = a[:, 2] & 15, a[:, 3]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    print(f"2^{args.log2n} x {args.batch}, flags {args.flags}, sets {args.over}, tail {args.tail}: {us:.1f} us per launch, "
          f"{len(a)} workgroups recorded (the first 2048 of the last launch)")
    print(f"  workgroup durations in s_memtime ticks: min {dur.min():.0f}  p10 {np.percentile(dur, 10):.0f}  median {np.median(dur):.0f}  "
          f"p90 {np.percentile(dur, 90):.0f}  max {dur.max():.0f}   (max / median {dur.max() / np.median(dur):.3f})")
    for k in np.unique(xcc):
        m = xcc == k
        print(f"  XCD {k}: {m.sum():4d} workgroups  median {np.median(dur[m]):7.0f}  max {dur[m].max():7.0f}")
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    per_cu = np.array([dur[key == k].mean() for k in np.unique(key)])
    print(f"  mean duration per CU ({len(per_cu)} CUs seen): min {per_cu.min():.0f}  median {np.median(per_cu):.0f}  max {per_cu.max():.0f}")


if __name__ == "__main__":
    main()
