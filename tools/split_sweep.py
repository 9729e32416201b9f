#!/usr/bin/env python3
"""Sweep the radix split of the multi-pass lengths (tuning aid): for every 2^k in [LO, HI] time each
ordered split into 2..4 radices of 2^4..2^9 (spread at most 3 bits, fewest passes only) through
MFFT_SPLIT.  Data per launch sequence: 2^POINTS points.
    python tools/split_sweep.py [--dtype 0] [--lo 15 --hi 27] [--points 26]"""
import argparse, itertools, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", type=int, default=0); ap.add_argument("--lo", type=int, default=15)
ap.add_argument("--hi", type=int, default=27); ap.add_argument("--points", type=int, default=26)
a = ap.parse_args()
torch.cuda.set_device(0)
rdt = torch.float32 if a.dtype == 0 else torch.float64
stream = torch.cuda.current_stream().cuda_stream
warm = torch.randn(1 << 24, device="cuda")
for _ in range(3000): warm.mul_(1.0001)
for k in range(a.lo, a.hi + 1):
    n = 1 << k; batch = max(1, (1 << a.points) >> k)
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda")); y = torch.empty_like(x)
    npass = 2 if k <= 18 else (3 if k <= 27 else 4)
    res = []
    def run(tag):
        with fft_amd.Plan(n, batch=batch, dtype=a.dtype) as p:
            for _ in range(5): p.exec(x, y, stream=stream)
            torch.cuda.synchronize(); best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8): p.exec(x, y, stream=stream)
                e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / 8 * 1e3)
        res.append((round(best, 1), tag))
    os.environ.pop("MFFT_SPLIT", None); run("default")
    for c in itertools.product(range(4, 10), repeat=npass):
        if sum(c) != k or max(c) - min(c) > 3: continue
        os.environ["MFFT_SPLIT"] = ",".join(map(str, c)); run(os.environ["MFFT_SPLIT"])
    os.environ.pop("MFFT_SPLIT", None)
    d = res[0][0]; res.sort()
    print(f"2^{k} x {batch}: default {d} us | best " + "  ".join(f"{t}:{u}" for u, t in res[:5]), flush=True)
    del x, y
