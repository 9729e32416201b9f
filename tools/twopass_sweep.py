#!/usr/bin/env python3
"""Two-pass plans with radices up to 4096 (four-column tiles, XCD-aware tile order) against the
default split, complex64: correctness against a float64 transform of the same input, then time.
    python tools/twopass_sweep.py [--lo 19 --hi 24] [--points 24] [--splits "12,12 11,13"]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd

ap = argparse.ArgumentParser()
ap.add_argument("--lo", type=int, default=19); ap.add_argument("--hi", type=int, default=24)
ap.add_argument("--points", type=int, default=24)
ap.add_argument("--splits", default="", help="space-separated MFFT_SPLIT values tried for every length they fit")
a = ap.parse_args()
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
warm = torch.randn(1 << 24, device="cuda")
for _ in range(3000): warm.mul_(1.0001)
for k in range(a.lo, a.hi + 1):
    n = 1 << k; batch = max(1, (1 << a.points) >> k)
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float32, device="cuda")); y = torch.empty_like(x)
    ref = torch.fft.fft(x[0].to(torch.complex128))
    cands = ["default"] + [f"{g0},{k - g0}" for g0 in range(max(10, k - 12), min(12, k - 9) + 1) if k - g0 >= 9]
    cands += [s for s in a.splits.split() if sum(map(int, s.split(","))) == k]
    out = []
    for tag in cands:
        if tag == "default": os.environ.pop("MFFT_SPLIT", None)
        else: os.environ["MFFT_SPLIT"] = tag
        try:
            with fft_amd.Plan(n, batch=batch, dtype=0) as p:
                launches = p.launches
                y.zero_()
                for _ in range(5): p.exec(x, y, stream=stream)
                torch.cuda.synchronize()
                err = (torch.linalg.vector_norm(y[0].to(torch.complex128) - ref) / torch.linalg.vector_norm(ref)).item()
                errl = (torch.linalg.vector_norm(y[-1].to(torch.complex128) - torch.fft.fft(x[-1].to(torch.complex128))) / torch.linalg.vector_norm(ref)).item()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10): p.exec(x, y, stream=stream)
                    e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            out.append(f"{tag}[{launches}]: {best:7.1f} us err {max(err, errl):.1e}")
        except Exception as exc:                                       # noqa: BLE001
            out.append(f"{tag}: {str(exc)[:60]}")
    os.environ.pop("MFFT_SPLIT", None)
    print(f"2^{k} x {batch}: " + " | ".join(out), flush=True)
    del x, y
