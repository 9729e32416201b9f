#!/usr/bin/env python3
"""Whole-plan time of one complex128 transform under different pass splits (MFFT_SPLIT is read when the
plan is built):   python tools/split_c128_big.py LOG2N SPLIT [SPLIT ...]      ("default" = the library's choice)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
lg = int(sys.argv[1])
n = 1 << lg
torch.cuda.set_device(0)
x = torch.empty(n, dtype=torch.complex128, device="cuda")
xr = torch.view_as_real(x)
for lo in range(0, n, 1 << 27):
    xr[lo:lo + (1 << 27)].normal_()
y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
reps = max(2, min(20, (1 << 30) // n))
for sp in sys.argv[2:]:
    if sp == "default":
        os.environ.pop("MFFT_SPLIT", None)
    else:
        os.environ["MFFT_SPLIT"] = sp
    p = fft_amd.Plan(n, dtype=fft_amd.C128)
    p.exec(x, y, stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            p.exec(x, y, stream=s)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    print(f"2^{lg} complex128  split {sp:12s} launches {p.launches}  {best:9.3f} ms  {2*16*n/best/1e6:7.0f} GB/s algorithmic", flush=True)
    p.destroy()
