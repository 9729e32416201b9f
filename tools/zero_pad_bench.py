#!/usr/bin/env python3
"""Rates of zero-padded transforms (n_in < n) beside the dense ones, both precisions:   python tools/zero_pad_bench.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, fft_amd
torch.cuda.set_device(0)
s = torch.cuda.current_stream().cuda_stream
for dtype, cdt in ((0, torch.complex64), (1, torch.complex128)):
    for n, n_in in ((64, 64), (64, 40), (256, 256), (256, 128), (256, 100), (512, 300), (1024, 1024), (1024, 512), (1024, 700), (4096, 2048), (4096, 1501)):
        batch = (1 << 26) // n
        x = torch.empty((batch, n_in), dtype=cdt, device="cuda"); torch.view_as_real(x).normal_()
        y = torch.empty((batch, n), dtype=cdt, device="cuda")
        p = fft_amd.Plan(n, n_in=n_in, batch=batch, dtype=dtype)
        for _ in range(3): p.exec(x, y, stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): p.exec(x, y, stream=s)
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        esz = 8 if dtype == 0 else 16
        print(f"dtype={dtype} n={n:5d} n_in={n_in:5d}: {t*1e6:8.1f} us  {(batch*(n+n_in)*esz)/t/1e9:7.0f} GB/s moved", flush=True)
        p.destroy(); del x, y
