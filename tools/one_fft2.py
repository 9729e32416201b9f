#!/usr/bin/env python3
"""Run the one-rank 2-D slab transform a few times (for rocprofv3): python tools/one_fft2.py [N] [LIBTAG]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
if len(sys.argv) > 2:
    fft_amd._lib.LIB_PATH = os.path.join(ROOT, "fft_amd", "lib", f"libmfft_{sys.argv[2]}.so")
from fft_amd import dist as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.cuda.set_device(0)
a = torch.view_as_complex(torch.randn((n * n, 2), dtype=torch.float32, device="cuda"))
comm, ops = fd.Comm(), fd.HipOps()
for _ in range(6):
    out = fd.fft2_slab(a, n, n, comm, ops, restore_layout=True)
torch.cuda.synchronize()
