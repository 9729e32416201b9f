#!/usr/bin/env python3
"""Column-batched plan with and without the fused four-step twiddle on the same shape (per-kernel times
come from rocprofv3 around this script):  python tools/column_twiddle_ab.py LOG2N LOG2INNER [DTYPE]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
from fft_amd.engine import ColumnPlan
n, inner = 1 << int(sys.argv[1]), 1 << int(sys.argv[2])
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.cuda.set_device(0)
cdt = torch.complex128 if dtype else torch.complex64
x = torch.empty(n * inner, dtype=cdt, device="cuda")
torch.view_as_real(x).normal_()
y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for tw in (None, (n * inner * 8, 3 * inner)):
    p = ColumnPlan(n, inner, 1, dtype, twiddle=tw)
    for _ in range(3):
        p.exec(x, y, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        p.exec(x, y, stream=s)
    e1.record(); e1.synchronize()
    print(f"n=2^{sys.argv[1]} inner=2^{sys.argv[2]} dtype={dtype} twiddle={'yes' if tw else 'no '}: {e0.elapsed_time(e1)/10*1e3:8.1f} us per transform", flush=True)
    p.destroy()
