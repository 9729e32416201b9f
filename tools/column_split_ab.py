#!/usr/bin/env python3
"""Pass splits of a column-batched plan, alternated inside one process (MFFT_SPLIT is read when a plan is built):
    python tools/column_split_ab.py LOG2N LOG2INNER DTYPE SPLIT [SPLIT ...]        ("default" = the library's choice)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd.engine import ColumnPlan
lg, li, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n, inner = 1 << lg, 1 << li
torch.cuda.set_device(0)
cdt = torch.complex128 if dtype else torch.complex64
x = torch.empty(n * inner, dtype=cdt, device="cuda"); torch.view_as_real(x).normal_()
y = torch.empty_like(x); s = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for sp in sys.argv[4:]:
        if sp == "default": os.environ.pop("MFFT_SPLIT", None)
        else: os.environ["MFFT_SPLIT"] = sp
        try:
            p = ColumnPlan(n, inner, 1, dtype)
        except Exception as e:
            res.setdefault(sp, []).append(float("nan")); continue
        for _ in range(3): p.exec(x, y, stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): p.exec(x, y, stream=s)
        e1.record(); e1.synchronize()
        res.setdefault(sp, []).append(e0.elapsed_time(e1) / 10 * 1e3); p.destroy()
print(f"columns 2^{lg} x 2^{li} dtype={dtype}: " + " | ".join(f"{k} " + "/".join(f"{v:.0f}" for v in vs) for k, vs in res.items()), flush=True)
