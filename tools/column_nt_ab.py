#!/usr/bin/env python3
"""Cache policy of a column-batched plan's passes, alternated inside one process (MFFT_TILE_NT_PASS is read
when a plan is built; bit 0 = streaming loads, bit 1 = streaming stores, one value per pass):
    python tools/column_nt_ab.py LOG2N LOG2INNER DTYPE POLICY [POLICY ...]     ("default" = the library's choice,
"narrow" = the narrow tile width, MFFT_NARROW_COLUMNS, with the default policy)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd.engine import ColumnPlan
n, inner, dtype = 1 << int(sys.argv[1]), 1 << int(sys.argv[2]), int(sys.argv[3])
torch.cuda.set_device(0)
cdt = torch.complex128 if dtype else torch.complex64
x = torch.empty(n * inner, dtype=cdt, device="cuda")
torch.view_as_real(x).normal_()
y = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for pol in sys.argv[4:]:
        os.environ.pop("MFFT_TILE_NT_PASS", None)
        os.environ.pop("MFFT_NARROW_COLUMNS", None)
        if pol == "narrow":                                   # narrow tiles instead of a cache policy
            os.environ["MFFT_NARROW_COLUMNS"] = "1"
        elif pol != "default":
            os.environ["MFFT_TILE_NT_PASS"] = pol
        p = ColumnPlan(n, inner, 1, dtype, twiddle=(n * inner * 8, 3 * inner))
        for _ in range(2):
            p.exec(x, y, stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            p.exec(x, y, stream=s)
        e1.record(); e1.synchronize()
        res.setdefault(pol, []).append(e0.elapsed_time(e1) / 8 * 1e3)
        p.destroy()
print(f"n=2^{sys.argv[1]} inner=2^{sys.argv[2]} dtype={dtype}: " + " | ".join(f"{k} " + "/".join(f"{v:.0f}" for v in vs) for k, vs in res.items()), flush=True)
