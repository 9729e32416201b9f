#!/usr/bin/env python3
"""Per-pass cache-policy sweep for the multi-pass lengths (tuning aid):
    python tools/nt_pass_sweep.py [LOG2N] [BATCH] [DTYPE] [PASSES]
Builds one plan per combination of MFFT_TILE_NT_PASS (bit 0: non-temporal loads, bit 1:
non-temporal stores, per pass) and prints the best of 5 x 20 launches."""
import itertools, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 0
npass = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = 1 << log2n
torch.cuda.set_device(0)
rdt = torch.float32 if dtype == 0 else torch.float64
x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
warm = torch.randn(1 << 24, device="cuda")
for _ in range(3000): warm.mul_(1.0001)
torch.cuda.synchronize()
rows = []
for fto in (0,):
    for combo in itertools.product("0123", repeat=npass):
        os.environ["MFFT_TILE_NT_PASS"] = ",".join(combo)
        with fft_amd.Plan(n, batch=batch, dtype=dtype) as plan:
            for _ in range(20): plan.exec(x, y, stream=stream)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): plan.exec(x, y, stream=stream)
                e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        rows.append((round(best, 1), "".join(combo)))
        print(f"nt={''.join(combo)} {best:8.1f} us", flush=True)
rows.sort()
print("best:", rows[:8])
