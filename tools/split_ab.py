#!/usr/bin/env python3
"""Whole-plan time of batched power-of-two transforms under different pass splits, alternated inside one
process (MFFT_SPLIT is read when a plan is built):
    python tools/split_ab.py DTYPE LOG2N BATCH[,BATCH...] SPLIT [SPLIT ...]     (DTYPE 0 complex64, 1 complex128;
                                                                                  "default" = the library's choice)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd
dt, lg = int(sys.argv[1]), int(sys.argv[2])
batches = [int(v) for v in sys.argv[3].split(",")]
splits = sys.argv[4:]
n = 1 << lg
torch.cuda.set_device(0)
cdt = torch.complex64 if dt == 0 else torch.complex128
s = torch.cuda.current_stream().cuda_stream
print(f"# 2^{lg} {'complex64' if dt == 0 else 'complex128'}: us per transform of the whole batch, median of 7 rounds")
print(f"{'batch':>7s} " + " ".join(f"{sp:>12s}" for sp in splits))
for batch in batches:
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float32 if dt == 0 else torch.float64, device="cuda"))
    y = torch.empty_like(x)
    plans = {}
    for sp in splits:
        if sp == "default":
            os.environ.pop("MFFT_SPLIT", None)
        else:
            os.environ["MFFT_SPLIT"] = sp
        plans[sp] = fft_amd.Plan(n, batch=batch, dtype=fft_amd.C64 if dt == 0 else fft_amd.C128)
        for _ in range(5):
            plans[sp].exec(x, y, stream=s)
    os.environ.pop("MFFT_SPLIT", None)
    torch.cuda.synchronize()
    reps = max(3, min(50, (1 << 28) // (n * batch)))
    res = {sp: [] for sp in splits}
    for _ in range(7):
        for sp in splits:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                plans[sp].exec(x, y, stream=s)
            e1.record(); e1.synchronize()
            res[sp].append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{batch:>7d} " + " ".join(f"{sorted(res[sp])[3]:>12.1f}" for sp in splits), flush=True)
    for p in plans.values():
        p.destroy()
    del x, y
