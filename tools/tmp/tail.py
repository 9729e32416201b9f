import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fft_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libmfft_{sys.argv[1]}.so")
import fft_amd as fa
def timeit(p, x, y, warm=100, reps=400):
    global st
    for _ in range(warm): p.exec(x, y, d_stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): p.exec(x, y, d_stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
SET = ["none", "2,1", "4,1", "6,1", "8,1", "6,2", "10,2", "16,2", "16,4"]
for log2n, batch, flags in ((10, 65536, 3), (10, 65533, 3), (9, 131072, 3), (12, 16384, 3), (11, 32768, 3), (10, 65536, 19), (12, 16384, 19)):
    n = 1 << log2n
    x = torch.randn(batch, n, dtype=torch.complex64, device="cuda")
    y = torch.empty(batch, n, dtype=torch.float32 if flags & 2 else torch.complex64, device="cuda")
    y0 = torch.empty_like(y)
    st = torch.zeros(batch, 3, dtype=torch.float64, device='cuda') if flags & 16 else None
    plans = {}
    for o in SET:
        if o == "none": os.environ.pop("MFFT_LDS_TAIL", None)
        else: os.environ["MFFT_LDS_TAIL"] = o
        if flags == 0: os.environ["MFFT_LDS_GRID_OVER"] = "1"
        plans[o] = fa.Plan(n, batch=batch, dtype=0, flags=flags)
    os.environ.pop("MFFT_LDS_TAIL", None); os.environ.pop("MFFT_LDS_GRID_OVER", None)
    plans["none"].exec(x, y0, d_stats=st)
    same = []
    for o in SET[1:]:
        y.zero_(); plans[o].exec(x, y, d_stats=st); torch.cuda.synchronize()
        same.append(bool(torch.equal(torch.view_as_real(y) if y.is_complex() else y, torch.view_as_real(y0) if y0.is_complex() else y0)))
    res = {o: [] for o in SET}
    for r in range(3):
        for o in SET: res[o].append(timeit(plans[o], x, y))
    for p in plans.values(): p.destroy()
    print(f"2^{log2n} x {batch} flags {flags}: identical {all(same)}  " + "  ".join(f"{o}: {np.median(t):.1f}" for o, t in res.items()), flush=True)
