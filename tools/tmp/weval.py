import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fft_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libmfft_{sys.argv[1]}.so")
import fft_amd as fa
def timeit(p, x, y, st, warm=100, reps=400):
    for _ in range(warm): p.exec(x, y, d_stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): p.exec(x, y, d_stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
OV = ["1", "2", "4", "8", "16"]
for log2n, flags in ((10, 3), (9, 3), (11, 3), (12, 3), (10, 1), (10, 19), (9, 19), (11, 19), (12, 19), (12, 1)):
    n = 1 << log2n; batch = (1 << 26) >> log2n
    x = torch.randn(batch, n, dtype=torch.complex64, device="cuda")
    y = torch.empty(batch, n, dtype=torch.float32 if flags & 2 else torch.complex64, device="cuda")
    st = torch.zeros(batch, 3, dtype=torch.float64, device="cuda") if flags & 16 else None
    plans = {}
    for o in OV:
        os.environ["MFFT_LDS_GRID_OVER"] = o
        plans[o] = fa.Plan(n, batch=batch, dtype=0, flags=flags)
    os.environ.pop("MFFT_LDS_GRID_OVER", None)
    plans["1"].exec(x, y, d_stats=st); torch.cuda.synchronize()
    xs = x[:64].cpu().numpy().astype(np.complex128)
    ref = np.fft.fft(xs * np.hanning(n), axis=-1)
    ref = np.abs(ref) ** 2 if flags & 2 else ref
    got = y[:64].cpu().numpy()
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    mx = np.max(np.abs(got - ref)) / np.sqrt(np.mean(np.abs(ref) ** 2))
    res = {o: [] for o in OV}
    for r in range(3):
        for o in OV: res[o].append(timeit(plans[o], x, y, st))
    for p in plans.values(): p.destroy()
    print(f"2^{log2n} flags {flags}: rel l2 {err:.2e} max/rms {mx:.2e}  " + "  ".join(f"x{o}: {np.median(t):.1f}" for o, t in res.items()), flush=True)
