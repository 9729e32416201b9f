#!/usr/bin/env python3
"""Sweep several builds of libmfft over lengths (plain complex64/128 transforms), interleaved.
    python tools/sweep_variants.py --tags default ns --lo 6 --hi 14 [--dtype 0] [--flags 0]"""
import argparse, ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import _lib as L

def load(tag):
    L._share_hip_runtime_with_torch()
    lib = ctypes.CDLL(os.path.join(ROOT, "fft_amd", "lib", "libmfft.so" if tag == "default" else f"libmfft_{tag}.so"))
    for fn, (res, args) in L.SIGNATURES.items():
        getattr(lib, fn).restype = res; getattr(lib, fn).argtypes = args
    return lib

ap = argparse.ArgumentParser()
ap.add_argument("--tags", nargs="+", default=["default"])
ap.add_argument("--lo", type=int, default=4); ap.add_argument("--hi", type=int, default=14)
ap.add_argument("--dtype", type=int, default=0); ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--points", type=int, default=26)
a = ap.parse_args()
torch.cuda.set_device(0)
libs = {t: load(t) for t in a.tags}
rdt = torch.float32 if a.dtype == 0 else torch.float64
esz = 8 if a.dtype == 0 else 16
power = bool(a.flags & 2)
stream = torch.cuda.current_stream().cuda_stream
print(f"# dtype={a.dtype} flags={a.flags}; GB/s algorithmic (median of 7 rounds x 10 launches)")
print(f"{'n':>9s} " + " ".join(f"{t:>12s}" for t in a.tags))
for k in range(a.lo, a.hi + 1):
    n = 1 << k; batch = max(1, (1 << a.points) // n)
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
    y = torch.empty((batch, n), dtype=rdt if power else x.dtype, device="cuda")
    plans = {}
    st = torch.zeros((batch, 3), dtype=torch.float64, device="cuda") if a.flags & 16 else None
    stp = st.data_ptr() if st is not None else None
    for t, lib in libs.items():
        h = ctypes.c_void_p()
        assert lib.mfft_plan_1d(ctypes.byref(h), n, n, batch, a.dtype, a.flags) == 0, lib.mfft_last_error()
        plans[t] = h
        for _ in range(10):
            lib.mfft_exec(h, x.data_ptr(), y.data_ptr(), 1.0, 0.0, 0.0, 1.0, stp, stream)
    torch.cuda.synchronize()
    res = {t: [] for t in a.tags}
    for _ in range(7):
        for t, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.mfft_exec(plans[t], x.data_ptr(), y.data_ptr(), 1.0, 0.0, 0.0, 1.0, stp, stream)
            e1.record(); e1.synchronize()
            res[t].append(e0.elapsed_time(e1) / 10 * 1e-3)
    byts = n * batch * (esz + (esz // 2 if power else esz))
    print(f"{n:>9d} " + " ".join(f"{byts / sorted(res[t])[3] / 1e9:>12.0f}" for t in a.tags), flush=True)
    for t, lib in libs.items():
        lib.mfft_plan_destroy(plans[t])
    del x, y
