#!/usr/bin/env python3
"""LDS-DMA pipeline against the register-load form of the same kernels (MFFT_NO_DMA=1), one process,
interleaved rounds: parity against a float64 transform, then microseconds per launch.
    python tools/dma_ab.py [--points 26]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=26)
ap.add_argument("--sizes", default="14:0 13:0 13:1")
ap.add_argument("--lib", default="", help="variant library, e.g. fft_amd/lib/libmfft_nofft.so (diagnostic builds: errors are meaningless)")
a = ap.parse_args()
if a.lib:
    fft_amd._lib.LIB_PATH = os.path.abspath(a.lib)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
warm = torch.randn(1 << 24, device="cuda")
for _ in range(3000): warm.mul_(1.0001)
for item in a.sizes.split():
    k, dtype, flags = (list(map(int, item.split(":"))) + [0])[:3]        # log2 n : dtype : flags (1 = Hann, 2 = power)
    n = 1 << k
    batch = (1 << a.points) >> k
    rdt, cdt = (torch.float32, torch.complex64) if dtype == 0 else (torch.float64, torch.complex128)
    bufs = [(torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda")),
             torch.empty((batch, n), dtype=rdt if flags & 2 else cdt, device="cuda")) for _ in range(3)]       # > Infinity Cache in total
    win = torch.hann_window(n, periodic=False, dtype=torch.float64, device="cuda") if flags & 1 else None
    x, y = bufs[0]
    res = {}
    with fft_amd.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
        for tag, env in (("dma", None), ("regs", "1")):
            if env: os.environ["MFFT_NO_DMA"] = env
            else: os.environ.pop("MFFT_NO_DMA", None)
            y.zero_(); p.exec(x, y, stream=stream); torch.cuda.synchronize()
            rows = [0, 1, batch // 2, batch - 1] + list(range(7, batch, max(1, batch // 23)))
            err = 0.0
            for r in rows:
                xr = x[r].to(torch.complex128)
                ref = torch.fft.fft(xr * win if win is not None else xr)
                if flags & 2: ref = ref.real ** 2 + ref.imag ** 2
                err = max(err, (torch.linalg.vector_norm(y[r].to(ref.dtype) - ref) / torch.linalg.vector_norm(ref)).item())
            res[tag] = [err, 1e9]
        for rnd in range(6):
            for tag, env in (("dma", None), ("regs", "1")):
                if env: os.environ["MFFT_NO_DMA"] = env
                else: os.environ.pop("MFFT_NO_DMA", None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(3): p.exec(*bufs[i % 3], stream=stream)
                e0.record()
                for i in range(12): p.exec(*bufs[i % 3], stream=stream)
                e1.record(); e1.synchronize()
                res[tag][1] = min(res[tag][1], e0.elapsed_time(e1) / 12 * 1e3)
    os.environ.pop("MFFT_NO_DMA", None)
    lib = fft_amd._lib.load()
    if hasattr(lib, "mfft_diag_read"):             # diagnostic build: phase ticks of the last DMA launch
        import ctypes, numpy as np
        with fft_amd.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
            p.exec(x, y, stream=stream); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (16 * 256))()
        lib.mfft_diag_read(buf, 16 * 256)
        st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16)[:, :10].astype(float)
        trips = batch / 256
        names = ["read", "pass0", "exch0", "pass1", "exch1", "pass2", "exch2+dma", "pass3", "scale+stores", "wait"]
        print("   shader-clock ticks per trip, mean over workgroups: " + " | ".join(
            f"{nm} {v:.0f}" for nm, v in zip(names, st.mean(0) / trips)) + f" | total {st.sum(1).mean() / trips:.0f}")
    bytes_ = batch * n * (8 if dtype == 0 else 16) * (3 if flags & 2 else 4) // 2
    print(f"2^{k} x {batch} {'c64' if dtype == 0 else 'c128'} flags {flags}: " + " | ".join(
        f"{t}: {v[1]:7.1f} us {bytes_ / v[1] / 1e3:6.0f} GB/s err {v[0]:.1e}" for t, v in res.items()), flush=True)
    del bufs, x, y
