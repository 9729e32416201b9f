#!/usr/bin/env python3
"""A/B timing of several builds of libmfft on one GPU, interleaved in one process.

    make -C fft_amd/csrc VARIANT=a KNOBS="-DMFFT_TW_SRC=2"     (repeat per variant)
    python tools/variant_bench.py [--n 1024 --batch 65536 --flags 3] a b c ...

Every variant runs the same plan on the same buffers; rounds are interleaved so
clock and thermal drift hit all of them alike.  Reports median / min time per
launch and the algorithmic HBM rate (8 B read + 4 or 8 B written per point)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import _lib as L  # noqa: E402  (prototypes only)


def load(tag):
    name = "libmfft.so" if tag == "default" else f"libmfft_{tag}.so"
    lib = ctypes.CDLL(os.path.join(ROOT, "fft_amd", "lib", name))
    for fn, (res, args) in L.SIGNATURES.items():
        if not hasattr(lib, fn):          # an older build of the library: entry points added since are absent
            continue
        getattr(lib, fn).restype = res
        getattr(lib, fn).argtypes = args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--flags", type=int, default=3)      # HANN | POWER
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--inner", type=int, default=10)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    n, batch = args.n, args.batch
    cdt = torch.complex64 if args.dtype == 0 else torch.complex128
    rdt = torch.float32 if args.dtype == 0 else torch.float64
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda", generator=g))
    power = bool(args.flags & 2)
    outs = {}
    libs, plans = {}, {}
    for tag in args.variants:
        lib = load(tag)
        h = ctypes.c_void_p()
        rc = lib.mfft_plan_1d(ctypes.byref(h), n, n, batch, args.dtype, args.flags)
        assert rc == 0, (tag, rc, lib.mfft_last_error())
        libs[tag], plans[tag] = lib, h
        outs[tag] = torch.empty((batch, n), dtype=rdt if power else cdt, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run(tag):
        rc = libs[tag].mfft_exec(plans[tag], x.data_ptr(), outs[tag].data_ptr(), 0.01, 0.0, 0.0, 1.0, None, stream)
        assert rc == 0, (tag, libs[tag].mfft_last_error())

    for tag in args.variants:
        for _ in range(5):
            run(tag)
    torch.cuda.synchronize()
    ref = outs[args.variants[0]].clone()
    for tag in args.variants[1:]:
        err = (torch.linalg.vector_norm((outs[tag] - ref).flatten().to(torch.complex128 if not power else torch.float64)) /
               torch.linalg.vector_norm(ref.flatten().to(torch.complex128 if not power else torch.float64))).item()
        assert err < 1e-5 or tag.startswith("z"), (tag, err)   # z* = diagnostic builds with wrong output
    times = {t: [] for t in args.variants}
    for _ in range(args.rounds):
        for tag in args.variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.inner):
                run(tag)
            e1.record()
            e1.synchronize()
            times[tag].append(e0.elapsed_time(e1) / args.inner)
    esz = 8 if args.dtype == 0 else 16
    bytes_alg = n * batch * (esz + (esz // 2 if power else esz))
    print(f"# n={n} batch={batch} dtype={args.dtype} flags={args.flags} bytes/launch={bytes_alg}")
    for tag in args.variants:
        ts = sorted(times[tag])
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"{tag:>12s}  median {med*1e3:8.2f} us  min {mn*1e3:8.2f} us   "
              f"{bytes_alg/med/1e6:8.1f} GB/s (median)  {bytes_alg/mn/1e6:8.1f} GB/s (best)", flush=True)


if __name__ == "__main__":
    main()
