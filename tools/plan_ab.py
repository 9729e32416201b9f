#!/usr/bin/env python3
"""A/B of ONE plan configuration under several settings of a variable the library reads when a plan is built
(or executed), alternated round by round in one process on the same device buffers.

    python tools/plan_ab.py --log2n 24 --env MFFT_SPLIT --values default 8,8,8 12,12
    python tools/plan_ab.py --log2n 30 --dtype 1 --env MFFT_TILE_NT_PASS --values default 3,3,3 3,3,1
    python tools/plan_ab.py --log2n 18 --batch 256 --dtype 1 --flags 3 --env MFFT_TABLE_WINDOW --values default 1
    python tools/plan_ab.py --log2n 22 --dtype 1 --flags 8 --env MFFT_NO_R2C --values default 1
    python tools/plan_ab.py --log2n 14 --inner 2048 --env MFFT_BALANCED_COLUMNS --values default 1      (column-batched)
    python tools/plan_ab.py --log2n 18 --inner 1024 --dtype 1 --twiddle 2147483648 0 --env MFFT_TILE_NT --values default 0 1 3

Variables: MFFT_SPLIT (log2 radices of the global passes), MFFT_TILE_NT / MFFT_TILE_NT_PASS (cache policy per pass:
bit 0 streaming loads, bit 1 streaming stores), MFFT_FOUR_PASSES, MFFT_NO_NARROW_RADIX, MFFT_BALANCED_COLUMNS,
MFFT_NARROW_COLUMNS, MFFT_WORK2, MFFT_TABLE_WINDOW, MFFT_NO_R2C, MFFT_NO_WAVE, MFFT_TILE_GRID_OVER / MFFT_LDS_GRID_OVER (resident sets of workgroups per tile pass / single-kernel transform), MFFT_LDS_TAIL (pct,chunk), MFFT_TILE_BUF (0 global accesses, 4 buffer-form
loads, 12 and stores), and -- read at execution --
MFFT_NO_DMA, MFFT_GENERAL_PROLOGUE.
"default" leaves the variable unset.  --n-in zero-pads; flags as in include/mfft.h (1 Hann, 2 power, 8 real input,
16 statistics).  Reports the median and best time per execution and the algorithmic HBM rate (one read of the
input + one write of the output)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402
from fft_amd.engine import ColumnPlan  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--log2n", type=int, required=True)
    ap.add_argument("--n-in", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--inner", type=int, default=0, help="> 0: column-batched plan over (batch, n, inner)")
    ap.add_argument("--twiddle", type=int, nargs=2, default=None, metavar=("N_TOTAL", "C0"))
    ap.add_argument("--env", required=True)
    ap.add_argument("--values", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--inner-reps", type=int, default=0, help="executions per timed bracket (0: by size)")
    ap.add_argument("--lib", default="", help="A/B build of the library to load (make VARIANT=<tag>: fft_amd/lib/libmfft_<tag>.so)")
    args = ap.parse_args()
    if args.lib:
        from fft_amd import _lib
        _lib.LIB_PATH = os.path.join(ROOT, "fft_amd", "lib", f"libmfft_{args.lib}.so")
    torch.cuda.set_device(0)
    n = 1 << args.log2n
    n_in = args.n_in or n
    rdt = torch.float32 if args.dtype == 0 else torch.float64
    cdt = torch.complex64 if args.dtype == 0 else torch.complex128
    real_in, power, stats = bool(args.flags & 8), bool(args.flags & 2), bool(args.flags & 16)
    inner = max(1, args.inner)
    count_in, count_out = args.batch * n_in * inner, args.batch * n * inner

    def filled(count, dt):
        t = torch.empty(count, dtype=dt, device="cuda")
        flat = torch.view_as_real(t).reshape(-1) if t.is_complex() else t
        for lo in range(0, flat.numel(), 1 << 27):
            flat[lo:lo + (1 << 27)].normal_()
        return t
    x = filled(count_in, rdt if real_in else cdt)
    y = torch.empty(count_out, dtype=rdt if power else cdt, device="cuda")
    d_stats = torch.zeros(3 * args.batch, dtype=torch.float64, device="cuda") if stats else None
    stream = torch.cuda.current_stream().cuda_stream
    reps = args.inner_reps or max(1, min(20, (1 << 28) // max(count_in, count_out)))
    plans = {}
    for v in args.values:
        if v == "default":
            os.environ.pop(args.env, None)
        else:
            os.environ[args.env] = v
        if args.inner:
            plans[v] = ColumnPlan(n, inner, batch=args.batch, dtype=args.dtype, twiddle=tuple(args.twiddle) if args.twiddle else None)
        else:
            plans[v] = fft_amd.Plan(n, n_in=n_in, batch=args.batch, dtype=args.dtype, flags=args.flags)
    times = {v: [] for v in args.values}

    def run(v):
        if v == "default":
            os.environ.pop(args.env, None)
        else:
            os.environ[args.env] = v                 # variables read at execution (MFFT_NO_DMA)
        if args.inner:
            plans[v].exec(x, y, stream=stream)
        else:
            plans[v].exec(x, y, scale=1.0, d_stats=d_stats, stream=stream)
    for v in args.values:
        run(v)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for v in args.values:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(v)
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps)
    os.environ.pop(args.env, None)
    esz = 8 if args.dtype == 0 else 16
    alg = count_in * (esz // 2 if real_in else esz) + count_out * (esz // 2 if power else esz)
    what = f"2^{args.log2n}" + (f" from {n_in} samples" if n_in != n else "") + (f" x {inner} columns" if args.inner else "") + \
           f" x {args.batch}, {'complex64' if args.dtype == 0 else 'complex128'}, flags {args.flags}"
    for v in args.values:
        ts = sorted(times[v])
        med, best = ts[len(ts) // 2], ts[0]
        print(f"{what}  {args.env}={v:12s} launches {plans[v].launches}  median {med * 1e3:10.1f} us  best {best * 1e3:10.1f} us  "
              f"{alg / med / 1e6:7.0f} GB/s algorithmic", flush=True)
    for p in plans.values():
        p.destroy()


if __name__ == "__main__":
    main()
