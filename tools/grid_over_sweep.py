#!/usr/bin/env python3
"""Resident sets of workgroups per launch (MFFT_LDS_GRID_OVER) against time, for every single-kernel length,
precision and variant: the measurement behind lds_grid_over() and the lane-distributed kernel's factors.

    python tools/grid_over_sweep.py [--out gpurun_out/over_sweep.json]

Each case runs 2^26 (complex64) / 2^25 (complex128) points per launch; one plan per factor, the factors
alternated over two rounds with 40 warm-up and 120 timed executions each (clocks settled), best round kept.
flags as in include/mfft.h: 0 plain, 1 Hann, 2 power, 3 Hann + power, 8 real input, 18 / 19 power (+ Hann) with
per-signal statistics.  "0" in the factor list is the library's own choice.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_amd as fa  # noqa: E402

OVERS = ["0", "1", "2", "4", "8", "16"]


def timeit(p, x, y, st, warm=40, reps=120):
    for _ in range(warm):
        p.exec(x, y, d_stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        p.exec(x, y, d_stats=st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--min-log2n", type=int, default=4)
    ap.add_argument("--max-log2n", type=int, default=14)
    args = ap.parse_args()
    out = {}
    for dtype in (0, 1):
        cdt = torch.complex64 if dtype == 0 else torch.complex128
        rdt = torch.float32 if dtype == 0 else torch.float64
        for log2n in range(args.min_log2n, min(args.max_log2n, 14 if dtype == 0 else 13) + 1):
            n = 1 << log2n
            batch = (1 << (26 if dtype == 0 else 25)) >> log2n
            for flags in (0, 1, 2, 3, 8, 18, 19):
                x = torch.randn(batch, n, dtype=rdt if flags & 8 else cdt, device="cuda")
                y = torch.empty(batch, n, dtype=rdt if flags & 2 else cdt, device="cuda")
                st = torch.zeros(batch, 3, dtype=torch.float64, device="cuda") if flags & 16 else None
                plans = {}
                for o in OVERS:
                    os.environ["MFFT_LDS_GRID_OVER"] = o
                    plans[o] = fa.Plan(n, batch=batch, dtype=dtype, flags=flags)
                os.environ.pop("MFFT_LDS_GRID_OVER", None)
                res = {o: [] for o in OVERS}
                timeit(plans["1"], x, y, st)              # clocks back up after the plan builds: not counted
                for _ in range(2):
                    for o in OVERS:
                        res[o].append(timeit(plans[o], x, y, st))
                for p in plans.values():
                    p.destroy()
                best_of = {o: float(np.min(t)) for o, t in res.items()}
                best = min((o for o in OVERS if o != "0"), key=best_of.get)
                out[f"{dtype}/{log2n}/{flags}"] = best_of
                print(f"dtype {dtype} 2^{log2n:2d} flags {flags:2d}: " + "  ".join(f"x{o}: {best_of[o]:6.1f}" for o in OVERS) +
                      f"   best x{best} ({100 * (best_of['1'] / best_of[best] - 1):.1f} % on x1; library's choice "
                      f"{100 * (best_of['1'] / best_of['0'] - 1):+.1f} %)", flush=True)
                del x, y, st
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
