#!/usr/bin/env python3
"""One plan executed launch by launch on a stream against the same execution recorded once into a graph
(torch.cuda.graph around mfft_exec) and replayed: time per execution, back to back.
    python tools/graph_bench.py [--n 256032 --n-in 8001 --batch 1 --dtype 1 --flags 19]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fft_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256032)
    ap.add_argument("--n-in", type=int, default=8001)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", type=int, default=1)
    ap.add_argument("--flags", type=int, default=19)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    rdt = torch.float32 if args.dtype == 0 else torch.float64
    cdt = torch.complex64 if args.dtype == 0 else torch.complex128
    power, stats = bool(args.flags & 2), bool(args.flags & 16)
    x = torch.view_as_complex(torch.randn((args.batch, args.n_in, 2), dtype=rdt, device="cuda"))
    y = torch.empty((args.batch, args.n), dtype=rdt if power else cdt, device="cuda")
    d_stats = torch.zeros(3 * args.batch, dtype=torch.float64, device="cuda") if stats else None
    p = fft_amd.Plan(args.n, n_in=args.n_in, batch=args.batch, dtype=args.dtype, flags=args.flags)
    cur = torch.cuda.current_stream().cuda_stream
    p.exec(x, y, d_stats=d_stats, stream=cur)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            p.exec(x, y, d_stats=d_stats, stream=side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)

    def timed(fn):
        best = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            e1.synchronize()
            best.append(e0.elapsed_time(e1) / args.reps * 1e3)
        return sorted(best)[len(best) // 2]
    t_stream = timed(lambda: p.exec(x, y, d_stats=d_stats, stream=cur))
    t_graph = timed(g.replay)
    print(f"n={args.n} from {args.n_in} x {args.batch} dtype {args.dtype} flags {args.flags} launches {p.launches}: "
          f"stream {t_stream:7.1f} us   graph replay {t_graph:7.1f} us", flush=True)
    del g
    p.destroy()


if __name__ == "__main__":
    main()
