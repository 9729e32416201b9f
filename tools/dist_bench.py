#!/usr/bin/env python3
"""Time the 2-D slab and 1-D four-step decompositions (fft_amd/dist.py) on the ranks given.

    python tools/dist_bench.py                                  # one GPU: local work only
    python -m torch.distributed.run --nproc-per-node 8 tools/dist_bench.py    # RCCL all-to-alls

BASELINE config 4: 16384 x 16384 complex64; config 5: 2^31 complex128 over 8 GPUs (here the
1-D length is scaled to what the ranks present can hold: 2^28 per 1 GPU)."""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import dist as fd  # noqa: E402


def timeit(fn, comm, reps=5):
    import torch.distributed as dist
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        if comm.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if comm.world > 1:
            dist.barrier()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n2d", type=int, default=16384)
    ap.add_argument("--log2n1d", type=int, default=0, help="0 = 28 + log2(world), capped at 31")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    comm, ops = fd.Comm(), fd.HipOps()
    P = comm.world
    # ---- 2-D, complex64 ----------------------------------------------------------------
    n = args.n2d
    rows = n // P
    a = torch.view_as_complex(torch.randn((rows * n, 2), dtype=torch.float32, device="cuda"))
    for restore in (False, True):
        t = timeit(lambda: fd.fft2_slab(a, n, n, comm, ops, restore_layout=restore), comm)
        flops = 5.0 * n * n * 2 * math.log2(n)
        if comm.rank == 0:
            print(f"fft2 {n}x{n} c64 ranks={P} restore_layout={restore}: {t*1e3:8.2f} ms  {flops/t/1e9:9.0f} GFLOP/s",
                  flush=True)
    del a
    # ---- 1-D four-step, complex128 -------------------------------------------------------
    lg = args.log2n1d or min(31, 28 + int(math.log2(P)))
    n1, n2 = fd.fourstep_factors(lg)
    x = torch.view_as_complex(torch.randn(((n1 * n2) // P, 2), dtype=torch.float64, device="cuda"))
    for natural in (False, True):
        t = timeit(lambda: fd.fft1d_fourstep(x, n1, n2, comm, ops, natural_order=natural), comm, reps=3)
        flops = 5.0 * n1 * n2 * lg
        if comm.rank == 0:
            print(f"fft1d 2^{lg} c128 ranks={P} natural_order={natural}: {t*1e3:8.2f} ms  {flops/t/1e9:9.0f} GFLOP/s",
                  flush=True)
    ops.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
