#!/usr/bin/env python3
"""Per-GPU LOCAL work of the multi-GPU decompositions at the shapes an N-GPU run gives each
rank, timed on one GPU (the exchanges are replaced by a no-op, so the results are garbage but
every kernel runs on its real per-rank shape, layouts included).

    python tools/local_work_bench.py [--ranks 1 2 4 8]

Beside each time: the modelled all-to-all time of the same run over xGMI, (local bytes / P)
per peer link at 153 GB/s peak, and the resulting projected speed-up over one GPU."""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import dist as fd  # noqa: E402

LINK_GBS = 153.0


class FakeComm:
    """world/rank of an N-rank job, exchange = hand the buffer back (timing of local work only)."""

    def __init__(self, world, rank=0):
        self.world, self.rank, self.backend = world, rank, "none"
        self.exchanges = 0
        self.chunks_seen = 4          # the decompositions' default chunk count with several ranks

    def all_to_all(self, dst, src, label=None):
        self.exchanges += 1
        return src

    def start_exchange(self, send_blocks, recv_blocks, label=None):
        """One chunk of the pipelined first exchange: counted as a fraction of an all-to-all."""
        self.exchanges += 1.0 / self.chunks_seen if self.chunks_seen else 0

        class _Done:
            def wait(self):
                pass
        return _Done()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--log2n", type=int, default=31)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    ops = fd.HipOps()
    base = {}
    for name in ("fft2 16384^2 c64", f"fft1d 2^{args.log2n} c128"):
        for restore in (False, True):
            for P in args.ranks:
                comm = FakeComm(P)
                if name.startswith("fft2"):
                    n = 16384
                    local = torch.view_as_complex(torch.randn(((n // P) * n, 2), dtype=torch.float32, device="cuda"))
                    fn = lambda: fd.fft2_slab(local, n, n, comm, ops, restore_layout=restore)
                    esz, flops = 8, 5.0 * n * n * 2 * math.log2(n)
                    label = "natural layout" if restore else "column slabs"
                else:
                    lg = args.log2n
                    n1, n2 = fd.fourstep_factors(lg)
                    if (n1 * n2 // P) * 16 * 5 > 250 * (1 << 30):
                        continue                      # does not fit one GPU at this rank count
                    local = torch.view_as_complex(torch.randn((n1 * n2 // P, 2), dtype=torch.float64, device="cuda"))
                    fn = lambda: fd.fft1d_fourstep(local, n1, n2, comm, ops, natural_order=restore)
                    esz, flops = 16, 5.0 * n1 * n2 * lg
                    label = "natural order" if restore else "transposed order"
                t = timed(fn)
                comm.exchanges = 0
                fn()
                ex = int(round(comm.exchanges)) if P > 1 else 0
                per_link = local.numel() * esz / P
                t_x = ex * per_link / (LINK_GBS * 1e9)
                key = (name, restore)
                if P == args.ranks[0]:
                    base[key] = (t + t_x) * P / args.ranks[0]
                proj = base[key] * args.ranks[0] / (t + t_x) if key in base else float("nan")
                print(f"{name} {label:17s} P={P}: local {t*1e3:8.2f} ms/GPU | {ex} all-to-all x {per_link/2**20:7.1f} MiB per link "
                      f"= {t_x*1e3:6.2f} ms at {LINK_GBS:.0f} GB/s | no-overlap total {(t+t_x)*1e3:8.2f} ms "
                      f"{flops/(t+t_x)/1e9:8.0f} GFLOP/s", flush=True)
                del local
                ops.close()
                torch.cuda.empty_cache()
    ops.close()


if __name__ == "__main__":
    main()
