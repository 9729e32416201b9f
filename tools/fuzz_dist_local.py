#!/usr/bin/env python3
"""Randomised sweep of the C ABI's distributed plans over local communicators (2, 4 or 8 ranks sharing the one
GPU: the multi-rank index algebra, chunked pipelines and event ordering of csrc/dist.hip with device copies as the
transport): random power-of-two shapes, chunk and column-group counts and precisions, both output layouts / orders, against numpy's
transform of the assembled input (tests/test_gpu_dist_cabi.py's checkers).  Shapes the tile passes do not serve are
refused by the library (MfftError) and counted apart.
    python tools/fuzz_dist_local.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED]]]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

failures = []
refused = []


def run(first, last, cases, verbose=True):
    import torch
    from fft_amd import dist as fd, MfftError
    import test_gpu_dist_cabi as chk
    torch.cuda.set_device(0)
    count = 0
    for seed in range(first, last + 1):
        rng = np.random.default_rng(seed)
        for _ in range(cases):
            P = int(rng.choice([2, 4, 8]))
            dtype = int(rng.integers(0, 2))
            wmin = 5 if dtype == 0 else 4
            chunks = [None, 1, 2, 3, 4, 8][int(rng.integers(0, 6))]
            groups = [None, 1, 2, 3, 4, 8][int(rng.integers(0, 6))]
            comms = fd.CComm.local(P)
            try:
                if rng.random() < 0.5:
                    lw = int(rng.integers(wmin, 11))                      # columns per rank
                    n1 = P << lw
                    n0 = 1 << int(rng.integers(max(4, P.bit_length() - 1), 12))
                    while n0 * n1 > (1 << 21):
                        n0 >>= 1
                    if n0 < 16 or n0 % P:
                        continue
                    case = ("2d", seed, P, n0, n1, dtype, chunks, groups)
                    chk._check_2d(fd, comms, n0, n1, dtype, chunks, groups)
                else:
                    lw = int(rng.integers(wmin, 11))
                    n2 = P << lw
                    if n2 > (8192 if dtype else 16384):
                        continue
                    n1 = 1 << int(rng.integers(max(4, P.bit_length() - 1), 13))
                    while n1 * n2 > (1 << 21):
                        n1 >>= 1
                    if n1 < 16 or n1 % P:
                        continue
                    case = ("1d", seed, P, n1, n2, dtype, chunks, groups)
                    chk._check_1d(fd, comms, n1, n2, dtype, chunks, groups)
                count += 1
            except AssertionError as e:
                msg = str(e)
                if "MfftError" in msg and ("unsupported" in msg.lower() or "tile" in msg.lower() or "fewer columns" in msg.lower() or "segment" in msg.lower()):
                    refused.append(case)
                else:
                    failures.append(case + (msg[:200],))
            except MfftError as e:
                refused.append(case)
            finally:
                for c in comms:
                    c.close()
        if verbose:
            print("seed", seed, "done,", count, "checked,", len(refused), "refused,", len(failures), "failures", flush=True)
    return count


if __name__ == "__main__":
    argv = [int(v) for v in sys.argv[1:]]
    count = run((argv + [1])[0], (argv + [1, 6])[1], (argv + [1, 6, 30])[2])
    for f in failures:
        print("FAIL", f)
    print("fuzz_dist_local", "ok" if not failures else "FAILED", count, "cases checked,", len(refused), "refused")
    sys.exit(1 if failures else 0)
