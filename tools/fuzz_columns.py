#!/usr/bin/env python3
"""Randomised sweep of column-batched plans (mfft_plan_1d_columns: np.fft.fft(a, axis=-2) on a (batch, n, inner)
array, optionally with the fused four-step twiddle) against numpy in float64: lengths 16 .. 2^17, inner extents
from the narrowest a tile takes to 4096, 1-3 arrays, both precisions, twiddle moduli 2 .. 2^40 with large column
offsets.  Relative L2 <= 1e-5 (complex64) / 1e-12 (complex128).
    python tools/fuzz_columns.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED]]]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import fft_amd as fa

failures = []
worst = {0: 0.0, 1: 0.0}


def run(first, last, cases, verbose=True):
    count = 0
    for seed in range(first, last + 1):
        rng = np.random.default_rng(seed)
        for _ in range(cases):
            dtype = int(rng.integers(0, 2))
            cdt = np.complex64 if dtype == 0 else np.complex128
            lg = int(rng.integers(4, 18))
            li = int(rng.integers(5 if dtype == 0 else 4, 13))
            batch = int(rng.integers(1, 4))
            while lg + li + (batch - 1).bit_length() > 22:
                if li > (5 if dtype == 0 else 4) and rng.random() < 0.5:
                    li -= 1
                elif lg > 4:
                    lg -= 1
                else:
                    batch = 1
            n, inner = 1 << lg, 1 << li
            tw = None
            if rng.random() < 0.5:
                n_total = 1 << int(rng.integers(1, 41))
                c0 = int(rng.integers(0, max(1, min(n_total, 1 << 34))))
                tw = (n_total, c0)
            x = (rng.standard_normal((batch, n, inner)) + 1j * rng.standard_normal((batch, n, inner))).astype(cdt)
            case = (seed, dtype, n, inner, batch, tw)
            try:
                with fa.ColumnPlan(n, inner, batch=batch, dtype=dtype, twiddle=tw) as p:
                    y = p.exec_host(x)
            except Exception as e:                                # noqa: BLE001
                failures.append(("raised", repr(e)[:120]) + case)
                continue
            ref = np.fft.fft(x.astype(np.complex128), axis=1)
            if tw is not None:
                k = np.arange(n, dtype=object)[:, None]
                c = (tw[1] + np.arange(inner, dtype=object))[None, :]
                kc = np.array((k * c) % tw[0], dtype=np.float64)          # exact integer reduction
                ref = ref * np.exp(-2j * np.pi * (kc / tw[0]))[None]
            e = float(np.linalg.norm((y - ref).ravel()) / np.linalg.norm(ref.ravel()))
            worst[dtype] = max(worst[dtype], e)
            count += 1
            if not e < (1e-5 if dtype == 0 else 1e-12):
                failures.append(("values", e) + case)
        if verbose:
            print("seed", seed, "done,", count, "cases,", len(failures), "failures", flush=True)
    return count


if __name__ == "__main__":
    argv = [int(v) for v in sys.argv[1:]]
    count = run((argv + [1])[0], (argv + [1, 6])[1], (argv + [1, 6, 40])[2])
    for f in failures:
        print("FAIL", f)
    print("fuzz_columns", "ok" if not failures else "FAILED", count, "cases; worst rel L2 complex64 %.2e complex128 %.2e" % (worst[0], worst[1]))
    sys.exit(1 if failures else 0)
