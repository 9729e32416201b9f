#!/usr/bin/env python3
"""Randomised sweep of real-input plans above one LDS residency (the packed route with the split fused into the last
pass) against np.fft.fft in float64: lengths 2^15 .. 2^24, even and odd sample counts (odd ones take the promoting
route), batches, Hann / power / statistics, both precisions, random MFFT_SPLIT of the half-length plan.  With
COMPLEX=1 the samples are complex: the windowed / zero-padded first passes of the tile kernels at random splits.
    python tools/fuzz_long_plans.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED [COMPLEX]]]]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import fft_amd as fa


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def random_split(rng, log2m, dtype):
    """log2 radices of a two- or three-pass split of 2^log2m the tile kernels serve, or None for the default"""
    if rng.random() < 0.5:
        return None
    top = 10 if dtype == 0 else 11
    for _ in range(20):
        k = 2 if log2m <= 20 and rng.random() < 0.7 else 3
        parts = [int(rng.integers(4, top + 1)) for _ in range(k - 1)]
        last = log2m - sum(parts)
        if 4 <= last <= top and parts[0] >= 6:
            return ",".join(str(v) for v in parts + [last])
    return None


def main():
    argv = [int(v) for v in sys.argv[1:]]
    first, last, cases = (argv + [1])[0], (argv + [1, 4])[1], (argv + [1, 4, 25])[2]
    cplx = bool((argv + [1, 4, 25, 0])[3])
    failures, count, refused = [], 0, 0
    worst = {0: 0.0, 1: 0.0}
    for seed in range(first, last + 1):
        rng = np.random.default_rng(seed)
        for _ in range(cases):
            dtype = int(rng.integers(0, 2))
            log2n = int(rng.integers(15 if dtype == 1 else 16, 25))
            n = 1 << log2n
            batch = int(rng.integers(1, max(2, min(9, (1 << 24) // n + 1))))
            n_in = n if rng.random() < 0.4 else int(rng.integers(n // 4, n + 1))
            if rng.random() < 0.8:
                n_in -= n_in & 1
            n_in = max(n_in, 2)
            flags = (0 if cplx else fa.REAL_IN) | (fa.HANN if rng.random() < 0.5 else 0)
            power = rng.random() < 0.5
            stats = power and rng.random() < 0.5
            flags |= (fa.POWER if power else 0) | (fa.STATS if stats else 0)
            split = random_split(rng, log2n, dtype) if cplx else (random_split(rng, log2n - 1, dtype) if n_in % 2 == 0 else None)
            if split:
                os.environ["MFFT_SPLIT"] = split
            else:
                os.environ.pop("MFFT_SPLIT", None)
            rdt = np.float32 if dtype == 0 else np.float64
            x = rng.standard_normal((batch, n_in)).astype(rdt)
            if cplx:
                x = (x + 1j * rng.standard_normal((batch, n_in))).astype(np.complex64 if dtype == 0 else np.complex128)
            scale = float(rng.choice([1.0, 0.01]))
            xw = x.astype(np.complex128 if cplx else np.float64) * (np.hanning(n_in) if flags & fa.HANN else 1.0)
            ref = scale * np.fft.fft(xw, n, axis=-1)
            if power:
                ref = np.abs(ref) ** 2
            case = (seed, dtype, log2n, batch, n_in, flags, split, scale)
            try:
                with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
                    out = p.exec_host(x, scale=scale)
            except fa.MfftError as e:
                if split and "split" in str(e).lower() or "radix" in str(e).lower() or "tile" in str(e).lower():
                    refused += 1
                    continue
                failures.append(("error", str(e)) + case)
                continue
            finally:
                os.environ.pop("MFFT_SPLIT", None)
            y, st = out if stats else (out, None)
            err = rel(y, ref)
            worst[dtype] = max(worst[dtype], err)
            tol = 1e-5 if dtype == 0 else 1e-12
            if not err < tol:
                failures.append(("bins", err) + case)
            if st is not None:
                f = np.fft.fftfreq(n) * n
                stol = 1e-5 if dtype == 0 else 1e-10
                ok = (np.allclose(st[:, 0], ref.max(axis=1), rtol=stol) and np.allclose(st[:, 1], ref.sum(axis=1), rtol=stol)
                      and np.all(np.abs(st[:, 2] - (ref * f).sum(axis=1)) <= stol * np.abs(ref * f).sum(axis=1)))
                if not ok:
                    failures.append(("stats",) + case)
            count += 1
        print(f"seed {seed} done, {count} cases, {refused} splits refused, {len(failures)} failures", flush=True)
    for f in failures:
        print("FAIL", f)
    print(f"fuzz_long_plans {'ok' if not failures else 'FAILED'} {count} cases; worst rel L2: float32 {worst[0]:.2e}, float64 {worst[1]:.2e}")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
