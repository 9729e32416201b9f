#!/usr/bin/env python3
"""Randomised sweep (8 seeds x 60 cases by default) over length class, zero padding, batch, precision and
fused options through the C ABI against the numpy chain in float64; BASELINE tolerances.  Failures are
collected and listed at the end (exit status 1).
    python tools/fuzz_combinations.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED [BIG_BATCH]]]]
BIG_BATCH=1 draws batches of up to 40 000 signals for lengths up to 4096, so that the persistent
workgroups of the single-kernel variants take several trips."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import fft_amd as fa
from test_gpu_combinations import _reference
def rel(a, b): return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
all_flags = [0, fa.HANN, fa.POWER, fa.HANN | fa.POWER, fa.POWER | fa.STATS, fa.HANN | fa.POWER | fa.STATS, fa.DEMOD,
             fa.HANN | fa.DEMOD | fa.POWER | fa.STATS, fa.REAL_IN, fa.REAL_IN | fa.HANN | fa.POWER | fa.STATS, fa.REAL_IN | fa.DEMOD, fa.INVERSE]
worst = {0: 0.0, 1: 0.0}
count = 0
failures = []
argv = [int(v) for v in sys.argv[1:]]
first, last = (argv + [1])[0], (argv + [1, 8])[1]
cases, big = (argv + [1, 8, 60])[2], (argv + [1, 8, 60, 0])[3]
for seed in range(first, last + 1):
    rng = np.random.default_rng(seed)
    for _ in range(cases):
        dtype = int(rng.integers(0, 2))
        kind = rng.integers(0, 4)
        if kind == 0: n = 1 << int(rng.integers(1, 15))
        elif kind == 1: n = 1 << int(rng.integers(15, 21))
        elif kind == 2: n = int(rng.integers(2, 5000))
        else: n = int(rng.integers(5000, 300000))
        n_in = n if rng.random() < 0.4 else int(rng.integers(1, n + 1))
        flags = all_flags[int(rng.integers(0, len(all_flags)))]
        if flags & fa.INVERSE and n & (n - 1): flags = 0
        batch = int(rng.integers(1, 4)) if n > 4096 else int(rng.integers(1, 70))
        if big and n <= 4096 and rng.random() < 0.5:
            batch = int(rng.integers(1, max(2, min(40000, (1 << 23) // n))))
        rdt, cdt = (np.float32, np.complex64) if dtype == 0 else (np.float64, np.complex128)
        x = rng.standard_normal((batch, n_in)).astype(rdt) if flags & fa.REAL_IN else \
            (rng.standard_normal((batch, n_in)) + 1j * rng.standard_normal((batch, n_in))).astype(cdt)
        ref = _reference(x, n, flags, fa, 0.37, -0.81, 0.25, 0.01)
        with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
            out = p.exec_host(x, scale=0.37, mu=-0.81, t0=0.25, dt=0.01)
        case = (seed, n, n_in, batch, dtype, flags)
        if flags & fa.STATS:
            out, st = out
            f = np.fft.fftfreq(n) * n
            stol = 1e-5 if dtype == 0 else 1e-11
            if not np.allclose(st[:, 0], ref.max(-1), rtol=stol): failures.append(("max",) + case)
            if not np.allclose(st[:, 1], ref.sum(-1), rtol=stol): failures.append(("sum",) + case)
            if not np.allclose(st[:, 2], (ref * f).sum(-1), rtol=0, atol=stol * np.abs(ref * f).sum(-1).max()):
                failures.append(("centroid",) + case)
        e = rel(out, ref)
        worst[dtype] = max(worst[dtype], e)
        if not e < (1e-5 if dtype == 0 else 1e-12): failures.append(("spectrum", e) + case)
        count += 1
    print("seed", seed, "done,", count, "cases,", len(failures), "failures", flush=True)
for f in failures: print("FAIL", f)
print("fuzz", "ok" if not failures else "FAILED", count, "cases; worst rel L2: c64 %.2e, c128 %.2e" % (worst[0], worst[1]))
sys.exit(1 if failures else 0)
