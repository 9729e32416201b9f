#!/usr/bin/env python3
"""Randomised check that the launch shape of the single-kernel transforms never changes a result: random length
(2^2 .. 2^14), precision, flag set and ragged batch (several resident sets' worth of signals), executed under random
MFFT_LDS_GRID_OVER / MFFT_LDS_TAIL settings and compared bit for bit with the same plan under one striding resident set
(statistics of multi-wave signals are summed with atomics: compared to 1e-12), and against numpy on a few signals.

    python tools/fuzz_launch_shapes.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED]]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_amd as fa  # noqa: E402

FLAGS = (0, fa.HANN, fa.POWER, fa.HANN | fa.POWER, fa.REAL_IN, fa.POWER | fa.STATS, fa.HANN | fa.POWER | fa.STATS)


def build(n, batch, dtype, flags, over, tail):
    for k, v in (("MFFT_LDS_GRID_OVER", over), ("MFFT_LDS_TAIL", tail)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    p = fa.Plan(n, batch=batch, dtype=dtype, flags=flags)
    os.environ.pop("MFFT_LDS_GRID_OVER", None)
    os.environ.pop("MFFT_LDS_TAIL", None)
    return p


def main():
    argv = [int(v) for v in sys.argv[1:]]
    first = argv[0] if argv else 0
    last = argv[1] if len(argv) > 1 else first + 3
    per_seed = argv[2] if len(argv) > 2 else 40
    cases = failures = 0
    for seed in range(first, last + 1):
        rng = np.random.default_rng(seed)
        for _ in range(per_seed):
            dtype = int(rng.integers(0, 2))
            log2n = int(rng.integers(2, 15 if dtype == 0 else 14))
            n = 1 << log2n
            flags = FLAGS[int(rng.integers(0, len(FLAGS)))]
            points = 1 << int(rng.integers(21, 25))
            batch = max(1, points // n + int(rng.integers(-7, 8)))
            cdt, rdt = (torch.complex64, torch.float32) if dtype == 0 else (torch.complex128, torch.float64)
            g = torch.Generator(device="cuda").manual_seed(seed * 1000 + cases)
            x = torch.randn(batch, n, dtype=rdt if flags & fa.REAL_IN else cdt, device="cuda", generator=g)
            odt = rdt if flags & fa.POWER else cdt
            y0, y1 = torch.empty(batch, n, dtype=odt, device="cuda"), torch.empty(batch, n, dtype=odt, device="cuda")
            st0 = st1 = None
            if flags & fa.STATS:
                st0 = torch.zeros(batch, 3, dtype=torch.float64, device="cuda")
                st1 = torch.zeros(batch, 3, dtype=torch.float64, device="cuda")
            over = [None, "2", "3", "5", "16", "64"][int(rng.integers(0, 6))]
            tail = [None, f"{int(rng.integers(1, 101))},{int(rng.integers(1, 6))}", "-1,1"][int(rng.integers(0, 3))]
            with build(n, batch, dtype, flags, "1", "-1,1") as p0, build(n, batch, dtype, flags, over, tail) as p1:
                p0.exec(x, y0, d_stats=st0)
                for _ in range(2):                                  # twice: nothing is left behind by the first execution
                    y1.zero_()
                    if st1 is not None:
                        st1.zero_()
                    p1.exec(x, y1, d_stats=st1)
                torch.cuda.synchronize()
            ok = bool(torch.equal(torch.view_as_real(y0) if y0.is_complex() else y0, torch.view_as_real(y1) if y1.is_complex() else y1))
            if st0 is not None:
                # (the centroid sum cancels: its error scales with n * sum S, not with its own size)
                ok = ok and bool(torch.allclose(st0[:, :2], st1[:, :2], rtol=1e-12, atol=0))
                ok = ok and bool(((st0[:, 2] - st1[:, 2]).abs() <= 1e-12 * n * st0[:, 1]).all())
            rows = [0, batch // 2, batch - 1]
            xs = x[rows].cpu().numpy().astype(np.float64 if flags & fa.REAL_IN else np.complex128)
            ref = np.fft.fft(xs * (np.hanning(n) if flags & fa.HANN else 1.0), axis=-1)
            ref = np.abs(ref) ** 2 if flags & fa.POWER else ref
            got = y1[rows].cpu().numpy()
            err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)
            ok = ok and err < (1e-5 if dtype == 0 else 1e-12)
            cases += 1
            if not ok:
                failures += 1
                print(f"MISMATCH seed {seed}: n={n} batch={batch} dtype={dtype} flags={flags} over={over} tail={tail} rel L2 {err:.2e}", flush=True)
            del x, y0, y1, st0, st1
        print(f"seed {seed} done, {cases} cases, {failures} failures", flush=True)
    print(f"fuzz_launch_shapes {'ok' if not failures else 'FAILED'} {cases} cases")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
