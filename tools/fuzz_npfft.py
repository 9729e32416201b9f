#!/usr/bin/env python3
"""Randomised comparison of fft_amd.fft / fft2 with numpy.fft.fft / fft2 (the call of final2.py:143)
over what the numpy signature admits: 1-4 dimensions, any axis, n shorter / equal / longer than the
axis, every norm, real / complex / integer / half dtypes, non-contiguous views, `out`, and torch
CUDA tensors.  Checks shape, result dtype and relative L2 (1e-5 single, 1e-12 double).
    python tools/fuzz_npfft.py [FIRST_SEED [LAST_SEED [CASES_PER_SEED]]]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import fft_amd as fa

DTYPES = [np.complex128, np.complex64, np.float64, np.float32, np.float16, np.int32, np.int8, np.bool_]
failures = []
worst = {"single": 0.0, "double": 0.0}


def rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    den = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / den) if den > 0 else float(np.linalg.norm(a.ravel()))


def draw(rng):
    nd = int(rng.integers(1, 5))
    kind = rng.integers(0, 3)
    shape = []
    for _ in range(nd):
        if kind == 0:
            shape.append(int(rng.integers(1, 40)))
        elif kind == 1:
            shape.append(1 << int(rng.integers(0, 8)))
        else:
            shape.append(int(rng.choice([1, 2, 3, 16, 17, 64, 100, 256])))
    while np.prod(shape) > (1 << 20):
        shape[int(np.argmax(shape))] //= 2
    if rng.random() < 0.15:                                  # one long axis
        shape[int(rng.integers(0, nd))] = int(rng.choice([1000, 4096, 5001, 1 << 15, 24016, 70000]))
        while np.prod(shape) > (1 << 21):
            shape[int(np.argmin([s if s > 1 else 1 << 30 for s in shape]))] = 1
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    if dt in (np.complex128, np.complex64):
        a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
    elif dt == np.bool_:
        a = rng.random(shape) < 0.5
    elif dt in (np.int32, np.int8):
        a = rng.integers(-100, 100, shape).astype(dt)
    else:
        a = rng.standard_normal(shape).astype(dt)
    if nd >= 2 and rng.random() < 0.3:                       # a non-contiguous view
        a = a.swapaxes(0, nd - 1)
    elif rng.random() < 0.2 and a.shape[-1] >= 4:
        a = a[..., ::2]
    axis = int(rng.integers(-nd, nd))
    have = a.shape[axis]
    r = rng.random()
    n = None if r < 0.4 else int(rng.integers(1, have + 1)) if r < 0.6 else int(rng.integers(have, 3 * have + 2))
    norm = [None, "backward", "ortho", "forward"][int(rng.integers(0, 4))]
    return a, n, axis, norm


def run(first, last, cases, verbose=True):
    import torch
    count = 0
    for seed in range(first, last + 1):
        rng = np.random.default_rng(seed)
        for _ in range(cases):
            a, n, axis, norm = draw(rng)
            ref = np.fft.fft(a, n=n, axis=axis, norm=norm)
            case = (seed, a.shape, str(a.dtype), n, axis, norm, a.flags["C_CONTIGUOUS"])
            single = ref.dtype == np.complex64
            tol = 1e-5 if single else 1e-12
            mode = rng.integers(0, 4)
            if mode == 0 and a.dtype != np.float16 and a.dtype.kind in "fc":
                got = fa.fft(torch.from_numpy(np.ascontiguousarray(a)).cuda(), n=n, axis=axis, norm=norm).cpu().numpy()
            elif mode == 1:
                out = np.empty(ref.shape, dtype=ref.dtype)
                got = fa.fft(a, n=n, axis=axis, norm=norm, out=out)
                if got is not out:
                    failures.append(("out identity",) + case)
            else:
                got = fa.fft(a, n=n, axis=axis, norm=norm)
            count += 1
            if got.shape != ref.shape or got.dtype != ref.dtype:
                failures.append(("shape/dtype", got.shape, str(got.dtype), ref.shape, str(ref.dtype)) + case)
                continue
            e = rel(got, ref)
            worst["single" if single else "double"] = max(worst["single" if single else "double"], e)
            if not e < tol:
                failures.append(("values", e) + case)
            if a.ndim >= 2 and rng.random() < 0.25:
                r2 = np.fft.fft2(a, norm=norm)
                g2 = fa.fft2(a, norm=norm)
                count += 1
                if g2.shape != r2.shape or g2.dtype != r2.dtype or not rel(g2, r2) < tol:
                    failures.append(("fft2", g2.shape, str(g2.dtype), rel(g2, r2) if g2.shape == r2.shape else None) + case)
        if verbose:
            print("seed", seed, "done,", count, "cases,", len(failures), "failures", flush=True)
    return count


if __name__ == "__main__":
    argv = [int(v) for v in sys.argv[1:]]
    count = run((argv + [1])[0], (argv + [1, 4])[1], (argv + [1, 4, 60])[2])
    for fl in failures:
        print("FAIL", fl)
    print("fuzz_npfft", "ok" if not failures else "FAILED", count, "cases; worst rel L2 single %.2e double %.2e"
          % (worst["single"], worst["double"]))
    sys.exit(1 if failures else 0)
