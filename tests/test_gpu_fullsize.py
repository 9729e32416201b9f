"""BASELINE configurations 4 and 5 at their full sizes on one GPU (the single-rank case of
the multi-GPU decompositions), checked through size-independent properties: a host
transform of 2^31 points is out of reach, so random input is checked on sampled bins against
the definition of the DFT evaluated directly in float64 (exact integer reduction of j*k mod n,
one pass over the input per bin, torch on the device as the calculator), plus Parseval.
Tolerances as in BASELINE.json: 1e-5 (complex64), 1e-12 (complex128), relative to the
root-mean-square bin magnitude."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _energy(t, chunk=1 << 27):
    """sum |t|^2 in float64 without a full-size temporary."""
    import torch
    r = torch.view_as_real(t).reshape(-1)
    total = 0.0
    for lo in range(0, r.numel(), chunk):
        part = r[lo:lo + chunk].double()
        total += float(torch.sum(part * part).item())
    return total


def _direct_bins_1d(x, ks, chunk=1 << 26):
    """X[k] = sum_j x[j] exp(-2 pi i j k / n) for the given bins, float64 accumulation."""
    import torch
    n = x.numel()
    out = []
    for k in ks:
        acc = torch.zeros((), dtype=torch.complex128, device=x.device)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            j = torch.arange(lo, hi, dtype=torch.int64, device=x.device)
            frac = ((j * int(k)) % n).to(torch.float64) / n          # exact: j*k < 2^62
            ang = -2.0 * math.pi * frac
            w = torch.complex(torch.cos(ang), torch.sin(ang))
            acc = acc + torch.sum(x[lo:hi].to(torch.complex128) * w)
            del j, frac, ang, w
        out.append(complex(acc.item()))
    return np.array(out)


def test_cfg5_one_signal_2pow31_c128():
    """BASELINE config 5 (N = 2^31 complex128, 32 GiB) through the four-step decomposition."""
    import torch
    from fft_amd import dist as fd
    torch.cuda.set_device(0)
    free, _ = torch.cuda.mem_get_info()
    if free < 180 * (1 << 30):
        pytest.skip("needs about 170 GiB of free device memory")
    n1, n2 = fd.fourstep_factors(31)                                # 2^18 x 2^13
    n = n1 * n2
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty(n, dtype=torch.complex128, device="cuda")
    xr = torch.view_as_real(x)
    for lo in range(0, n, 1 << 27):                                 # chunked: bounded temporaries
        xr[lo:lo + (1 << 27)].normal_(generator=g)
    e_in = _energy(x)
    comm, ops = fd.Comm(), fd.HipOps()
    try:
        y = fd.fft1d_fourstep(x, n1, n2, comm, ops, natural_order=True)
        torch.cuda.synchronize()
        e_out = _energy(y)
        assert abs(e_out / (n * e_in) - 1.0) < 1e-12                # Parseval
        ks = [0, 1, 2, n1 - 1, n1, n1 + 1, n // 2 - 1, n // 2, n - n2, n - 1, 1234567891, 987654321]
        got = y[torch.tensor(ks, device="cuda")].cpu().numpy()
        del y
        ops.close()
        torch.cuda.empty_cache()
        want = _direct_bins_1d(x, ks)
        rms = math.sqrt(n * e_in / n)                               # root-mean-square |X[k]|
        assert np.max(np.abs(got - want)) / rms < 1e-12
    finally:
        ops.close()


def test_cfg4_2d_16384_c64():
    """BASELINE config 4 (16384 x 16384 complex64, 2 GiB) through the slab decomposition."""
    import torch
    from fft_amd import dist as fd
    torch.cuda.set_device(0)
    n = 16384
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.view_as_complex(torch.randn((n * n, 2), dtype=torch.float32, device="cuda", generator=g))
    comm, ops = fd.Comm(), fd.HipOps()
    try:
        e_in = _energy(a)
        for restore in (False, True):                               # one rank: same (n x n) spectrum either way
            f = fd.fft2_slab(a, n, n, comm, ops, restore_layout=restore).view(n, n)
            torch.cuda.synchronize()
            e_out = _energy(f)
            assert abs(e_out / (float(n) * n * e_in) - 1.0) < 1e-5
            a2 = a.view(n, n)
            j = torch.arange(n, dtype=torch.int64, device="cuda")
            rms = math.sqrt(e_in)
            worst = 0.0
            for k0, k1 in ((0, 0), (1, 0), (0, 1), (127, 128), (128, 127), (8191, 8192), (16383, 16383),
                           (12345, 6789), (4096, 1), (1, 4096)):
                w1 = torch.exp(-2j * math.pi * ((j * k1) % n).double() / n)
                w0 = torch.exp(-2j * math.pi * ((j * k0) % n).double() / n)
                rows = torch.zeros(n, dtype=torch.complex128, device="cuda")
                for lo in range(0, n, 2048):                        # A @ w1 in float64, 2048 rows at a time
                    rows[lo:lo + 2048] = a2[lo:lo + 2048].to(torch.complex128) @ w1
                want = complex(torch.dot(w0, rows).item())
                worst = max(worst, abs(complex(f[k0, k1].item()) - want) / rms)
            assert worst < 1e-5, worst
    finally:
        ops.close()


def test_largest_app_length():
    """The reference's widgets allow tmax/dt up to N0 = 2 000 001 samples and pad 64
    (final2.py:534-535, :599): Nfft = 128 000 064, a chirp-z length.  Plan, run, and compare
    sampled bins of the power spectrum with the direct sum (exact integer phase reduction)."""
    import time
    import fft_amd as fa
    n0, pad, dt = 2000001, 64, 0.01
    n = n0 * pad
    rng = np.random.default_rng(7)
    times = np.arange(n0) * dt
    x = (0.7 * np.exp(-1j * 0.37 * times) + 0.3 * np.exp(+1j * 2.5 * times)
         + 1e-3 * (rng.standard_normal(n0) + 1j * rng.standard_normal(n0)))
    t0 = time.perf_counter()
    with fa.Plan(n, n_in=n0, dtype=fa.C128, flags=fa.HANN | fa.POWER | fa.STATS) as p:
        assert time.perf_counter() - t0 < 10.0, "plan tables are generated on the device"
        S, st = p.exec_host(x, scale=dt)
    assert S.shape == (n,) and S.dtype == np.float64
    k_peak = int(np.argmax(S))
    assert k_peak == n - round(0.37 * dt * n / (2 * np.pi))           # E = -omega convention: bin of -0.37
    assert st[0, 0] == S[k_peak]
    xw = x * np.hanning(n0)
    j = np.arange(n0, dtype=np.int64)
    for k in (0, 1, k_peak - 1, k_peak, k_peak + 1, n // 2, n - 1, 50937212, round(2.5 * dt * n / (2 * np.pi))):
        ph = ((j * int(k)) % n).astype(np.float64) / n                 # exact: j*k < 2^63
        want = np.abs(dt * np.sum(xw * np.exp(-2j * np.pi * ph))) ** 2
        assert abs(S[k] - want) <= 1e-11 * S[k_peak] + 1e-9 * want, (k, S[k], want)


def test_cfg5_2pow31_c128_single_plan():
    """The same 2^31-point complex128 transform as ONE plan on one GPU: three global passes
    (2048 * 1024 * 1024), natural-order output, 32 GiB of workspace."""
    import torch
    import fft_amd as fa
    torch.cuda.set_device(0)
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs about 130 GiB of free device memory")
    n = 1 << 31
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.empty(n, dtype=torch.complex128, device="cuda")
    xr = torch.view_as_real(x)
    for lo in range(0, n, 1 << 27):
        xr[lo:lo + (1 << 27)].normal_(generator=g)
    y = torch.empty_like(x)
    with fa.Plan(n, dtype=fa.C128) as p:
        assert p.launches == 3
        p.exec(x, y, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    e_in, e_out = _energy(x), _energy(y)
    assert abs(e_out / (n * e_in) - 1.0) < 1e-12
    ks = [0, 1, 255, 256, 65535, 65536, n // 2 - 1, n // 2, n - 256, n - 1, 1234567891, 987654321]
    got = y[torch.tensor(ks, device="cuda")].cpu().numpy()
    del y
    torch.cuda.empty_cache()
    want = _direct_bins_1d(x, ks)
    assert np.max(np.abs(got - want)) / math.sqrt(e_in) < 1e-12


def _on_rank_threads(comms, fn):
    """fn(comm) on one host thread per rank of a local communicator (an exchange waits for all)."""
    import threading
    out, errs = [None] * len(comms), []

    def work(i):
        try:
            out[i] = fn(comms[i])
        except Exception as e:                                          # noqa: BLE001
            errs.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(comms))]
    [t.start() for t in threads]
    [t.join(timeout=280) for t in threads]
    assert not any(t.is_alive() for t in threads), "a rank did not finish"
    assert not errs, errs
    return out


@pytest.mark.parametrize("P", [2, 4, 8])
def test_multi_rank_shapes_of_the_scaling_runs(P):
    """The exact shapes `bench.py --gpus P` gives the distributed plans of the C ABI -- 16384 x 16384
    complex64 over P ranks (both output layouts) and one complex128 signal of 2^(28 + log2 P) points
    (2^31 on 8 ranks) in natural order -- with the P ranks as a local communicator on the one GPU:
    slab widths, chunking, segment sizes and 64-bit offsets are those of the multi-GPU job, only the
    transport differs.  Sampled bins against the DFT evaluated directly in float64."""
    import torch
    from fft_amd import dist as fd
    torch.cuda.set_device(0)
    lg = 31 if P == 8 else 28 + int(math.log2(P))
    torch.cuda.empty_cache()                     # the plans' buffers come from hipMalloc, not from torch's cache
    free, _ = torch.cuda.mem_get_info()
    if free < (6 * 16 << lg) + (8 << 30):
        pytest.skip("needs about %d GiB of free device memory" % ((6 * 16 << lg) >> 30))
    comms = fd.CComm.local(P)
    try:
        # ---- config 4
        n = 16384
        R, W = n // P, n // P
        g = torch.Generator(device="cuda").manual_seed(44)
        a = torch.view_as_complex(torch.randn((n * n, 2), dtype=torch.float32, device="cuda", generator=g))
        a2 = a.view(n, n)
        rms = math.sqrt(_energy(a))
        j = torch.arange(n, dtype=torch.int64, device="cuda")
        bins = [(0, 0), (1, 0), (0, 1), (R - 1, W), (R, W - 1), (8191, 8192), (16383, 16383), (12345, 6789), (n - R, 1), (1, n - W)]
        want = []
        for k0, k1 in bins:
            w1 = torch.exp(-2j * math.pi * ((j * k1) % n).double() / n)
            w0 = torch.exp(-2j * math.pi * ((j * k0) % n).double() / n)
            acc = torch.zeros((), dtype=torch.complex128, device="cuda")
            for lo in range(0, n, 2048):
                acc = acc + torch.dot(w0[lo:lo + 2048], a2[lo:lo + 2048].to(torch.complex128) @ w1)
            want.append(complex(acc.item()))
        for restore in (False, True):
            def one(c):
                torch.cuda.set_device(0)
                with torch.cuda.stream(torch.cuda.Stream()):
                    f = fd.fft2_slab(a[c.rank * R * n:(c.rank + 1) * R * n], n, n, c, None, restore_layout=restore)
                    torch.cuda.current_stream().synchronize()
                    return f
            outs = _on_rank_threads(comms, one)
            for (k0, k1), w in zip(bins, want):
                got = outs[k0 // R][(k0 % R) * n + k1] if restore else outs[k1 // W][k0 * W + k1 % W]
                assert abs(complex(got.item()) - w) / rms < 1e-5, (P, restore, k0, k1)
            del outs
        for c in comms:                                               # free the 2-D plans' buffers
            for h in c._plans.values():
                fd._lib.load().mfft_dist_plan_destroy(h)
            c._plans.clear()
        del a, a2
        torch.cuda.empty_cache()
        # ---- config 5
        n1, n2 = fd.fourstep_factors(lg)
        N = n1 * n2
        cnt = N // P
        g = torch.Generator(device="cuda").manual_seed(55)
        x = torch.empty(N, dtype=torch.complex128, device="cuda")
        xr = torch.view_as_real(x)
        for lo in range(0, N, 1 << 27):
            xr[lo:lo + (1 << 27)].normal_(generator=g)
        e_in = _energy(x)

        def one1(c):
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                y = fd.fft1d_fourstep(x[c.rank * cnt:(c.rank + 1) * cnt], n1, n2, c, None, natural_order=True)
                torch.cuda.current_stream().synchronize()
                return y
        outs = _on_rank_threads(comms, one1)
        e_out = sum(_energy(o) for o in outs)
        assert abs(e_out / (N * e_in) - 1.0) < 1e-12                    # Parseval over all ranks
        ks = [0, 1, cnt - 1, cnt, n1 - 1, n1, n1 + 1, N // 2 - 1, N // 2, N - n2, N - 1, 1234567891 % N, 987654321 % N]
        got = np.array([complex(outs[k // cnt][k % cnt].item()) for k in ks])
        del outs
        for c in comms:
            c.close()
        comms = []
        torch.cuda.empty_cache()
        want = _direct_bins_1d(x, ks)
        assert np.max(np.abs(got - want)) / math.sqrt(e_in) < 1e-12
    finally:
        for c in comms:
            c.close()
