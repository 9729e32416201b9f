"""GPU parity for the multi-kernel paths: global Stockham passes (lengths above
one LDS residency) and the chirp-z wrapper (lengths that are not powers of two),
through the C ABI, against the CPU oracle.  Tolerances as in BASELINE.json:
relative L2 <= 1e-5 (complex64) / 1e-12 (complex128) against float64."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu

TOL = {0: 1e-5, 1: 1e-12}


def _rand(shape, seed, dtype):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


@pytest.fixture(scope="module")
def fa():
    import fft_amd
    assert fft_amd.device_count() >= 1
    return fft_amd


@pytest.mark.parametrize("split,log2n,batch", [("9,9,4", 22, 1), ("9,9,5", 23, 1), ("9,5,9", 23, 1), ("5,9,9", 23, 1), ("9,9,4", 22, 3)])
def test_prefetching_512_point_tiles_against_numpy(fa, monkeypatch, split, log2n, batch):
    """The 512-point complex128 tile passes keep a second register set and request the next tile behind
    their first exchange; the sets swap roles from trip to trip.  Forced onto lengths numpy still
    transforms in a second (MFFT_SPLIT is read when the plan is built), as first, middle and last pass,
    with two to twelve trips per workgroup (odd and even counts), all bins against the oracle's engine."""
    monkeypatch.setenv("MFFT_SPLIT", split)
    n = 1 << log2n
    x = _rand((batch, n), 77 + log2n + batch, np.complex128)
    with fa.Plan(n, batch=batch, dtype=1) as p:
        assert p.launches == 3
        y = p.exec_host(x)
    assert rel_l2(y, so.dft(x, engine="numpy")) < TOL[1]


@pytest.mark.parametrize("split,log2n,batch", [("9,9", 18, 5), ("10,9", 19, 3), ("9,9,4", 22, 1), ("4,9,9", 22, 2), ("5,9,9", 23, 1)])
def test_complex64_512_point_tiles_with_evaluated_roots(fa, monkeypatch, split, log2n, batch):
    """The complex64 512-point tile passes evaluate their inter-pass roots per thread (every other complex64
    tile reads the two-level tables): as last pass of two- and three-pass plans and as middle pass, all
    bins against the oracle's engine."""
    monkeypatch.setenv("MFFT_SPLIT", split)
    n = 1 << log2n
    x = _rand((batch, n), 99 + log2n + batch, np.complex64)
    with fa.Plan(n, batch=batch, dtype=0) as p:
        assert p.launches == len(split.split(","))
        y = p.exec_host(x)
    assert rel_l2(y, so.dft(x, engine="numpy")) < TOL[0]


@pytest.mark.parametrize("split,log2n,batch,stats", [("9,9", 18, 24, True), ("9,9", 18, 17, False), ("9,4,9", 22, 1, True),
                                                     ("5,9,9", 23, 1, False), ("10,9", 19, 9, True)])
def test_prefetching_power_last_pass_against_numpy(fa, monkeypatch, split, log2n, batch, stats):
    """The power-spectrum last pass of the 512-point complex128 tiles prefetches like the plain passes: one to
    four trips per workgroup (odd and even counts, batches that end inside a sweep), with and without the
    statistics, every bin and the three statistics against numpy."""
    monkeypatch.setenv("MFFT_SPLIT", split)
    n = 1 << log2n
    x = _rand((batch, n), 55 + log2n + batch, np.complex128)
    flags = fa.POWER | (fa.STATS if stats else 0)
    with fa.Plan(n, batch=batch, dtype=1, flags=flags) as p:
        assert p.launches >= len(split.split(","))          # the statistics add their reduction
        res = p.exec_host(x, scale=0.5)
    S, st = res if stats else (res, None)
    ref = np.abs(0.5 * np.fft.fft(x, axis=-1)) ** 2
    assert rel_l2(S, ref) < 1e-13
    if stats:
        f = np.fft.fftfreq(n) * n
        assert np.max(np.abs(st[:, 0] - ref.max(axis=1)) / ref.max(axis=1)) < 1e-12
        assert np.max(np.abs(st[:, 1] - ref.sum(axis=1)) / ref.sum(axis=1)) < 1e-12
        assert np.max(np.abs(st[:, 2] - (ref * f).sum(axis=1)) / np.abs(ref * f).sum(axis=1)) < 1e-11


@pytest.mark.parametrize("split,log2n,batch,n_in,power", [
    ("9,9", 18, 3, 1 << 18, False), ("9,9", 18, 24, 200001, True), ("9,9", 18, 17, 131072, False), ("9,9", 18, 2, 2, False),
    ("9,9", 18, 2, 3, True), ("9,9", 18, 5, 513, False), ("9,9", 18, 5, (1 << 18) - 1, True), ("9,8", 17, 40, 100000, True),
    ("9,9,4", 22, 1, 3000001, True), ("9,4,9", 22, 2, 1 << 21, False)])
def test_first_pass_with_evaluated_hann_window_against_numpy(fa, monkeypatch, split, log2n, batch, n_in, power):
    """np.fft.fft(x * np.hanning(n_in), n) (final2.py:139-143) where the first pass is a prefetching 512-point
    complex128 tile: the window is evaluated per thread instead of read from its table, samples past n_in are
    clamped loads turned into zeros.  Full and padded lengths, odd and tiny sample counts (np.hanning(2) is
    all zeros), one to several trips per workgroup; and the table route gives the same spectrum."""
    monkeypatch.setenv("MFFT_SPLIT", split)
    n = 1 << log2n
    x = _rand((batch, n_in), 31 + log2n + batch, np.complex128)
    flags = fa.HANN | (fa.POWER if power else 0)
    ref = np.fft.fft(x * np.hanning(n_in), n, axis=-1)
    if power:
        ref = np.abs(ref) ** 2
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=1, flags=flags) as p:
        y = p.exec_host(x)
    monkeypatch.setenv("MFFT_TABLE_WINDOW", "1")              # read when a plan is built
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=1, flags=flags) as p:
        yt = p.exec_host(x)
    monkeypatch.delenv("MFFT_TABLE_WINDOW")
    if np.any(ref):
        assert rel_l2(y, ref) < 1e-13 and rel_l2(yt, ref) < 1e-13
    else:
        assert not np.any(y) and not np.any(yt)
    if n_in < n:
        # the same pass serves zero padding without a window: np.fft.fft(x, n)
        with fa.Plan(n, n_in=n_in, batch=batch, dtype=1, flags=fa.POWER if power else 0) as p:
            y = p.exec_host(x)
        ref = np.fft.fft(x, n, axis=-1)
        assert rel_l2(y, np.abs(ref) ** 2 if power else ref) < 1e-13


@pytest.mark.parametrize("dtype,split,log2n,batch,n_in,power", [
    (0, "4,9", 13, 9, 8191, False), (0, "5,9", 14, 5, 9001, True), (0, "6,8", 14, 7, 1 << 14, False), (0, "7,8", 15, 6, 20001, True),
    (0, "8,9", 17, 5, 100003, False), (0, "9,8", 17, 5, 1 << 17, True), (0, "9,9", 18, 3, 2, False), (0, "10,9", 19, 3, 333333, True),
    (0, "10,9", 19, 2, 1 << 19, False), (0, "8,8,8", 24, 1, 9000001, True),
    (1, "4,9", 13, 4, 5000, True), (1, "6,8", 14, 3, 1 << 14, False), (1, "7,9", 16, 5, 40001, False), (1, "8,8", 16, 9, 3, True),
    (1, "10,9", 19, 2, 400001, True), (1, "10,9", 19, 9, 1 << 19, False), (1, "11,9", 20, 1, 777777, False)])
def test_first_pass_with_evaluated_window_and_range_checked_loads(fa, monkeypatch, dtype, split, log2n, batch, n_in, power):
    """np.fft.fft(x * np.hanning(n_in), n) (final2.py:139-143) and np.fft.fft(x, n) where the first pass is any tile
    other than the prefetching one: the samples come through buffer loads whose range check returns zero past sample
    n_in - 1 (the zero padding), the window is evaluated per thread.  Every first-pass radix of both precisions, full
    and padded lengths, odd and tiny sample counts; the table route gives the same spectrum."""
    monkeypatch.setenv("MFFT_SPLIT", split)
    n = 1 << log2n
    cdt, tol = (np.complex64, 1e-5) if dtype == 0 else (np.complex128, 1e-13)
    x = _rand((batch, n_in), 77 + log2n + batch, cdt)
    x64 = x.astype(np.complex128)
    for window in (True, False):
        if not window and n_in == n:
            continue                      # that is the plain first pass
        flags = (fa.HANN if window else 0) | (fa.POWER if power else 0)
        ref = np.fft.fft(x64 * np.hanning(n_in) if window else x64, n, axis=-1)
        if power:
            ref = np.abs(ref) ** 2
        with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
            y = p.exec_host(x)
        if np.any(ref):
            assert rel_l2(y, ref) < tol, (window, rel_l2(y, ref))
        else:
            assert not np.any(y)
        if window:
            monkeypatch.setenv("MFFT_TABLE_WINDOW", "1")
            with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
                yt = p.exec_host(x)
            monkeypatch.delenv("MFFT_TABLE_WINDOW")
            assert rel_l2(yt, ref) < tol if np.any(ref) else not np.any(yt)


@pytest.mark.parametrize("var,value,case", [
    ("MFFT_FOUR_PASSES", "1", "c128_2p20"), ("MFFT_GENERAL_PROLOGUE", "1", "padded_window"), ("MFFT_GENERAL_PROLOGUE", "1", "chirpz"),
    ("MFFT_NO_NARROW_RADIX", "1", "c64_2p19"), ("MFFT_TILE_BUF", "0", "c64_2p19"), ("MFFT_TILE_BUF", "4", "c64_2p20"),
    ("MFFT_TILE_BUF", "12", "c128_2p20"), ("MFFT_TILE_NT", "0", "c64_2p20"), ("MFFT_TILE_NT", "3", "c128_2p20"),
    ("MFFT_TILE_NT_PASS", "1,3,2", "c64_2p20"), ("MFFT_WORK2", "1", "c64_2p20"), ("MFFT_R2C_FROM", "30", "real_2p17"),
    ("MFFT_BALANCED_COLUMNS", "1", "columns"), ("MFFT_NARROW_COLUMNS", "1", "columns"),
    ("MFFT_COPY_THREADS", "1", "host_path"), ("MFFT_COPY_STREAM", "0", "host_path"),
    ("MFFT_TILE_GRID_OVER", "1", "c64_2p22"), ("MFFT_TILE_GRID_OVER", "3", "c64_2p22"), ("MFFT_TILE_GRID_OVER", "64", "c64_2p19"),
    ("MFFT_LDS_GRID_OVER", "1", "lds_batches"), ("MFFT_LDS_GRID_OVER", "3", "lds_batches"), ("MFFT_LDS_GRID_OVER", "64", "lds_batches"),
    ("MFFT_LDS_TAIL", "-1,1", "lds_batches"), ("MFFT_LDS_TAIL", "30,3", "lds_batches"), ("MFFT_LDS_TAIL", "100,1", "lds_batches")])
def test_tuning_variables_leave_results_unchanged(fa, monkeypatch, var, value, case):
    """Every variable the library reads when a plan is built or executed (tuning aids: pass splits, cache policies,
    access forms, routes) selects another way to the same bins: np.fft.fft (final2.py:143) stays the reference."""
    from fft_amd.engine import ColumnPlan
    monkeypatch.setenv(var, value)
    rng = np.random.default_rng(len(var) + len(case))

    def cplx(shape, cdt):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(cdt)
    if case in ("c64_2p19", "c64_2p20", "c64_2p22", "c128_2p20"):
        dtype = 0 if case.startswith("c64") else 1
        n = 1 << int(case[-2:])
        x = cplx((2, n), np.complex64 if dtype == 0 else np.complex128)
        with fa.Plan(n, batch=2, dtype=dtype) as p:
            y = p.exec_host(x)
        assert rel_l2(y, np.fft.fft(x.astype(np.complex128), axis=-1)) < TOL[dtype]
    elif case == "lds_batches":             # more workgroups than one resident set, at a lane-distributed and two LDS lengths
        for n, batch, dtype, flags in ((32, 70001, 0, 0), (512, 9001, 0, fa.HANN | fa.POWER), (4096, 1201, 0, fa.HANN | fa.POWER),
                                      (2048, 1501, 1, 0), (2048, 2203, 1, fa.HANN | fa.POWER), (8192, 700, 0, 0)):
            x = cplx((batch, n), np.complex64 if dtype == 0 else np.complex128)
            ref = np.fft.fft(x.astype(np.complex128) * (np.hanning(n) if flags & fa.HANN else 1.0), axis=-1)
            with fa.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
                y = p.exec_host(x)
            assert rel_l2(y, np.abs(ref) ** 2 if flags & fa.POWER else ref) < TOL[dtype], (n, batch)
    elif case == "padded_window":
        n, n_in = 1 << 17, 100001
        x = cplx((3, n_in), np.complex64)
        with fa.Plan(n, n_in=n_in, batch=3, dtype=0, flags=fa.HANN) as p:
            y = p.exec_host(x)
        assert rel_l2(y, np.fft.fft(x.astype(np.complex128) * np.hanning(n_in), n, axis=-1)) < TOL[0]
    elif case == "chirpz":
        n, n_in = 24016, 1501
        x = cplx((2, n_in), np.complex128)
        with fa.Plan(n, n_in=n_in, batch=2, dtype=1, flags=fa.HANN) as p:
            y = p.exec_host(x)
        assert rel_l2(y, np.fft.fft(x * np.hanning(n_in), n, axis=-1)) < TOL[1]
    elif case == "real_2p17":
        n = 1 << 17
        x = rng.standard_normal((2, n))
        with fa.Plan(n, batch=2, dtype=1, flags=fa.REAL_IN) as p:
            y = p.exec_host(x)
        assert rel_l2(y, np.fft.fft(x, axis=-1)) < TOL[1]
    elif case == "columns":
        n, inner = 4096, 256
        x = cplx((n, inner), np.complex64)
        with ColumnPlan(n, inner, dtype=0) as p:
            y = p.exec_host(x)
        assert rel_l2(y, np.fft.fft(x.astype(np.complex128), axis=0)) < TOL[0]
    else:                                   # the pipelined host path (pinned staging, copy threads)
        n = 4096
        x = cplx((6000, n), np.complex64)
        with fa.Plan(n, batch=6000, dtype=0, flags=fa.POWER) as p:
            y = p.exec_host(x)
        assert rel_l2(y[:50], np.abs(np.fft.fft(x[:50].astype(np.complex128), axis=-1)) ** 2) < TOL[0]
        assert rel_l2(y[-50:], np.abs(np.fft.fft(x[-50:].astype(np.complex128), axis=-1)) ** 2) < TOL[0]


def test_large_complex64_plans_on_sampled_bins(fa):
    """2^26 and 2^27 complex64 points (9,9,8 and 9,9,9: one and two twiddled 512-point passes): sampled bins
    against the DFT evaluated directly in float64, and Parseval."""
    import math
    import torch
    from test_gpu_fullsize import _direct_bins_1d, _energy
    torch.cuda.set_device(0)
    for lg in (26, 27):
        n = 1 << lg
        g = torch.Generator(device="cuda").manual_seed(lg)
        x = torch.empty(n, dtype=torch.complex64, device="cuda")
        torch.view_as_real(x).normal_(generator=g)
        y = torch.empty_like(x)
        with fa.Plan(n, dtype=fa.C64) as p:
            assert p.launches == 3
            p.exec(x, y, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        e_in = _energy(x)
        assert abs(_energy(y) / (n * e_in) - 1.0) < 1e-5
        ks = [0, 1, 511, 512, 513, n // 2 - 1, n // 2, n - 512, n - 1, 123456789 % n, 98765432 % n]
        got = y[torch.tensor(ks, device="cuda")].cpu().numpy()
        want = _direct_bins_1d(x, ks)
        assert np.max(np.abs(got - want)) / math.sqrt(e_in) < 2e-6, lg


def test_large_complex128_plans_on_sampled_bins(fa):
    """2^25 .. 2^27 complex128 points (512-point first / middle / last passes as the library itself picks
    them): sampled bins against the DFT evaluated directly in float64, and Parseval."""
    import math
    import torch
    from test_gpu_fullsize import _direct_bins_1d, _energy
    torch.cuda.set_device(0)
    for lg in (25, 26, 27):
        n = 1 << lg
        g = torch.Generator(device="cuda").manual_seed(lg)
        x = torch.empty(n, dtype=torch.complex128, device="cuda")
        torch.view_as_real(x).normal_(generator=g)
        y = torch.empty_like(x)
        with fa.Plan(n, dtype=fa.C128) as p:
            assert p.launches == 3
            p.exec(x, y, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        e_in = _energy(x)
        assert abs(_energy(y) / (n * e_in) - 1.0) < 1e-12
        ks = [0, 1, 511, 512, 513, n // 2 - 1, n // 2, n - 512, n - 1, 123456789 % n, 98765432 % n]
        got = y[torch.tensor(ks, device="cuda")].cpu().numpy()
        want = _direct_bins_1d(x, ks)
        assert np.max(np.abs(got - want)) / math.sqrt(e_in) < 1e-12, lg


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("log2n", [14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_tiled_pow2(fa, dtype, log2n):
    if dtype == 0 and log2n == 14:
        pytest.skip("single-kernel length for complex64")
    n = 1 << log2n
    batch = 3 if log2n <= 16 else 1
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), log2n + 10 * dtype, cdt)
    with fa.Plan(n, batch=batch, dtype=dtype) as p:
        assert p.launches >= 2
        y = p.exec_host(x)
    assert rel_l2(y, so.dft(x, engine="numpy")) < TOL[dtype]


def test_cfg3_2pow24_c64(fa):
    """BASELINE config 3: one complex64 transform of 2^24 points."""
    n = 1 << 24
    x = _rand((1, n), 24, np.complex64)
    with fa.Plan(n, dtype=0) as p:
        y = p.exec_host(x)
    ref = np.fft.fft(x.astype(np.complex128))
    assert rel_l2(y, ref) < 1e-5
    # size-independent properties at full size: Parseval and a shifted impulse
    e_in = float(np.sum(np.abs(x.astype(np.complex128)) ** 2))
    assert abs(float(np.sum(np.abs(y.astype(np.complex128)) ** 2)) / (n * e_in) - 1.0) < 1e-5


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n", [1, 3, 5, 6, 7, 12, 30, 100, 127, 1000, 1501, 8001, 24016, 100003])
def test_chirpz_arbitrary_length(fa, dtype, n):
    batch = 2
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), n + dtype, cdt)
    with fa.Plan(n, batch=batch, dtype=dtype) as p:
        y = p.exec_host(x)
    assert y.shape == (batch, n)
    ref = so.dft(x, engine="numpy")
    assert rel_l2(y, ref) < TOL[dtype]
    if n <= 1000:
        assert rel_l2(y[0], so.dft(x[0], engine="direct")) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
def test_app_lengths_window_pad_power_stats(fa, dtype):
    """The reference apps' own call shapes: N0 = 8001 -> 256032 (final2 defaults) and
    N0 = 1501 -> 24016 with demodulation (combined app defaults)."""
    cdt = np.complex64 if dtype == 0 else np.complex128
    for n_in, n, mu in ((8001, 256032, 0.0), (1501, 24016, -0.81)):
        dt = 0.01
        times = np.arange(n_in) * dt
        x = _rand((1, n_in), n_in, cdt)
        flags = fa.HANN | fa.POWER | fa.STATS | (fa.DEMOD if mu else 0)
        with fa.Plan(n, n_in=n_in, dtype=dtype, flags=flags) as p:
            S, stats = p.exec_host(x, scale=dt, mu=mu, t0=0.0, dt=dt)
        _, ref = so.windowed_power(x, n=n, use_hann=True, dt=dt, engine="numpy", mu=mu or None, times=times)
        assert rel_l2(S, ref) < TOL[dtype]
        tol = 1e-5 if dtype == 0 else 1e-11
        assert abs(stats[0, 0] - ref.max()) < tol * ref.max()
        assert abs(stats[0, 1] - ref.sum()) < tol * ref.sum()
        f = np.fft.fftfreq(n) * n
        assert abs(stats[0, 2] - (ref * f).sum()) < tol * np.abs(ref * f).sum()


def test_real_input_chirpz_and_one_point(fa):
    rng = np.random.default_rng(11)
    xr = rng.standard_normal((2, 1000))
    with fa.Plan(1000, batch=2, dtype=1, flags=fa.REAL_IN) as p:
        y = p.exec_host(xr)
    assert rel_l2(y, np.fft.fft(xr)) < 1e-12
    x1 = _rand((5, 1), 1, np.complex128)
    with fa.Plan(1, batch=5, dtype=1, flags=fa.POWER | fa.STATS) as p:
        S, st = p.exec_host(x1, scale=2.0)
    assert np.allclose(S[:, 0], np.abs(2.0 * x1[:, 0]) ** 2, rtol=1e-14)
    assert np.allclose(st[:, 0], S[:, 0]) and np.allclose(st[:, 2], 0.0)


@pytest.mark.parametrize("dtype", [0, 1])
def test_column_batched_plans(fa, dtype):
    """np.fft.fft(a, axis=-2) on the strided columns (no transpose pass), one to three tile
    passes, with and without the fused four-step twiddle."""
    cdt = np.complex64 if dtype == 0 else np.complex128
    wmin = 32 if dtype == 0 else 16
    for batch, n, inner in ((1, 16, wmin), (2, 256, wmin), (3, 1024, 64), (1, 16384, 64), (1, 1 << 19, wmin),
                            (2, 64, 2048)):
        x = _rand((batch, n, inner), n + inner + dtype, cdt)
        with fa.ColumnPlan(n, inner, batch=batch, dtype=dtype) as p:
            assert p.launches == (1 if n <= 512 else (2 if n <= 1 << 18 else 3))
            y = p.exec_host(x)
        ref = np.fft.fft(x.astype(np.complex128), axis=1)
        assert rel_l2(y, ref) < TOL[dtype], (batch, n, inner)
    # fused inter-step twiddle of the four-step algorithm
    n, inner, c0, n_total = 1024, 128, 384, 1024 * 4096
    x = _rand((1, n, inner), 99, cdt)
    with fa.ColumnPlan(n, inner, dtype=dtype, twiddle=(n_total, c0)) as p:
        y = p.exec_host(x)
    k = np.arange(n)[:, None].astype(np.float64)
    c = (c0 + np.arange(inner))[None, :].astype(np.float64)
    ref = np.fft.fft(x[0].astype(np.complex128), axis=0) * np.exp(-2j * np.pi * (k * c) / n_total)
    assert rel_l2(y[0], ref) < TOL[dtype]
    # the twiddle's roots are evaluated per thread (quarter-turn reduction + small-angle polynomial): moduli
    # from the degenerate ones up to 2^40, exponents k*c reduced exactly on the host
    n, inner = 256, wmin
    x = _rand((1, n, inner), 7, cdt)
    base = np.fft.fft(x[0].astype(np.complex128), axis=0)
    for n_total, c0 in ((1, 0), (2, 1), (4, 3), (1 << 12, 5), (1 << 31, (1 << 20) + 7), (1 << 40, (1 << 33) + 12345)):
        with fa.ColumnPlan(n, inner, dtype=dtype, twiddle=(n_total, c0)) as p:
            y = p.exec_host(x)
        kc = np.array([[(int(k) * (c0 + int(c))) % n_total for c in range(inner)] for k in range(n)], dtype=np.float64)
        ref = base * np.exp(-2j * np.pi * (kc / n_total))
        assert rel_l2(y[0], ref) < TOL[dtype], (n_total, c0)
    with pytest.raises(fa.MfftError):
        fa.ColumnPlan(n, inner, dtype=dtype, twiddle=(1 << 41, 0))
    # shapes the tile passes do not serve are refused, not mis-computed
    for n, inner in ((8, 64), (48, 64), (64, 8), (64, 48)):
        with pytest.raises(fa.MfftError):
            fa.ColumnPlan(n, inner, dtype=dtype)


def test_pipelined_host_path(fa, monkeypatch):
    """Large host batches go through pinned slabs (copy threads, two slabs in flight): results must
    equal the single-copy path bit for bit, ragged last slab and per-signal statistics included."""
    rng = np.random.default_rng(31)
    cases = [
        (1024, 1024, 8192 + 77, 0, 0),                                   # single kernel, complex64
        (1024, 700, 9000, 0, fa.HANN | fa.POWER | fa.STATS),             # zero padding + statistics
        (1000, 1000, 9000, 0, 0),                                        # chirp-z
        (1 << 16, 1 << 16, 131, 0, fa.REAL_IN | fa.POWER),               # tile passes, real input
        (4096, 4096, 1100, 1, fa.DEMOD | fa.HANN),                       # complex128, demodulation table
    ]
    for n, n_in, batch, dtype, flags in cases:
        rdt, cdt = (np.float32, np.complex64) if dtype == 0 else (np.float64, np.complex128)
        if flags & fa.REAL_IN:
            x = rng.standard_normal((batch, n_in)).astype(rdt)
        else:
            x = (rng.standard_normal((batch, n_in), dtype=np.float32)
                 + 1j * rng.standard_normal((batch, n_in), dtype=np.float32)).astype(cdt)
        with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
            monkeypatch.setenv("MFFT_NO_PIPELINE", "1")
            ref = p.exec_host(x, scale=0.5, mu=0.3, t0=0.1, dt=0.01)
            monkeypatch.delenv("MFFT_NO_PIPELINE")
            for _ in range(2):                                           # second call reuses the pinned slabs
                got = p.exec_host(x, scale=0.5, mu=0.3, t0=0.1, dt=0.01)
                if flags & fa.STATS:
                    assert np.array_equal(got[0], ref[0]) and np.allclose(got[1], ref[1], rtol=1e-12)
                else:
                    assert np.array_equal(got, ref), (n, n_in, batch, flags)
            if flags & fa.STATS:
                st = p.exec_host(x, scale=0.5, stats_only=True)
                assert np.allclose(st, ref[1], rtol=1e-12)


def test_pipelined_host_path_concurrent_callers(fa, monkeypatch):
    """Several threads run pipelined host transfers at once, each on its own plan: the copy threads
    are shared (one job at a time per direction), so every caller must still get exactly its own
    result -- bit for bit the single-copy path's."""
    import threading
    rng = np.random.default_rng(77)
    shapes = [(1024, 8192 + 5, 0), (2048, 4096 + 3, 0), (512, 9000, 1), (1000, 8200, 0)]
    inputs, refs, plans = [], [], []
    monkeypatch.setenv("MFFT_NO_PIPELINE", "1")
    for n, batch, dtype in shapes:
        cdt = np.complex64 if dtype == 0 else np.complex128
        x = (rng.standard_normal((batch, n), dtype=np.float32)
             + 1j * rng.standard_normal((batch, n), dtype=np.float32)).astype(cdt)
        p = fa.Plan(n, batch=batch, dtype=dtype)
        inputs.append(x); plans.append(p); refs.append(p.exec_host(x, scale=2.0))
    monkeypatch.delenv("MFFT_NO_PIPELINE")
    errors = []

    def work(i):
        try:
            for _ in range(3):
                got = plans[i].exec_host(inputs[i], scale=2.0)
                if not np.array_equal(got, refs[i]):
                    errors.append(f"caller {i}: result differs")
        except Exception as e:                                            # noqa: BLE001
            errors.append(f"caller {i}: {e!r}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(shapes))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    [p.destroy() for p in plans]
    assert not errors, errors


@pytest.mark.parametrize("dtype", [0, 1])
def test_in_place_complex_transforms(fa, dtype):
    """d_out == d_in is allowed for dense complex -> complex transforms of full-length input: every
    route reads a signal (or writes its first pass to the workspace) before it stores."""
    import torch
    torch.cuda.set_device(0)
    cdt = np.complex64 if dtype == 0 else np.complex128
    stream = torch.cuda.current_stream().cuda_stream
    for n, batch in ((64, 300), (1024, 37), (8192, 5), (1 << 16, 3), (1 << 20, 1), (1000, 4), (24016, 1)):
        x = _rand((batch, n), n + 17 * dtype, cdt)
        ref = np.fft.fft(x.astype(np.complex128))
        d = torch.from_numpy(x).cuda()
        with fa.Plan(n, batch=batch, dtype=dtype) as p:
            p.exec(d, d, stream=stream)
            torch.cuda.synchronize()
        assert rel_l2(d.cpu().numpy(), ref) < TOL[dtype], (n, batch)
    # windowed in place
    n, batch = 2048, 9
    x = _rand((batch, n), 5, cdt)
    d = torch.from_numpy(x).cuda()
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.HANN) as p:
        p.exec(d, d, stream=stream)
        torch.cuda.synchronize()
    assert rel_l2(d.cpu().numpy(), np.fft.fft(x.astype(np.complex128) * np.hanning(n))) < TOL[dtype]


@pytest.mark.parametrize("n", [1 << 16, 24016])
def test_one_plan_two_streams_two_threads(fa, n):
    """A cached plan is shared by every thread of the process.  Multi-pass and chirp-z plans keep
    per-execution state (pass workspaces, the convolution buffer), so executions issued from two
    threads on two streams must be ordered by the library on the device, not only on the host."""
    import threading
    import torch
    torch.cuda.set_device(0)
    batch = 4
    plan = fa.Plan(n, batch=batch, dtype=fa.C128)
    xs = [_rand((batch, n), 40 + i, np.complex128) for i in range(2)]
    refs = [np.fft.fft(x) for x in xs]
    errs = []

    def work(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                d_x = torch.from_numpy(xs[i]).cuda()
                d_y = torch.empty_like(d_x)
                for _ in range(25):
                    plan.exec(d_x, d_y, stream=st.cuda_stream)
                st.synchronize()
                errs.append(rel_l2(d_y.cpu().numpy(), refs[i]))
        except Exception as e:                                            # noqa: BLE001
            errs.append(repr(e))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    plan.destroy()
    assert len(errs) == 2 and all(isinstance(e, float) and e < 1e-12 for e in errs), errs


def test_device_buffers_are_checked_against_the_plan(fa):
    import torch
    torch.cuda.set_device(0)
    with fa.Plan(256, batch=4, dtype=fa.C64, flags=fa.POWER | fa.STATS) as p:
        x = torch.zeros((4, 256), dtype=torch.complex64, device="cuda")
        s = torch.zeros((4, 256), dtype=torch.float32, device="cuda")
        st = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
        p.exec(x, s, d_stats=st)                                                   # the good case
        with pytest.raises(ValueError, match="elements"):
            p.exec(x[:3], s, d_stats=st)
        with pytest.raises(ValueError, match="dtype"):
            p.exec(x.to(torch.complex128), s, d_stats=st)
        with pytest.raises(ValueError, match="dtype"):
            p.exec(x, s.double(), d_stats=st)
        with pytest.raises(ValueError, match="contiguous"):
            p.exec(x.t(), s, d_stats=st)
        with pytest.raises(ValueError, match="device"):
            p.exec(x.cpu(), s, d_stats=st)
        with pytest.raises(ValueError, match="elements"):
            p.exec(x, s, d_stats=st[:2])
        torch.cuda.synchronize()


def test_column_plans_on_random_shapes(fa):
    """Seeded slice of tools/fuzz_columns.py: column-batched plans of random length / inner extent / batch,
    with and without the fused twiddle (moduli up to 2^40), against numpy in float64."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_columns as fz
    del fz.failures[:]
    assert fz.run(901, 902, 25, verbose=False) >= 45
    assert not fz.failures, fz.failures[:3]


@pytest.mark.parametrize("dtype,split,log2n,batch", [
    (0, "10,5", 16, 5), (0, "10,6", 17, 3), (0, "9,7", 17, 2), (0, "8,8", 17, 4), (0, "7,9", 17, 3), (0, "6,10", 17, 2),
    (0, "8,8,4", 21, 1), (0, "8,8,8", 25, 1),
    (1, "11,4", 16, 3), (1, "10,5", 16, 2), (1, "9,6", 16, 5), (1, "8,7", 16, 2), (1, "7,8", 16, 3), (1, "6,9", 16, 2),
    (1, "5,10", 16, 4), (1, "4,11", 16, 2), (1, "10,10", 21, 25), (1, "8,8,7", 24, 1)])
def test_fused_real_input_last_pass_at_every_radix(fa, monkeypatch, dtype, split, log2n, batch):
    """np.fft.fft of a real signal (final2.py:142-143 with a real f_t) where the split of the half-length transform into
    the signal's own bins is fused into the last pass: every last-pass radix and tile width of both precisions (the
    complex128 1024-point tiles in their four- and eight-column forms), one and several signals, complex and power
    output, against numpy in float64; the route with a separate split kernel gives the same bins."""
    monkeypatch.setenv("MFFT_SPLIT", split)           # of the half-length plan
    n = 1 << log2n
    rdt = np.float32 if dtype == 0 else np.float64
    tol = TOL[dtype]
    rng = np.random.default_rng(log2n + 100 * dtype + batch)
    x = rng.standard_normal((batch, n)).astype(rdt)
    power = batch >= 25                                # the large case: real output
    ref = np.fft.fft(x.astype(np.float64), axis=-1)
    if power:
        ref = np.abs(ref) ** 2
    flags = fa.REAL_IN | (fa.POWER if power else 0)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
        y = p.exec_host(x)
    assert rel_l2(y, ref) < tol, rel_l2(y, ref)
    for b in range(batch):                             # per signal, and the bins of the self-mirrored columns on their own
        assert rel_l2(y[b], ref[b]) < tol
    m = n // 2
    ns = m >> int(split.split(",")[-1])
    special = np.concatenate([np.arange(0, m + 1, ns), np.arange(ns // 2, m, ns)])
    special = np.concatenate([special, (n - special[(special > 0) & (special < m)])])
    assert rel_l2(y[:, special], ref[:, special]) < tol
    if not power:
        assert np.array_equal(y[:, 1:m], np.conj(y[:, :m:-1]))
        monkeypatch.setenv("MFFT_R2C_SPLIT_KERNEL", "1")
        with fa.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
            y2 = p.exec_host(x)
        monkeypatch.delenv("MFFT_R2C_SPLIT_KERNEL")
        assert rel_l2(y2, ref) < tol and rel_l2(y2, y) < tol


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("log2n", [14, 15, 16, 17, 19, 20, 22])
def test_real_input_above_one_lds_residency_packed_route(fa, dtype, log2n, monkeypatch):
    """np.fft.fft of real float32 / float64 signals (final2.py:142-143 with a real f_t) at lengths whose
    half-length transform still takes several passes: the packed route (n/2 complex samples, then the split
    into the n bins), against numpy in float64 -- plain, batched, and with the apps' own flag set (Hann over
    zero-padded samples, power spectrum, statistics); and against the route that promotes to complex."""
    n = 1 << log2n
    rdt = np.float32 if dtype == 0 else np.float64
    tol = TOL[dtype]
    rng = np.random.default_rng(log2n + 40 * dtype)
    batch = 3 if log2n <= 17 else 1
    x = rng.standard_normal((batch, n)).astype(rdt)
    ref = np.fft.fft(x.astype(np.float64), axis=-1)
    packed = log2n >= (16 if dtype == 0 else 15)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.REAL_IN) as p:
        y = p.exec_host(x, scale=0.5)
        launches = p.launches
    assert y.dtype == (np.complex64 if dtype == 0 else np.complex128)
    assert rel_l2(y, 0.5 * ref) < tol, (n, dtype)
    # Hermitian symmetry of a real signal's spectrum is exact on the packed route (both halves from one value)
    if packed:
        assert np.array_equal(y[:, 1:n // 2], np.conj(y[:, :n // 2:-1]))
    n_in = n - 1000
    xw = x[:, :n_in].astype(np.float64) * np.hanning(n_in)
    ref_s = np.abs(0.25 * np.fft.fft(xw, n=n, axis=-1)) ** 2
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=fa.REAL_IN | fa.HANN | fa.POWER | fa.STATS) as p:
        S, st = p.exec_host(np.ascontiguousarray(x[:, :n_in]), scale=0.25)
    assert rel_l2(S, ref_s) < tol
    f = np.fft.fftfreq(n) * n
    stol = 1e-5 if dtype == 0 else 1e-11
    np.testing.assert_allclose(st[:, 0], ref_s.max(axis=1), rtol=stol)
    np.testing.assert_allclose(st[:, 1], ref_s.sum(axis=1), rtol=stol)
    np.testing.assert_allclose(st[:, 2], (ref_s * f).sum(axis=1), rtol=0, atol=stol * np.abs(ref_s * f).sum(axis=1).max())
    # odd sample counts keep the promoting route; so does MFFT_NO_R2C=1 (read when the plan is built)
    with fa.Plan(n, n_in=n_in - 1, batch=batch, dtype=dtype, flags=fa.REAL_IN) as p:
        y_odd = p.exec_host(np.ascontiguousarray(x[:, :n_in - 1]))
    assert rel_l2(y_odd, np.fft.fft(x[:, :n_in - 1].astype(np.float64), n=n, axis=-1)) < tol
    monkeypatch.setenv("MFFT_NO_R2C", "1")
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.REAL_IN) as p:
        y2 = p.exec_host(x, scale=0.5)
        assert launches >= 1 and p.launches >= 1
    monkeypatch.delenv("MFFT_NO_R2C")
    assert rel_l2(y2, 0.5 * ref) < tol
    # the numpy-compatible entry point takes the same route
    assert rel_l2(fa.fft(x[0]), ref[0]) < tol
