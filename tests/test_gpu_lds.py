"""GPU parity: the LDS-resident Stockham kernels (through the C ABI) against the
CPU oracle on the same seeded inputs.  Tolerances are BASELINE.json's: relative
L2 <= 1e-5 for complex64, <= 1e-12 for complex128, both measured against the
oracle evaluated in float64."""
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


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("log2n", list(range(1, 15)))
def test_plain_fft_pow2(fa, dtype, log2n):
    n = 1 << log2n
    if dtype == 1 and log2n > 13:
        pytest.skip("complex128 single-kernel range ends at 8192")
    batch = 7 if n >= 2048 else 37          # ragged against the slots-per-workgroup
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), n + dtype, cdt)
    with fa.Plan(n, batch=batch, dtype=dtype) as p:
        y = p.exec_host(x)
    assert y.dtype == cdt and y.shape == (batch, n)
    ref = so.dft(x, engine="numpy")
    assert rel_l2(y, ref) < TOL[dtype]
    # one row against the plain-C engine as well
    assert rel_l2(y[0], so.dft(x[0], engine="c")) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 128, 256, 1024, 4096, 8192, 16384])
def test_hann_power_fused(fa, dtype, n):
    batch = 33 if n <= 4096 else 5
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), 3 * n + dtype, cdt)
    dt = 0.01
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.HANN | fa.POWER) as p:
        S = p.exec_host(x, scale=dt)
    assert S.dtype == (np.float32 if dtype == 0 else np.float64)
    _, ref = so.windowed_power(x, use_hann=True, dt=dt, engine="numpy")
    assert rel_l2(S, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
def test_zero_pad_window_demod_stats(fa, dtype):
    n_in, n, batch = 1501, 4096, 3
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n_in), 99 + dtype, cdt)
    dt, mu = 0.01, 0.37
    times = np.arange(n_in) * dt
    flags = fa.HANN | fa.POWER | fa.DEMOD | fa.STATS
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
        S, stats = p.exec_host(x, scale=dt, mu=mu, t0=0.0, dt=dt)
    _, ref = so.windowed_power(x, n=n, use_hann=True, dt=dt, engine="numpy", mu=mu, times=times)
    assert rel_l2(S, ref) < TOL[dtype]
    f = np.fft.fftfreq(n) * n
    tol = 1e-5 if dtype == 0 else 1e-11
    np.testing.assert_allclose(stats[:, 0], ref.max(axis=1), rtol=tol)
    np.testing.assert_allclose(stats[:, 1], ref.sum(axis=1), rtol=tol)
    np.testing.assert_allclose(stats[:, 2], (ref * f).sum(axis=1), rtol=0,
                               atol=tol * np.abs(ref * f).sum(axis=1).max())
    # complex output of the same prologue
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=fa.HANN | fa.DEMOD) as p:
        F = p.exec_host(x, scale=dt, mu=mu, t0=0.0, dt=dt)
    Fref, _ = so.windowed_power(x, n=n, use_hann=True, dt=dt, engine="numpy", mu=mu, times=times)
    assert rel_l2(F, Fref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
def test_real_input_and_inverse(fa, dtype):
    n, batch = 4096, 2
    rdt = np.float32 if dtype == 0 else np.float64
    rng = np.random.default_rng(5)
    xr = rng.standard_normal((batch, n)).astype(rdt)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.REAL_IN) as p:
        y = p.exec_host(xr)
    assert rel_l2(y, np.fft.fft(xr.astype(np.float64))) < TOL[dtype]
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), 6, cdt)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.INVERSE) as p:
        y = p.exec_host(x)
    assert rel_l2(y, np.fft.ifft(x.astype(np.complex128)) * n) < TOL[dtype]


def test_known_answers_on_device(fa):
    n = 1024
    imp = np.zeros((1, n), np.complex128); imp[0, 1] = 1.0
    with fa.Plan(n, dtype=1) as p:
        y = p.exec_host(imp)
        assert np.max(np.abs(y[0] - np.exp(-2j * np.pi * np.arange(n) / n))) < 1e-14
        lin_a = _rand((1, n), 1, np.complex128); lin_b = _rand((1, n), 2, np.complex128)
        ya, yb, yab = p.exec_host(lin_a), p.exec_host(lin_b), p.exec_host(2 * lin_a - 3j * lin_b)
        assert rel_l2(yab, 2 * ya - 3j * yb) < 1e-14
        # Parseval
        assert abs(np.sum(np.abs(ya) ** 2) - n * np.sum(np.abs(lin_a) ** 2)) < 1e-11 * n * np.sum(np.abs(lin_a) ** 2)


def test_argument_errors(fa):
    with pytest.raises(ValueError):
        fa.Plan(0)
    with pytest.raises(ValueError):
        fa.Plan(16, n_in=32)
    with pytest.raises(ValueError):
        fa.Plan(16, batch=0)
    with pytest.raises(ValueError):
        fa.Plan(16, flags=fa.STATS)
    with fa.Plan(16) as p:
        with pytest.raises(ValueError):
            p.exec_host(np.zeros(8, np.complex64))


@pytest.mark.parametrize("dtype", [0, 1])
def test_layouts_through_every_route(fa, dtype):
    """Segmented / strided device layouts: the fast segmented variant (power-of-two segments of at
    least n/16 elements), the staged route (segments combined with a window, zero padding or a
    chirp-z length) and the tile passes' own segmented load / store."""
    import torch
    import dist_helpers as dh
    cdt = np.complex64 if dtype == 0 else np.complex128
    tdt = torch.complex64 if dtype == 0 else torch.complex128
    cases = [
        # n, n_in, batch, flags, in_layout, out_layout
        (1024, 1024, 5, 0, (256, 256, 5 * 256), (128, 128, 5 * 128)),       # fast segmented variant
        (16384 >> dtype, 16384 >> dtype, 3, 0, None, (2048, 2048, 3 * 2048)),
        # segments shorter than a row of n/16 elements (column groups of the multi-GPU schedules): in, out, both,
        # at the lengths of the LDS-DMA pipeline and below it
        (16384 >> dtype, 16384 >> dtype, 5, 0, (128, 128, 5 * 128), None),
        (16384 >> dtype, 16384 >> dtype, 5, 0, None, (256, 256, 7 * 256)),
        (8192, 8192, 6, 0, (256 >> dtype, 256 >> dtype, 6 * 256), (128, 128, 9 * 128)),
        (8192 >> dtype, 8192 >> dtype, 4, 0, (128, 128, 4 * 128), (128, 128, 4 * 128)),
        (4096, 4096, 7, 0, (128, 128, 7 * 128), (128, 128, 8 * 128)),
        (2048, 2048, 3, 0, (64 << (1 - dtype), 64 << (1 - dtype), 3 * 128 + 20), (128, 128, 3 * 128)),
        (4096, 4096, 2, 0, (8192, 0, 0), (5000, 0, 0)),                     # strided batches only
        (1024, 1024, 5, fa.HANN, (256, 256, 5 * 256), (32, 32, 5 * 32)),    # window + segments: staged
        (2048, 1500, 3, 0, (500, 500, 3 * 500), (64, 64, 3 * 64)),          # zero padding + segments
        (256, 256, 9, 0, (8, 8, 9 * 8), (4, 4, 9 * 4)),                     # segments below n/16
        (1000, 1000, 4, 0, (250, 250, 4 * 250), (100, 100, 4 * 100)),       # chirp-z length
        (1 << 16, 1 << 16, 2, 0, (1 << 14, 1 << 14, 2 << 14), (1 << 13, 1 << 13, 2 << 13)),   # tile passes
    ]
    for n, n_in, batch, flags, lin, lout in cases:
        idx_in = dh.layout_index(n_in, batch, lin, n_in)
        idx_out = dh.layout_index(n, batch, lout, n)
        x = _rand((batch, n_in), n + n_in + dtype, cdt)
        buf_in = np.zeros(int(idx_in.max()) + 1, dtype=cdt)
        buf_in[idx_in] = x
        d_in = torch.from_numpy(buf_in).cuda()
        d_out = torch.full((int(idx_out.max()) + 1,), complex(7.0, -7.0), dtype=tdt, device="cuda")
        with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags, in_layout=lin, out_layout=lout) as p:
            p.exec(d_in, d_out)
            torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        xw = x * np.hanning(n_in) if flags & fa.HANN else x
        ref = np.fft.fft(xw.astype(np.complex128), n=n, axis=-1)
        tol = TOL[dtype]
        assert rel_l2(got[idx_out], ref) < tol, (n, n_in, flags, lin, lout)
        untouched = np.ones(got.shape[0], bool)
        untouched[idx_out.ravel()] = False
        assert np.all(got[untouched] == complex(7.0, -7.0)), "stored outside the layout"


@pytest.mark.parametrize("dtype", [0, 1])
def test_short_lengths_both_kernel_families(fa, dtype, monkeypatch):
    """Lengths up to 128 / 64 points run lane-distributed by default; MFFT_NO_WAVE=1 (read when the
    plan is built) selects the LDS-resident kernel for the same lengths.  Both must meet the bar,
    for the plain, window-only, power-only and fused Hann + power variants, on ragged batches."""
    cdt = np.complex64 if dtype == 0 else np.complex128
    for n in (2, 4, 8, 16, 32, 64, 128):
        for batch in (1, 5, 63, 1000):
            x = _rand((batch, n), n * batch + dtype, cdt)
            ref = np.fft.fft(x.astype(np.complex128))
            for no_wave in (False, True):
                if no_wave:
                    monkeypatch.setenv("MFFT_NO_WAVE", "1")
                else:
                    monkeypatch.delenv("MFFT_NO_WAVE", raising=False)
                with fa.Plan(n, batch=batch, dtype=dtype) as p:
                    assert rel_l2(p.exec_host(x, scale=2.0), 2.0 * ref) < TOL[dtype], (n, batch, no_wave)
                with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.POWER) as p:         # power only
                    S = p.exec_host(x, scale=0.5)
                assert rel_l2(S, np.abs(0.5 * ref) ** 2) < TOL[dtype], (n, batch, no_wave)
                if n >= 4:
                    xw = x.astype(np.complex128) * np.hanning(n)
                    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.HANN | fa.POWER) as p:
                        S = p.exec_host(x, scale=0.5)
                    assert rel_l2(S, np.abs(0.5 * np.fft.fft(xw)) ** 2) < TOL[dtype], (n, batch, no_wave)
                    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.HANN) as p:      # window only
                        F = p.exec_host(x, scale=0.5)
                    assert rel_l2(F, 0.5 * np.fft.fft(xw)) < TOL[dtype], (n, batch, no_wave)
    monkeypatch.delenv("MFFT_NO_WAVE", raising=False)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("log2n", list(range(8, 15)))
def test_window_only_and_power_only(fa, dtype, log2n):
    """Full-length dense transforms with only a window (complex out) or only the power epilogue:
    their own fast variants of the LDS-resident kernel (tile passes beyond its range)."""
    n = 1 << log2n
    batch = 9 if n >= 2048 else 41
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), 7 * n + dtype, cdt)
    xw = x.astype(np.complex128) * np.hanning(n)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.HANN) as p:
        F = p.exec_host(x, scale=0.25)
    assert F.dtype == cdt and rel_l2(F, 0.25 * np.fft.fft(xw)) < TOL[dtype]
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.POWER) as p:
        S = p.exec_host(x, scale=0.25)
    assert rel_l2(S, np.abs(0.25 * np.fft.fft(x.astype(np.complex128))) ** 2) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_real_input_full_length(fa, dtype, n):
    """np.fft.fft on real arrays (numpy promotes them to complex): real samples are read as they
    are, 4 / 8 bytes per point."""
    if dtype == 1 and n > 8192:
        pytest.skip("complex128 single-kernel range ends at 8192")
    rdt = np.float32 if dtype == 0 else np.float64
    batch = 6 if n >= 2048 else 35
    xr = np.random.default_rng(n + dtype).standard_normal((batch, n)).astype(rdt)
    with fa.Plan(n, batch=batch, dtype=dtype, flags=fa.REAL_IN) as p:
        y = p.exec_host(xr, scale=3.0)
    assert rel_l2(y, 3.0 * np.fft.fft(xr.astype(np.float64))) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("log2n", list(range(2, 15)))
def test_power_with_statistics_full_length(fa, dtype, log2n):
    """Dense full-length power spectra with per-signal statistics (the apps' own flag set,
    final2.py:139-146, and the same without the window): max, sum and signed-bin centroid."""
    n = 1 << log2n
    batch = 7 if n >= 2048 else 45
    cdt = np.complex64 if dtype == 0 else np.complex128
    x = _rand((batch, n), 11 * n + dtype, cdt)
    f = np.fft.fftfreq(n) * n
    tol = 1e-5 if dtype == 0 else 1e-11
    for flags, xin in ((fa.HANN | fa.POWER | fa.STATS, x.astype(np.complex128) * np.hanning(n)),
                       (fa.POWER | fa.STATS, x.astype(np.complex128))):
        ref = np.abs(0.5 * np.fft.fft(xin)) ** 2
        with fa.Plan(n, batch=batch, dtype=dtype, flags=flags) as p:
            S, st = p.exec_host(x, scale=0.5)
        assert rel_l2(S, ref) < TOL[dtype], (n, flags)
        np.testing.assert_allclose(st[:, 0], ref.max(axis=1), rtol=tol)
        np.testing.assert_allclose(st[:, 1], ref.sum(axis=1), rtol=tol)
        np.testing.assert_allclose(st[:, 2], (ref * f).sum(axis=1), rtol=0, atol=tol * np.abs(ref * f).sum(axis=1).max())


@pytest.mark.parametrize("dtype", [0, 1])
def test_strided_block_copy_and_pitched_column_plans(fa, dtype):
    """mfft_swap_axes_strided (one column group packed out of rows of the full width) against numpy
    indexing, and mfft_plan_1d_columns_ld (a block of columns transformed straight into its place in a
    wider array) against np.fft.fft(axis=0) -- the two local pieces of the column-group schedules."""
    import ctypes
    import torch
    from fft_amd import _lib
    from fft_amd.engine import ColumnPlan
    lib = _lib.load()
    cdt = np.complex64 if dtype == 0 else np.complex128
    tdt = torch.complex64 if dtype == 0 else torch.complex128
    for a, b, w, sa, sb, off in ((12, 4, 64, 4 * 4 * 64, 64, 2 * 4 * 64), (5, 3, 7, 100, 9, 11), (64, 8, 256, 8192, 256, 4096),
                                 (1, 1, 1, 1, 1, 0), (33, 2, 16, 32, 16, 0)):
        total = off + (a - 1) * sa + (b - 1) * sb + w
        x = _rand(total, a * b + w + dtype, cdt)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((a * b * w + 3,), complex(7.0, -7.0), dtype=tdt, device="cuda")
        _lib.check(lib.mfft_swap_axes_strided(d_in.data_ptr() + off * x.itemsize, d_out.data_ptr(), a, b, w, sa, sb, dtype, None))
        torch.cuda.synchronize()
        idx = off + np.arange(b)[:, None, None] * sb + np.arange(a)[None, :, None] * sa + np.arange(w)[None, None, :]
        got = d_out.cpu().numpy()
        assert np.array_equal(got[:a * b * w], x[idx].reshape(-1)) and np.all(got[a * b * w:] == complex(7.0, -7.0))
    wmin = 32 if dtype == 0 else 16
    for n, inner, ld, c0 in ((256, wmin, 4 * wmin, wmin), (4096, 2 * wmin, 8 * wmin, 0), (16384 >> dtype, 64, 256, 128),
                             (1 << 16, 32, 96, 64), (64, wmin, wmin, 0)):
        x = _rand((n, inner), n + inner + dtype, cdt)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((n * ld,), complex(7.0, -7.0), dtype=tdt, device="cuda")
        with ColumnPlan(n, inner, dtype=dtype, out_ld=ld) as p:
            p.exec(d_in, d_out[c0:])
            torch.cuda.synchronize()
        got = d_out.cpu().numpy().reshape(n, ld)
        assert rel_l2(got[:, c0:c0 + inner], np.fft.fft(x.astype(np.complex128), axis=0)) < TOL[dtype], (n, inner, ld)
        rest = np.ones((n, ld), bool)
        rest[:, c0:c0 + inner] = False
        assert np.all(got[rest] == complex(7.0, -7.0)), "stored outside the column block"
    h = ctypes.c_void_p()
    assert lib.mfft_plan_1d_columns_ld(ctypes.byref(h), 256, 64, 1, dtype, 0, 0, 32) == _lib.E_INVALID       # ld below the width
    assert lib.mfft_plan_1d_columns_ld(ctypes.byref(h), 256, 64, 1, dtype, 1 << 20, 0, 128) == _lib.E_UNSUPPORTED   # with the twiddle
