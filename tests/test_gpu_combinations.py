"""Seeded sweep over combinations of length class, zero padding, precision and fused options
through the C ABI, against the numpy restatement of the same chain evaluated in float64:
window (np.hanning), demodulation exp(+i mu t), zero-padded forward DFT, scale, power, and the
max / sum / centroid statistics (final2.py:139-146, streamlit_app_combined.py:199-215)."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _reference(x, n, flags, fa, scale, mu, t0, dt):
    n_in = x.shape[-1]
    xw = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    if flags & fa.HANN:
        xw = xw * np.hanning(n_in)
    if flags & fa.DEMOD:
        xw = xw * np.exp(+1j * mu * (t0 + dt * np.arange(n_in)))
    if flags & fa.INVERSE:
        F = np.fft.ifft(xw, n=n, axis=-1) * n
    else:
        F = np.fft.fft(xw, n=n, axis=-1)
    F = scale * F
    return np.abs(F) ** 2 if flags & fa.POWER else F


LENGTHS = [  # (n, n_in): single-kernel, tile passes, chirp-z; with and without zero padding
    (2, 2), (8, 5), (64, 64), (256, 100), (1024, 1024), (1024, 1), (4096, 4095), (8192, 8192), (8192, 5001), (16384, 3000),
    (1 << 15, 1 << 15), (1 << 16, 12345), (1 << 19, 1 << 19), (1 << 20, 70001),
    (3, 3), (17, 17), (1000, 1000), (1000, 37), (24016, 1501), (65537, 65537), (256032, 8001),
]


@pytest.mark.parametrize("dtype", [0, 1])
def test_option_combinations(dtype):
    import fft_amd as fa
    rng = np.random.default_rng(2025 + dtype)
    all_flags = [0, fa.HANN, fa.POWER, fa.HANN | fa.POWER, fa.POWER | fa.STATS, fa.HANN | fa.POWER | fa.STATS,
                 fa.DEMOD, fa.HANN | fa.DEMOD | fa.POWER | fa.STATS, fa.REAL_IN, fa.REAL_IN | fa.HANN | fa.POWER | fa.STATS,
                 fa.REAL_IN | fa.DEMOD]
    rdt, cdt = (np.float32, np.complex64) if dtype == 0 else (np.float64, np.complex128)
    checked = 0
    for n, n_in in LENGTHS:
        pow2 = n & (n - 1) == 0
        for flags in [all_flags[i] for i in rng.choice(len(all_flags), size=4, replace=False)] + \
                     ([fa.INVERSE] if pow2 else []):
            batch = int(rng.integers(1, 4)) if n > 4096 else int(rng.integers(1, 40))
            if flags & fa.REAL_IN:
                x = rng.standard_normal((batch, n_in)).astype(rdt)
            else:
                x = (rng.standard_normal((batch, n_in)) + 1j * rng.standard_normal((batch, n_in))).astype(cdt)
            scale, mu, t0, dt = 0.37, -0.81, 0.25, 0.01
            ref = _reference(x, n, flags, fa, scale, mu, t0, dt)
            with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
                out = p.exec_host(x, scale=scale, mu=mu, t0=t0, dt=dt)
            stats = None
            if flags & fa.STATS:
                out, stats = out
            tol = 1e-5 if dtype == 0 else 1e-12        # BASELINE.json's bars, also for chirp-z lengths and |F|^2
            assert out.shape == ref.shape and rel_l2(out, ref) < tol, (n, n_in, batch, flags, rel_l2(out, ref))
            if stats is not None:
                f = np.fft.fftfreq(n) * n
                stol = 1e-5 if dtype == 0 else 1e-11
                np.testing.assert_allclose(stats[:, 0], ref.max(axis=-1), rtol=stol)
                np.testing.assert_allclose(stats[:, 1], ref.sum(axis=-1), rtol=stol)
                np.testing.assert_allclose(stats[:, 2], (ref * f).sum(axis=-1), rtol=0,
                                           atol=stol * np.abs(ref * f).sum(axis=-1).max())
            checked += 1
    assert checked >= 80
