"""The entry point that allocates nothing: mfft_lsq_normal_static (device arrays in and out,
partial sums in a fixed device array inside the library) against the existing
mfft_lsq_normal_host on the same inputs, through the C ABI.  Two calls in a row, the second with
a smaller input, so a value left in the shared scratch by the first call would show.
Tolerance: 1e-12 relative (float64 reductions in a different summation order)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_reference(lib, check, x, win, times, mu, E):
    m = E.shape[0]
    G = np.empty((m, m), dtype=np.complex128)
    r = np.empty(m, dtype=np.complex128)
    yy = ctypes.c_double(0.0)
    check(lib.mfft_lsq_normal_host(x.ctypes.data, win.ctypes.data if win is not None else None, times.ctypes.data,
                                   x.shape[0], float(mu), E.ctypes.data, int(m), G.ctypes.data, r.ctypes.data,
                                   ctypes.byref(yy)))
    return G, r, yy.value


def _case(lib, check, torch, n, batch, m, use_win, use_mu, seed, poison):
    rng = np.random.default_rng(seed)
    times = np.arange(n) * 0.01
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    win = np.hanning(n) if use_win else None
    E = np.ascontiguousarray(rng.uniform(-4.0, 4.0, size=(batch, m)))
    mu = rng.uniform(-1.0, 1.0, size=batch) if use_mu else np.zeros(batch)
    nres = 2 * m * m + 2 * m + 1
    d_x = torch.from_numpy(x).cuda()
    d_t = torch.from_numpy(times).cuda()
    d_w = torch.from_numpy(win).cuda() if use_win else None
    d_E = torch.from_numpy(E).cuda()
    d_mu = torch.from_numpy(mu).cuda() if use_mu else None
    d_out = torch.full((batch, nres), poison, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    check(lib.mfft_lsq_normal_static(d_x.data_ptr(), d_w.data_ptr() if use_win else None, d_t.data_ptr(), n, batch,
                                     d_mu.data_ptr() if use_mu else None, d_E.data_ptr(), m, d_out.data_ptr(),
                                     ctypes.c_void_p(stream)))
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    for b in range(batch):
        G, r, yy = _host_reference(lib, check, x[b], win, times, mu[b], E[b])
        Gs = out[b, :2 * m * m].view(np.complex128).reshape(m, m)
        rs = out[b, 2 * m * m:2 * m * m + 2 * m].view(np.complex128)
        assert np.linalg.norm(Gs - G) <= 1e-12 * np.linalg.norm(G), (n, batch, m, b)
        assert np.linalg.norm(rs - r) <= 1e-12 * max(np.linalg.norm(r), np.sqrt(yy * G[0, 0].real)), (n, batch, m, b)
        assert abs(out[b, -1] - yy) <= 1e-12 * yy
        # and directly against numpy: B^H B, B^H y, y^H y of the reference's basis (final2.py:174-175)
        w = win if win is not None else np.ones(n)
        basis = np.exp(-1j * np.outer(times, E[b])) * w[:, None]
        y = x[b] * w * np.exp(1j * mu[b] * times)
        nb, ny = np.linalg.norm(basis), np.linalg.norm(y)
        assert np.linalg.norm(Gs - basis.conj().T @ basis) <= 1e-12 * nb * nb, (n, batch, m, b)
        assert np.linalg.norm(rs - basis.conj().T @ y) <= 1e-12 * nb * ny, (n, batch, m, b)
        assert abs(out[b, -1] - ny * ny) <= 1e-12 * ny * ny
    return out


def test_static_scratch_entry_point_matches_allocating_one():
    import torch
    from fft_amd import _lib
    torch.cuda.set_device(0)
    lib, check = _lib.load(), _lib.check
    # first call: many sample blocks, several column chunks (m > 16), window and demodulation
    _case(lib, check, torch, n=40000, batch=3, m=20, use_win=True, use_mu=True, seed=1, poison=float("nan"))
    # second call, smaller in every extent: fewer samples, signals and columns
    a = _case(lib, check, torch, n=1501, batch=2, m=6, use_win=True, use_mu=False, seed=2, poison=float("nan"))
    # and again: identical bits (fixed summation order, nothing carried over from the call before)
    _case(lib, check, torch, n=8001, batch=5, m=64, use_win=False, use_mu=True, seed=3, poison=123.0)
    b = _case(lib, check, torch, n=1501, batch=2, m=6, use_win=True, use_mu=False, seed=2, poison=-1.0)
    assert np.array_equal(a, b)


def test_static_scratch_limits_are_refused():
    import torch
    from fft_amd import _lib
    torch.cuda.set_device(0)
    lib = _lib.load()
    d = torch.zeros(64, dtype=torch.float64, device="cuda")
    p = d.data_ptr()
    # m = 64 needs 64*4*36 doubles per signal and sample block: 29 signals exceed the 2 MiB scratch
    assert lib.mfft_lsq_normal_static(p, None, p, 8, 29, None, p, 64, p, None) == _lib.E_UNSUPPORTED
    assert b"scratch" in lib.mfft_last_error()
    # ... and 28 signals fit (buffers of the real extents: the call computes)
    big_x = torch.zeros(28 * 8 * 2, dtype=torch.float64, device="cuda")
    big_E = torch.zeros(28 * 64, dtype=torch.float64, device="cuda")
    big_out = torch.zeros(28 * (2 * 64 * 64 + 2 * 64 + 1), dtype=torch.float64, device="cuda")
    assert lib.mfft_lsq_normal_static(big_x.data_ptr(), None, p, 8, 28, None, big_E.data_ptr(), 64, big_out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    # batch counts whose products would wrap 64-bit arithmetic are refused by the batch bound, not divided by
    for huge in (1 << 62, (1 << 63) - 1, 1 << 40, 65536):
        assert lib.mfft_lsq_normal_static(p, None, p, 8, huge, None, p, 64, p, None) == _lib.E_UNSUPPORTED
        assert b"split the batch" in lib.mfft_last_error()
    assert lib.mfft_lsq_normal_static(p, None, p, 8, 1, None, p, 65, p, None) == _lib.E_INVALID
    assert lib.mfft_lsq_normal_static(p, None, p, 0, 1, None, p, 4, p, None) == _lib.E_INVALID
    assert lib.mfft_lsq_normal_static(None, None, p, 8, 1, None, p, 4, p, None) == _lib.E_INVALID
