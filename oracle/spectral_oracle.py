"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the spectral hot path (numpy side).

Nothing under ``fft_amd/`` may import this module.  It is imported by ``tests/``,
by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``,
always as the checker and never as the thing shipped or measured.

It restates, on the CPU, what the reference computes on its spectral path:

* ``hanning``            np.hanning(M)                      /root/reference/final2.py:139
* ``fftfreq``            np.fft.fftfreq(n, d)               /root/reference/final2.py:144
* ``dft``                np.fft.fft(x, n=Nfft)              /root/reference/final2.py:143
* ``power_spectrum``     dt*F, |F|^2, S/max                 /root/reference/final2.py:143-146
* ``fft_from_timeseries``                                   /root/reference/final2.py:115-188
* ``fft_extract_E_p``                                       /root/reference/streamlit_app_combined.py:190-251
* ``heff_block`` / ``return_amplitude_block`` (input generator)
                                                            /root/reference/final2.py:47-64, :85-97
* ``return_amplitude_top``                                  /root/reference/streamlit_app_combined.py:86-93, :166-169
* ``lanczos_ab``  (consumer of the path's output, for end-to-end checks)
                                                            /root/reference/final2.py:203-246

The transform itself has two interchangeable engines:

``engine="c"``      oracle/pocket_dft.c -- a plain-C restatement of pocketfft's
                    published algorithm (mixed radix + Bluestein), built into
                    oracle/liboracle.so by oracle/Makefile.
``engine="numpy"``  numpy.fft.fft -- the third-party routine (numpy's bundled
                    pocketfft, numpy 2.2.6 in this image) that the reference
                    itself calls.  It is not part of /root/reference, so calling
                    it here copies nothing.

tests/test_oracle.py pins both engines against each other, against an O(n^2)
long-double DFT, and against the golden vectors under tests/golden/ that were
produced by running the reference's own functions (tests/golden/make_golden.py).
The reference has no tests of its own; those generated fixtures are the pin.
"""
from __future__ import annotations

import ctypes
import itertools
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        for name in ("odft_forward_c128", "odft_forward_c128_bluestein", "odft_direct_ld"):
            fn = getattr(lib, name)
            fn.argtypes = [dp, ctypes.c_int64, dp, ctypes.c_int64]
            fn.restype = ctypes.c_int
        lib.opower_c128.argtypes = [dp, ctypes.c_double, dp, ctypes.c_int64]
        lib.opower_c128.restype = None
        _LIB = lib
    return _LIB


# --------------------------------------------------------------------------- window / axis

def hanning(M: int) -> np.ndarray:
    """Symmetric Hann window, float64: w[n] = 0.5 + 0.5*cos(pi*(2n - M + 1)/(M - 1)).

    Equals 0.5 - 0.5*cos(2*pi*n/(M-1)); endpoints are 0, M == 1 gives [1.0], M < 1
    gives an empty array (np.hanning's contract)."""
    M = int(M)
    if M < 1:
        return np.array([], dtype=np.float64)
    if M == 1:
        return np.ones(1, dtype=np.float64)
    n = np.arange(1 - M, M, 2)
    return 0.5 + 0.5 * np.cos(np.pi * n / (M - 1))


def fftfreq(n: int, d: float = 1.0) -> np.ndarray:
    """Sample frequencies in FFT order: k/(n*d) for k < ceil(n/2), (k-n)/(n*d) after."""
    n = int(n)
    k = np.arange(n, dtype=np.int64)
    k[(n - 1) // 2 + 1:] -= n
    return k * (1.0 / (n * d))


# --------------------------------------------------------------------------- transform

def dft(x, n=None, engine: str = "c") -> np.ndarray:
    """Forward unnormalised DFT of the last axis, zero-padded/truncated to n, in
    complex128 (the precision class every parity comparison is made in)."""
    x = np.asarray(x)
    n_in = x.shape[-1]
    n = n_in if n is None else int(n)
    if n < 1:
        raise ValueError(f"Invalid number of FFT data points ({n}) specified.")
    if engine == "numpy":
        return np.fft.fft(x.astype(np.complex128, copy=False), n=n, axis=-1)
    if engine == "direct":
        fn = _lib().odft_direct_ld
    elif engine == "bluestein":
        fn = _lib().odft_forward_c128_bluestein
    elif engine == "c":
        fn = _lib().odft_forward_c128
    else:
        raise ValueError(engine)
    xin = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1, n_in)
    out = np.empty((xin.shape[0], n), dtype=np.complex128)
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(xin.shape[0]):
        rc = fn(xin[b].ctypes.data_as(dp), n_in, out[b].ctypes.data_as(dp), n)
        if rc != 0:
            raise RuntimeError(f"oracle transform failed rc={rc}")
    return out.reshape(x.shape[:-1] + (n,))


def windowed_power(x, n=None, use_hann=True, dt=1.0, engine="numpy", mu=None, times=None):
    """The fused chain the GPU path replaces, on a batch: window, optional
    demodulation by exp(+i*mu*t), zero-padded forward DFT, scale by dt, |.|^2.
    Returns (F, S) with F = dt * DFT and S = |F|^2, both in float64 precision."""
    x = np.asarray(x)
    m = x.shape[-1]
    xw = x.astype(np.complex128) * (hanning(m) if use_hann else 1.0)
    if mu is not None and mu != 0.0:
        xw = xw * np.exp(1j * mu * np.asarray(times, float))
    F = dt * dft(xw, n=n, engine=engine)
    S = F.real ** 2 + F.imag ** 2
    return F, S


def reference_numpy_chain(x, n=None):
    """Exactly the numpy calls the reference makes, in the input's own precision
    (numpy 2.x keeps complex64 in, complex64 out).  This is the CPU path that
    bench.py times as cpu_baseline: np.hanning, multiply, np.fft.fft, np.abs()**2
    (/root/reference/final2.py:139-145), one thread."""
    x = np.asarray(x)
    win = np.hanning(x.shape[-1]).astype(x.real.dtype)
    F = np.fft.fft(x * win, n=n, axis=-1)
    return np.abs(F) ** 2


# --------------------------------------------------------------------------- input generators

def heff_block(K, J, m):
    """L x L effective block for bottom excitation m: diagonal S for i == m, else
    S - 2(K[i] + K[m]) with S = sum(K); J on the two off-diagonals."""
    K = np.asarray(K, float)
    J = np.asarray(J, float)
    L = len(K)
    S = float(K.sum())
    d = S - 2.0 * (K + K[m])
    d[m] = S
    H = np.diag(d)
    idx = np.arange(L - 1)
    H[idx, idx + 1] = J
    H[idx + 1, idx] = J
    return H


def _return_amplitude(H, site, times):
    w, V = np.linalg.eigh(H)
    w = w.real
    psi0 = np.zeros(H.shape[0], complex)
    psi0[site] = 1.0
    c0 = V.conj().T @ psi0
    # f(t) = sum_k V[site,k] * exp(-i w_k t) * c0[k], evaluated per time like the app
    times = np.asarray(times, float)
    phases = np.exp(-1j * np.outer(times, w))
    return (phases * c0[None, :]) @ V[site, :]


def return_amplitude_block(K, J, bottom_m, site, times):
    return _return_amplitude(heff_block(K, J, bottom_m), site, times)


def return_amplitude_top(Htop, site, times):
    return _return_amplitude(np.asarray(Htop), site, times)


# --------------------------------------------------------------------------- estimator A

def _log_parabola(x, y, i):
    if i <= 0 or i >= len(y) - 1:
        return x[i]
    y0, y1, y2 = (np.log(max(y[i + s], 1e-300)) for s in (-1, 0, 1))
    den = 2.0 * (y0 - 2.0 * y1 + y2)
    if den == 0:
        return x[i]
    return x[i] + (y0 - y2) / den * 0.5 * (x[i + 1] - x[i - 1])


def fft_from_timeseries(f_t, times, n_levels=3, pad_factor=16, use_hann=True, include_dc=False,
                        dc_guard=1.5, post_prune=True, min_weight=1e-6, keep_top=None,
                        guard_mult=1.2, engine="numpy"):
    times = np.asarray(times, float)
    f_t = np.asarray(f_t)
    if times.ndim != 1 or times.size < 2:
        raise ValueError("`times` must be 1D with length >= 2.")
    if f_t.shape[0] != times.shape[0]:
        raise ValueError("`f_t` and `times` must have the same length.")
    steps = np.diff(times)
    if not np.allclose(steps, steps[0]):
        raise ValueError("`times` must be uniformly spaced.")
    dt = float(steps[0])

    win = hanning(len(f_t)) if use_hann else np.ones_like(f_t)
    xw = f_t * win
    nfft = int(max(1, pad_factor) * len(xw))
    F = dt * dft(xw, n=nfft, engine=engine)
    omega = 2.0 * np.pi * fftfreq(nfft, d=dt)
    S = np.abs(F) ** 2
    top = S.max()
    S_norm = S / (top if top > 0 else 1.0)

    span = times[-1] - times[0]
    domega = (2 * np.pi) / (span + 1e-12)
    inner = S[1:-1]
    loc = np.nonzero((inner > S[:-2]) & (inner >= S[2:]))[0] + 1
    if include_dc:
        if S[0] > 0:
            loc = np.unique(np.concatenate(([0], loc)))
    else:
        loc = loc[np.abs(omega[loc]) > dc_guard * domega]
    if loc.size == 0:
        return np.array([]), np.array([]), -omega, S_norm

    ranked = loc[np.argsort(S[loc])[::-1]]
    guard = guard_mult * domega
    chosen = []
    for k in ranked:
        if all(abs(omega[k] - omega[j]) > guard for j in chosen):
            chosen.append(k)
        if len(chosen) >= max(n_levels, 1):
            break
    E_hat = -np.array([_log_parabola(omega, S, i) for i in chosen])

    A = np.column_stack([np.exp(-1j * E * times) * win for E in E_hat])
    c = np.linalg.lstsq(A, xw, rcond=None)[0]
    p = np.clip(np.real(c), 0.0, None)
    if p.sum() > 0:
        p /= p.sum()
    if post_prune:
        keep = p > min_weight
        E_hat, p = E_hat[keep], p[keep]
        if keep_top is not None and E_hat.size > keep_top:
            best = np.argsort(p)[::-1][:keep_top]
            E_hat, p = E_hat[best], p[best]
        if p.sum() > 0:
            p /= p.sum()
    order = np.argsort(E_hat)
    return E_hat[order], p[order], -omega, S_norm


# --------------------------------------------------------------------------- estimator B

def fft_extract_E_p(Htop, site, times, n_peaks=3, pad=16, use_hann=True, demod=True,
                    engine="numpy"):
    f_t = return_amplitude_top(Htop, site, times)
    times = np.asarray(times, float)
    dt = float(times[1] - times[0])
    w = hanning(len(f_t)) if use_hann else np.ones_like(f_t)
    xw = f_t * w

    n0 = len(xw)
    nfft0 = int(pad * n0)
    F0 = dt * dft(xw, n=nfft0, engine=engine)
    omega0 = 2.0 * np.pi * fftfreq(nfft0, d=dt)
    S0 = np.abs(F0) ** 2
    mu = float(np.sum(omega0 * S0) / np.sum(S0)) if (demod and S0.sum() > 0) else 0.0
    if demod and mu != 0.0:
        xw = xw * np.exp(+1j * mu * times)

    nfft = int(pad * len(xw))
    F = dt * dft(xw, n=nfft, engine=engine)
    omega = 2.0 * np.pi * fftfreq(nfft, d=dt)
    S = np.abs(F) ** 2

    inner = S[1:-1]
    loc = np.nonzero((inner > S[:-2]) & (inner >= S[2:]))[0] + 1
    if loc.size == 0:
        return np.array([]), np.array([]), omega, S, mu
    ranked = loc[np.argsort(S[loc])][::-1]
    tol = (2 * np.pi) / (times[-1] - times[0] + 1e-12)
    chosen = []
    for k in ranked:
        if len(chosen) >= max(n_peaks * 3, n_peaks):
            break
        if all(abs(abs(omega[k]) - abs(omega[j])) > 0.6 * tol for j in chosen):
            chosen.append(k)
    chosen = sorted(chosen, key=lambda j: S[j], reverse=True)[:n_peaks]
    if len(chosen) == 0:
        return np.array([]), np.array([]), omega, S, mu
    mags = np.sort(np.unique(np.round(np.abs(omega[chosen]), 12)))[:n_peaks]

    best_res, best_E, best_p = np.inf, None, None
    for signs in itertools.product([-1.0, +1.0], repeat=len(mags)):
        E_try = [s * m for s, m in zip(signs, mags)]
        A = np.column_stack([np.exp(-1j * E * times) * w for E in E_try])
        c = np.linalg.lstsq(A, xw, rcond=None)[0]
        res = np.linalg.norm(A @ c - xw)
        p_est = np.clip(np.real(c), 0.0, None)
        if p_est.sum() > 0:
            p_est /= p_est.sum()
        if res < best_res:
            best_res, best_E, best_p = res, np.array(E_try), p_est
    E_phys = best_E - mu
    order = np.argsort(E_phys)
    return E_phys[order], best_p[order], omega, S, mu


# --------------------------------------------------------------------------- consumer

def lanczos_ab(E, p):
    """Three-term (Stieltjes) recursion on (E, p): returns the Jacobi diagonal a and
    off-diagonal b that the reference's lanczos_detailed produces."""
    E = np.asarray(E, float)
    p = np.asarray(p, float)
    if p.sum() <= 0:
        raise ValueError("Weights p must have positive sum.")
    p = p / p.sum()
    n = len(E)
    h_prev = np.zeros_like(E)
    h = np.sqrt(np.clip(p, 0, None))
    b_prev = 0.0
    a, b = [], []
    for i in range(n):
        a_i = float(np.sum(E * h ** 2))
        r = E * h - a_i * h - b_prev * h_prev
        a.append(a_i)
        if i < n - 1:
            b_i = float(np.linalg.norm(r))
            b.append(abs(b_i))
            h_prev, h = h, r / (b_i if b_i > 1e-15 else 1.0)
            b_prev = b[-1]
    return np.array(a), np.array(b)
