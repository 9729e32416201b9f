"""Drop-in replacements for the reference's two spectral estimators.

``fft_from_timeseries``  same signature, return tuple, dtypes, FFT-order axis and
                         ``ValueError``s as /root/reference/final2.py:115-188
``fft_extract_E_p``      same as /root/reference/streamlit_app_combined.py:190-251

What runs where.  The window, zero padding, demodulation, forward transform,
``dt`` scaling, power spectrum and the max / centroid reductions run on the GPU
as one fused plan (final2.py:139-146, streamlit_app_combined.py:199-215).  So does
the peak pipeline on the device copy of the spectrum -- local-maximum scan, DC
guard, ranking by height, guard-spaced greedy pick and log-parabolic refinement
(final2.py:151-171, streamlit_app_combined.py:218-228) -- and the tall reductions
of the tail: the return-amplitude generator f(t) (final2.py:85-97) and the normal
equations of the windowed least squares, including all 2^k fits of the sign search
(final2.py:174-175, streamlit_app_combined.py:233-246).  ``fft_from_timeseries_batch``
does all of that for several signals (the app's blocks m) in one device pass.  What
stays on the host is O(#lines) scalar work: k x k solves, pruning, sorting, and the
eigendecomposition of the L <= 10 block.  The float64 spectrum comes back once
because both functions return it.

There is no CPU fallback for the transform: without libmfft.so or without a GPU
these functions raise.
"""
from __future__ import annotations

import ctypes
import threading
from collections import OrderedDict

import numpy as np

from . import _lib
from .npfft import cached_plan, engine_dtype, fftfreq

TWO_PI = 2.0 * np.pi

_axis_cache: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
_axis_lock = threading.Lock()


def _omega_axis(nfft: int, dt: float) -> np.ndarray:
    """2*pi*fftfreq(nfft, d=dt) (final2.py:144), kept for the few (nfft, dt) pairs an app
    session cycles through: three passes over nfft doubles otherwise repeated on every
    call.  The cached array is read-only; callers hand out copies (or its negation)."""
    key = (int(nfft), float(dt))
    with _axis_lock:
        ax = _axis_cache.get(key)
        if ax is not None:
            _axis_cache.move_to_end(key)
            return ax
    ax = TWO_PI * fftfreq(nfft, d=dt)
    ax.setflags(write=False)
    with _axis_lock:
        _axis_cache[key] = ax
        while len(_axis_cache) > 8:
            _axis_cache.popitem(last=False)
    return ax


# --------------------------------------------------------------------------- device part

def power_spectrum(x, nfft: int, dt: float, use_hann: bool = True, mu: float = 0.0, t0: float = 0.0,
                   want_spectrum: bool = True, want_peaks: bool = False):
    """|dt * FFT_nfft(window * x * exp(i mu t))|^2 on the GPU, in double precision.

    Returns ``(S, smax, ssum, centroid_bins)`` where ``S`` is float64[nfft] (``None``
    when ``want_spectrum`` is false), ``smax = S.max()``, ``ssum = S.sum()`` and
    ``centroid_bins = sum(f_k * S)`` with ``f_k`` the signed FFT-order bin index.  With
    ``want_peaks`` a fifth item follows: the ascending indices of the interior local maxima
    of S (left strict, right weak), found on the device."""
    x = np.asarray(x)
    _, real_in = engine_dtype(x)
    flags = _lib.POWER | _lib.STATS
    if use_hann:
        flags |= _lib.HANN
    if mu != 0.0:
        flags |= _lib.DEMOD
    if real_in:
        flags |= _lib.REAL_IN
    plan = cached_plan(nfft, x.shape[0], 1, _lib.C128, flags)
    xin = np.ascontiguousarray(x, dtype=np.float64 if real_in else np.complex128)
    if want_peaks:
        # a zero-padded spectrum has about one local maximum per side lobe, i.e. ~n_in of them
        S, stats, idx = plan.exec_host_peaks(xin, scale=dt, mu=mu, t0=t0, dt=dt,
                                             capacity=max(65536, 2 * x.shape[0]))
        if not isinstance(idx, np.ndarray):   # more than that (noise-like spectrum): rerun with room for all
            S, stats, idx = plan.exec_host_peaks(xin, scale=dt, mu=mu, t0=t0, dt=dt, capacity=int(idx))
        return S, float(stats[0, 0]), float(stats[0, 1]), float(stats[0, 2]), idx
    if want_spectrum:
        S, stats = plan.exec_host(xin, scale=dt, mu=mu, t0=t0, dt=dt)
    else:
        S, stats = None, plan.exec_host(xin, scale=dt, mu=mu, t0=t0, dt=dt, stats_only=True)
    return S, float(stats[0, 0]), float(stats[0, 1]), float(stats[0, 2])


# --------------------------------------------------------------------------- host tail

_hann_cache: "OrderedDict[int, np.ndarray]" = OrderedDict()


def _uniform(steps: np.ndarray) -> bool:
    """np.allclose(steps, steps[0]) (final2.py:133): its own inequality |a - b| <= atol + rtol |b| evaluated directly,
    which settles every finite grid in a third of the time; anything else (nan, inf) goes to np.allclose itself."""
    s0 = steps[0]
    with np.errstate(invalid="ignore"):
        worst = np.abs(steps - s0).max()
    if worst <= 1e-8 + 1e-5 * abs(s0):          # finite by construction: a nan or inf step fails this comparison
        return True
    return bool(np.allclose(steps, s0))


def _hanning(m: int) -> np.ndarray:
    """np.hanning(m) (final2.py:139), kept read-only for the few lengths a session uses."""
    with _axis_lock:
        w = _hann_cache.get(m)
        if w is not None:
            _hann_cache.move_to_end(m)
            return w
    w = np.hanning(m)
    w.setflags(write=False)
    with _axis_lock:
        _hann_cache[m] = w
        while len(_hann_cache) > 8:
            _hann_cache.popitem(last=False)
    return w


def _window_of(f_t, use_hann):
    return _hanning(len(f_t)) if use_hann else np.ones_like(f_t)


def _normal_equations(x, win, times, E_cols, mu=0.0):
    """(G, r, yy) of the windowed line fit, reduced on the GPU (mfft_lsq_normal_host):
    G = B^H B, r = B^H y, yy = y^H y for B[:, a] = win * exp(-i E_a t), y = x * win * exp(+i mu t)."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    t = np.ascontiguousarray(times, dtype=np.float64)
    E = np.ascontiguousarray(E_cols, dtype=np.float64)
    w = None if win is None else np.ascontiguousarray(win, dtype=np.float64)
    m = E.shape[0]
    G = np.empty((m, m), dtype=np.complex128)
    r = np.empty(m, dtype=np.complex128)
    yy = ctypes.c_double(0.0)
    _lib.check(_lib.load().mfft_lsq_normal_host(
        x.ctypes.data, w.ctypes.data if w is not None else None, t.ctypes.data, x.shape[0], float(mu),
        E.ctypes.data, int(m), G.ctypes.data, r.ctypes.data, ctypes.byref(yy)))
    return G, r, yy.value


def _solve_normal(G, r, yy, cols):
    """Coefficients, clipped/normalised weights and residual norm of the fit that uses the
    columns `cols`, from the normal equations G c = r.  The k x k system is solved in the
    minimum-norm least-squares sense, which is also what lstsq on the N0 x k basis returns when
    columns coincide (c = B^+ y = (B^H B)^+ B^H y)."""
    Gs = G[np.ix_(cols, cols)]
    rs = r[cols]
    coef = np.linalg.lstsq(Gs, rs, rcond=1e-12)[0]
    p = np.clip(coef.real, 0.0, None)
    total = p.sum()
    if total > 0:
        p = p / total
    res2 = yy - 2.0 * np.real(np.vdot(coef, rs)) + np.real(np.vdot(coef, Gs @ coef))
    return coef, p, float(np.sqrt(max(res2, 0.0)))


def _batch_weights(gram, counts, want):
    """Clipped, normalised weights of every block of a batched call (what `_solve_normal` returns for
    each): blocks with the same number of lines are solved together by one stacked Hermitian
    pseudo-inverse -- the minimum-norm least-squares solution of G c = r with the same relative
    cut-off (1e-12 of the largest singular value) that lstsq applies -- instead of one lstsq call
    per block (45 us each, most of it numpy's call overhead)."""
    gram = np.asarray(gram)
    G_all = gram[:, :2 * want * want].view(np.complex128).reshape(-1, want, want)
    r_all = gram[:, 2 * want * want:2 * want * want + 2 * want].view(np.complex128)
    out = [None] * len(counts)
    for k in sorted(set(counts)):
        if k == 0:
            continue
        idx = [b for b, c in enumerate(counts) if c == k]
        Gs = G_all[idx][:, :k, :k]
        rs = r_all[idx][:, :k]
        coef = np.einsum("bij,bj->bi", np.linalg.pinv(Gs, rcond=1e-12, hermitian=True), rs)
        p = np.clip(coef.real, 0.0, None)
        tot = p.sum(axis=1, keepdims=True)
        p = np.divide(p, tot, out=p.copy(), where=tot > 0)
        for row, b in enumerate(idx):
            out[b] = p[row]
    return out


def _sign_search(G, r, yy, k, chunk=4096):
    """The 2^k sign search of streamlit_app_combined.py:239-246 on the normal equations of the 2k
    columns exp(-i(+|w_i|)t) (first k) and exp(-i(-|w_i|)t) (last k): for every sign pattern, in the
    order itertools.product((-1, +1), repeat=k) visits them, the minimum-norm fit on its k columns;
    returns (signs, weights) of the first pattern with the smallest residual.  All patterns of a
    chunk are solved in one stacked pseudo-inverse (Hermitian k x k blocks) instead of one lstsq
    call each."""
    best = (np.inf, None, None)
    total = 1 << k
    shifts = np.arange(k - 1, -1, -1)
    base = np.arange(k)
    for lo in range(0, total, chunk):
        idx = np.arange(lo, min(total, lo + chunk))
        plus = ((idx[:, None] >> shifts) & 1).astype(bool)            # bit i set: sign +1 at position i
        cols = np.where(plus, base, k + base)                         # (m, k)
        Gs = G[cols[:, :, None], cols[:, None, :]]                    # (m, k, k)
        rs = r[cols]                                                  # (m, k)
        coef = np.einsum("mij,mj->mi", np.linalg.pinv(Gs, rcond=1e-12, hermitian=True), rs)
        res2 = yy - 2.0 * np.real(np.einsum("mi,mi->m", coef.conj(), rs)) \
            + np.real(np.einsum("mi,mij,mj->m", coef.conj(), Gs, coef))
        res2 = np.maximum(res2, 0.0)                                  # the reference compares sqrt(max(., 0))
        j = int(np.argmin(res2))
        if res2[j] < best[0]:
            p = np.clip(coef[j].real, 0.0, None)
            tot = p.sum()
            best = (float(res2[j]), np.where(plus[j], 1.0, -1.0), p / tot if tot > 0 else p)
    return best[1], best[2]


# --------------------------------------------------------------------------- estimator A

def fft_from_timeseries_batch(
    f_ts,
    times,
    n_levels=3,
    pad_factor=16,
    use_hann=True,
    include_dc=False,
    dc_guard=1.5,
    post_prune=True,
    min_weight=1e-6,
    keep_top=None,
    guard_mult=1.2,
):
    """``fft_from_timeseries`` for several signals on one time grid -- the app's loop over the
    blocks m (final2.py:752-758) as ONE device pass: one upload, one batched plan, the peak
    pipeline and the normal equations of every signal on the device, one download.

    ``f_ts``: sequence of equally long signals, or an array of shape (B, N0).  Returns a list of
    B tuples ``(E_hat ascending, p, -omega, S_norm)``, each what the single-signal function
    returns for that signal.  The axis ``-omega`` is the same for every signal and is ONE array
    shared by the tuples; the ``S_norm`` are rows of one (B, Nfft) array."""
    times = np.asarray(times, float)
    f_ts = [np.asarray(f) for f in f_ts]
    if times.ndim != 1 or times.size < 2:
        raise ValueError("`times` must be 1D with length >= 2.")
    for f in f_ts:
        if f.ndim < 1 or f.shape[0] != times.shape[0]:
            raise ValueError("`f_t` and `times` must have the same length.")
    steps = np.diff(times)
    if not _uniform(steps):
        raise ValueError("`times` must be uniformly spaced.")
    dt = float(steps[0])
    if not f_ts:
        return []
    n0 = times.shape[0]
    nfft = int(max(1, pad_factor) * n0)
    want = max(n_levels, 1)
    if want > 64:
        raise ValueError("at most 64 spectral lines can be fitted at once")
    B = len(f_ts)
    x = np.empty((B, n0), dtype=np.complex128)
    for b, f in enumerate(f_ts):
        x[b] = f

    span = times[-1] - times[0]
    domega = TWO_PI / (span + 1e-12)
    flags = _lib.POWER | _lib.STATS | (_lib.HANN if use_hann else 0)
    plan = cached_plan(nfft, n0, B, _lib.C128, flags)
    res = plan.exec_host_lines(x, times, axis_dt=dt, guard=guard_mult * domega, want=want,
                               dc_guard=dc_guard * domega, include_dc=bool(include_dc), mode=0, scale=dt,
                               want_spectrum=True, want_fit=True, normalize=True)
    return _tuples_from_lines(res, nfft, dt, want, post_prune, min_weight, keep_top)


def _tuples_from_lines(res, nfft, dt, want, post_prune, min_weight, keep_top):
    """The host tail of estimator A for every signal of a device pass: weights from the normal equations,
    pruning, sorting (final2.py:176-188) -> [(E_hat ascending, p, -omega, S_norm)]."""
    B = len(res["count"])
    E_axis = -_omega_axis(nfft, dt)                # one fresh array per call, shared by the batch's tuples
    counts = [int(c) for c in res["count"]]
    weights = _batch_weights(res["gram"], counts, want) if B > 1 else None
    out = []
    for b in range(B):
        S_norm = res["S"][b]                       # divided by its guarded maximum on the device
        k = counts[b]
        if k == 0:
            out.append((np.array([]), np.array([]), E_axis, S_norm))
            continue
        E_hat = -res["omega_hat"][b, :k]
        if weights is not None:
            p = weights[b]
        else:
            g = res["gram"][b]
            G = g[:2 * want * want].view(np.complex128).reshape(want, want)
            r = g[2 * want * want:2 * want * want + 2 * want].view(np.complex128)
            _, p, _ = _solve_normal(G, r, float(g[-1]), np.arange(k))
        if post_prune:
            keep = p > min_weight
            E_hat, p = E_hat[keep], p[keep]
            if keep_top is not None and E_hat.size > keep_top:
                top = np.argsort(p)[::-1][:keep_top]
                E_hat, p = E_hat[top], p[top]
            if p.sum() > 0:
                p = p / p.sum()
        order = np.argsort(E_hat)
        out.append((E_hat[order], p[order], E_axis, S_norm))
    return out


def fft_from_timeseries(
    f_t,
    times,
    n_levels=3,
    pad_factor=16,
    use_hann=True,
    include_dc=False,
    dc_guard=1.5,
    post_prune=True,
    min_weight=1e-6,
    keep_top=None,
    guard_mult=1.2,
):
    """Spectral lines (E, p) of a return amplitude f(t); see final2.py:115-188.

    Returns ``(E_hat ascending, p, -omega, S_norm)``; the last two are full-length
    arrays in FFT order.  With no admissible peak the first two are empty."""
    times = np.asarray(times, float)
    f_t = np.asarray(f_t)
    if times.ndim != 1 or times.size < 2:
        raise ValueError("`times` must be 1D with length >= 2.")
    if f_t.shape[0] != times.shape[0]:
        raise ValueError("`f_t` and `times` must have the same length.")
    return fft_from_timeseries_batch([f_t], times, n_levels=n_levels, pad_factor=pad_factor, use_hann=use_hann,
                                     include_dc=include_dc, dc_guard=dc_guard, post_prune=post_prune,
                                     min_weight=min_weight, keep_top=keep_top, guard_mult=guard_mult)[0]


def fft_from_timeseries_blocks(
    K,
    J,
    m_list,
    site,
    times,
    n_levels=3,
    pad_factor=16,
    use_hann=True,
    include_dc=False,
    dc_guard=1.5,
    post_prune=True,
    min_weight=1e-6,
    keep_top=None,
    guard_mult=1.2,
    hbar=1.0,
):
    """The app's loop over blocks (final2.py:752-758, :853-859)

        for m in m_list:
            f_t = return_amplitude_block_general(K, J, m, site=site, times=times)
            ... = fft_from_timeseries(f_t, times, ...)

    as ONE device pass that never brings f(t) to the host: the L x L blocks are diagonalised here
    (L <= 10), each block's return amplitude is generated on the device straight into the batched
    transform's input (final2.py:85-97), and the spectra, the peak pipeline and the normal equations
    follow as in `fft_from_timeseries_batch`.  Only the time grid goes up.  Returns one
    ``(E_hat ascending, p, -omega, S_norm)`` per entry of ``m_list``, each what
    ``fft_from_timeseries(return_amplitude_block_general(K, J, m, site, times), times, ...)`` returns."""
    times = np.asarray(times, float)
    if times.ndim != 1 or times.size < 2:
        raise ValueError("`times` must be 1D with length >= 2.")
    steps = np.diff(times)
    if not _uniform(steps):
        raise ValueError("`times` must be uniformly spaced.")
    dt = float(steps[0])
    m_list = list(m_list)
    if not m_list:
        return []
    L = len(np.asarray(K, float))
    if L > 32:
        raise ValueError("blocks larger than 32 x 32 are not supported by the device generator")
    B = len(m_list)
    w = np.empty((B, L))
    c0 = np.empty((B, L), complex)
    vrow = np.empty((B, L), complex)
    for b, m in enumerate(m_list):
        ev, V = np.linalg.eigh(Heff_block_general(K, J, m))      # final2.py:81-83
        w[b] = ev.real
        c0[b] = V.conj().T[:, site]                              # V^H |site>
        vrow[b] = V[site, :]
    n0 = times.shape[0]
    nfft = int(max(1, pad_factor) * n0)
    want = max(n_levels, 1)
    if want > 64:
        raise ValueError("at most 64 spectral lines can be fitted at once")
    domega = TWO_PI / (times[-1] - times[0] + 1e-12)
    flags = _lib.POWER | _lib.STATS | (_lib.HANN if use_hann else 0)
    plan = cached_plan(nfft, n0, B, _lib.C128, flags)
    res = plan.exec_host_lines_blocks(w, c0, vrow, times, hbar=hbar, axis_dt=dt, guard=guard_mult * domega, want=want,
                                      dc_guard=dc_guard * domega, include_dc=bool(include_dc), mode=0, scale=dt,
                                      want_spectrum=True, want_fit=True, normalize=True)
    return _tuples_from_lines(res, nfft, dt, want, post_prune, min_weight, keep_top)


# --------------------------------------------------------------------------- estimator B

def _eigen_data(H, site):
    """(w, c0, vrow) of the return amplitude <site| exp(-i H t) |site> = sum_k vrow[k] exp(-i w[k] t) c0[k]:
    eigenvalues, V^H |site> and V[site, :] of the small Hermitian block (host: L <= 10 in the apps)."""
    H = np.asarray(H)
    if H.ndim != 2 or H.shape[0] != H.shape[1]:
        raise ValueError("H must be a square matrix")
    if H.shape[0] > 32:
        raise ValueError("blocks larger than 32 x 32 are not supported by the device generator")
    w, V = np.linalg.eigh(H)
    w = np.ascontiguousarray(w.real, dtype=np.float64)
    psi0 = np.zeros(H.shape[0], complex)
    psi0[site] = 1.0
    c0 = np.ascontiguousarray(V.conj().T @ psi0, dtype=np.complex128)
    vrow = np.ascontiguousarray(V[site, :], dtype=np.complex128)
    return w, c0, vrow


def _amplitude_on_device(H, site, times, hbar=1.0):
    """f(t_j) = <site| exp(-i H t_j / hbar) |site> for every t_j, evaluated on the GPU.

    The eigendecomposition of the small Hermitian block stays on the host (L <= 10 in
    the apps); the per-time work -- the Python loop of evolve_state_in_H
    (final2.py:85-91, streamlit_app_combined.py:86-93), the apps' own hot loop -- runs
    as one kernel: f_j = V[site, :] . (exp(-i w t_j / hbar) * c0),  c0 = V^H |site>."""
    import ctypes
    w, c0, vrow = _eigen_data(H, site)
    t = np.ascontiguousarray(times, dtype=np.float64)
    out = np.empty(t.shape[0], dtype=np.complex128)
    lib = _lib.load()
    _lib.check(lib.mfft_return_amplitude_host(
        out.ctypes.data, t.shape[0], t.ctypes.data, 0.0, 0.0, w.ctypes.data, c0.ctypes.data,
        vrow.ctypes.data, int(w.shape[0]), float(hbar)))
    return out


def return_amplitude(Htop, site, times, hbar=1.0):
    """Drop-in for streamlit_app_combined.py:166-169 (return amplitude of the 3 x 3 top block)."""
    return _amplitude_on_device(Htop, site, times, hbar)


def Heff_block_general(K, J, m):
    """L x L effective block for one top excitation with the bottom one fixed at m
    (final2.py:47-64): diagonal S at i == m, S - 2(K[i] + K[m]) elsewhere, S = sum(K);
    J on the first off-diagonals."""
    K = np.asarray(K, float)
    J = np.asarray(J, float)
    L = len(K)
    S = float(np.sum(K))
    d = S - 2.0 * (K + K[m])
    d[m] = S
    H = np.diag(d)
    i = np.arange(L - 1)
    H[i, i + 1] = J
    H[i + 1, i] = J
    return H


def return_amplitude_block_general(K, J, bottom_m, site, times, hbar=1.0):
    """Drop-in for final2.py:93-97: f(t) of block `bottom_m` at top site `site`."""
    return _amplitude_on_device(Heff_block_general(K, J, bottom_m), site, times, hbar)


def fft_extract_E_p(Htop, site, times, n_peaks=3, pad=16, use_hann=True, demod=True):
    """(E_est ascending, p_est, omega, S, mu) from f_ss(t) by FFT, optional centroid
    demodulation and a 2^k sign-search least squares; see
    streamlit_app_combined.py:190-251.

    f_ss(t) is generated on the device and stays there: one call runs the rough pass (only its two sums are
    looked at), the demodulated main pass and the peak search on its spectrum; a second one reduces the Gram
    matrix of the sign search on the same samples."""
    w, c0, vrow = _eigen_data(Htop, site)
    times = np.asarray(times, float)
    dt = float(times[1] - times[0])
    n0 = times.shape[0]
    nfft = int(pad * n0)
    omega = _omega_axis(nfft, dt).copy()
    tol = TWO_PI / (times[-1] - times[0] + 1e-12)
    flags = _lib.POWER | _lib.STATS | (_lib.HANN if use_hann else 0)
    rough = cached_plan(nfft, n0, 1, _lib.C128, flags) if demod else None
    plan = cached_plan(nfft, n0, 1, _lib.C128, flags | (_lib.DEMOD if demod else 0))
    found = {}

    def columns_of(res):
        # local maxima ranked by height and de-duplicated on |omega| (the first n_peaks the reference's loop keeps
        # are the first n_peaks it picks: it walks the candidates by decreasing height); every fit of the 2^k sign
        # search uses k of the 2k columns exp(-i(+-|omega|)t): one Gram matrix of all 2k, each fit read off a sub-block
        chosen = [int(i) for i in res["idx"][0, :int(res["count"][0])]][:max(int(n_peaks), 0)]
        if len(chosen) == 0:
            return None
        mags = np.sort(np.unique(np.round(np.abs(omega[chosen]), 12)))[:n_peaks]
        if 2 * len(mags) > 64:
            raise ValueError("at most 32 peaks can enter the sign search")
        found["mags"] = mags
        return np.concatenate((mags, -mags))
    res = plan.exec_host_extract(rough, w, c0, vrow, times, axis_dt=dt, guard=0.6 * tol, want=max(int(n_peaks), 1),
                                 scale=dt, t0=float(times[0]), dt=dt, columns_of=columns_of)
    S = res["S"][0]
    mu = res["mu"]
    if res["gram"] is None:
        return np.array([]), np.array([]), omega, S, mu
    mags = found["mags"]
    k = len(mags)
    g = res["gram"]
    G = g[:8 * k * k].view(np.complex128).reshape(2 * k, 2 * k)
    r = g[8 * k * k:8 * k * k + 4 * k].view(np.complex128)
    signs, best_p = _sign_search(G, r, float(g[-1]), k)
    best_E = signs * mags
    E_phys = best_E - mu
    order = np.argsort(E_phys)
    return E_phys[order], best_p[order], omega, S, mu
