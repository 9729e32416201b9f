"""GPU parity of the Python drop-ins (fft, fft_from_timeseries, fft_extract_E_p)
against the golden vectors produced by the reference's own functions and against
the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fft_amd
    assert fft_amd.device_count() >= 1
    return fft_amd


def _rand(shape, seed, dtype):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def test_fft_dropin_matches_numpy_semantics(fa):
    x = _rand((3, 5, 64), 1, np.complex128)
    for axis in (-1, 0, 1):
        assert rel_l2(fa.fft(x, axis=axis), np.fft.fft(x, axis=axis)) < 1e-13
    assert rel_l2(fa.fft(x, n=100), np.fft.fft(x, n=100)) < 1e-12        # zero-pad, chirp-z length
    assert rel_l2(fa.fft(x, n=16), np.fft.fft(x, n=16)) < 1e-13          # crop
    for norm in ("backward", "ortho", "forward"):
        assert rel_l2(fa.fft(x, norm=norm), np.fft.fft(x, norm=norm)) < 1e-13
    y32 = fa.fft(x.astype(np.complex64))
    assert y32.dtype == np.complex64 and rel_l2(y32, np.fft.fft(x)) < 1e-5
    xr = np.random.default_rng(2).standard_normal((4, 1000))
    yr = fa.fft(xr)
    assert yr.dtype == np.complex128 and rel_l2(yr, np.fft.fft(xr)) < 1e-12
    assert fa.fft(xr.astype(np.float32)).dtype == np.complex64
    assert fa.fft(np.zeros((0, 8))).shape == (0, 8)


def test_raw_fft_golden(fa, golden):
    z = golden("raw_fft.npz")
    for n in z["sizes"]:
        n = int(n)
        for tag, dtype, tol in (("c128", np.complex128, 1e-12), ("c64", np.complex64, 1e-5)):
            x = _rand(n, n, dtype)
            y = fa.fft(x)
            assert y.dtype == dtype
            step = max(1, n // 64)
            assert rel_l2(y[::step], z[f"{tag}_{n}_sub"]) < tol, (tag, n)
    for n_in, n in ((8001, 256032), (1501, 24016), (100, 128), (1024, 4096)):
        rng = np.random.default_rng(n_in)
        x = rng.standard_normal(n_in) + 1j * rng.standard_normal(n_in)
        y = fa.fft(x, n=n)
        assert rel_l2(y[::max(1, n // 64)], z[f"pad_{n_in}_{n}_sub"]) < 1e-12


def test_fft_from_timeseries_final2_defaults(fa, golden):
    z = golden("final2_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    kw = dict(n_levels=3, pad_factor=32, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    a_lists, b_lists = [], []
    for m in (0, 1, 2):
        E, p, Eaxis, S_norm = fa.fft_from_timeseries(z[f"f_t_{m}"], times, **kw)
        assert S_norm.dtype == np.float64 and len(S_norm) == 256032
        np.testing.assert_allclose(E, z[f"E_{m}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, z[f"p_{m}"], rtol=0, atol=1e-9)
        assert int(np.argmax(S_norm)) == int(z[f"argmax_{m}"])
        assert rel_l2(S_norm[::61], z[f"Snorm_sub_{m}"]) < 1e-12
        lo = int(z[f"Snorm_peakwin_lo_{m}"]); win = z[f"Snorm_peakwin_{m}"]
        assert rel_l2(S_norm[lo:lo + len(win)], win) < 1e-12
        assert np.array_equal(Eaxis[::61], z[f"Eaxis_sub_{m}"])
        assert S_norm.max() == 1.0
        a, b = so.lanczos_ab(E, p)                      # consumer of the path's output
        np.testing.assert_allclose(a, z[f"lanczos_a_{m}"], atol=1e-8)
        np.testing.assert_allclose(b, z[f"lanczos_b_{m}"], atol=1e-8)
        # whole-function parity against the oracle on the same input
        Eo, po, Eax_o, Sn_o = so.fft_from_timeseries(z[f"f_t_{m}"], times, **kw)
        np.testing.assert_allclose(E, Eo, atol=1e-9)
        assert rel_l2(S_norm, Sn_o) < 1e-12 and np.array_equal(Eaxis, Eax_o)
    E, p, _, S_norm = fa.fft_from_timeseries(z["f_t_1"], times)
    np.testing.assert_allclose(E, z["own_defaults_E_1"], atol=1e-9)
    np.testing.assert_allclose(p, z["own_defaults_p_1"], atol=1e-9)
    assert rel_l2(S_norm[::61], z["own_defaults_Snorm_sub_1"]) < 1e-12
    E, p, _, S_norm = fa.fft_from_timeseries(z["f_t_1"], times, n_levels=3, pad_factor=4,
                                             use_hann=False, post_prune=False)
    np.testing.assert_allclose(E, z["rect_E_1"], atol=1e-9)
    np.testing.assert_allclose(p, z["rect_p_1"], atol=1e-9)
    assert rel_l2(S_norm[::61], z["rect_Snorm_sub_1"]) < 1e-12


def test_cfg1_real4096(fa, golden):
    """BASELINE config 1 (the reference's CPU-runnable case) on the GPU path."""
    z = golden("cfg1_real4096.npz")
    t = np.arange(4096) * 0.01
    Etrue = np.array([-3.0, 0.5, 4.25]); ptrue = np.array([0.2, 0.5, 0.3])
    x = (ptrue[None, :] * np.exp(-1j * Etrue[None, :] * t[:, None])).sum(axis=1).real.copy()
    E, p, Eaxis, S_norm = fa.fft_from_timeseries(x, t, n_levels=6, pad_factor=1, include_dc=False, keep_top=6)
    np.testing.assert_allclose(E, z["E"], atol=1e-9)
    np.testing.assert_allclose(p, z["p"], atol=1e-9)
    assert rel_l2(S_norm, z["S_norm"]) < 1e-12
    assert np.array_equal(Eaxis, z["Eaxis"])


def _assert_lines(E, p, E_gold, p_gold, symmetric):
    """(E, p) must equal the golden lines.  When the true spectrum is exactly symmetric
    (f(t) real, centroid mu = rounding noise ~1e-16) the reference's 2^k sign search sees
    two mirror-image fits with residuals equal to the last bit, and which one it keeps is
    decided by that noise; there the mirrored answer (-E reversed, p reversed) is the same
    result."""
    ok = np.allclose(E, E_gold, rtol=0, atol=1e-8) and np.allclose(p, p_gold, rtol=0, atol=1e-8)
    if not ok and symmetric:
        ok = (np.allclose(E, -E_gold[::-1], rtol=0, atol=1e-8)
              and np.allclose(p, p_gold[::-1], rtol=0, atol=1e-8))
    assert ok, (E, p, E_gold, p_gold)


def test_fft_extract_E_p_combined_defaults(fa, golden):
    z = golden("combined_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    Htop = z["Htop"]
    E, p, omega, S, mu = fa.fft_extract_E_p(Htop, site=2, times=times, n_peaks=3, pad=16,
                                            use_hann=True, demod=True)
    _assert_lines(E, p, z["E_est"], z["p_est"], symmetric=abs(float(z["mu"])) < 1e-12)
    assert len(S) == 24016 and rel_l2(S, z["S"]) < 1e-11
    assert abs(S.sum() - float(z["S_sum"])) < 1e-9 * float(z["S_sum"])
    assert np.array_equal(omega[::7], z["omega_sub"])
    assert abs(mu - float(z["mu"])) < 1e-9
    for tag, kw in (("nodemod", dict(demod=False)), ("rect", dict(use_hann=False)),
                    ("site0_pad4", dict(site=0, pad=4))):
        args = dict(site=2, pad=16); args.update(kw)
        E, p, omega, S, mu = fa.fft_extract_E_p(Htop, times=times, n_peaks=3, **args)
        _assert_lines(E, p, z[f"{tag}_E"], z[f"{tag}_p"], symmetric=abs(float(z[f"{tag}_mu"])) < 1e-12)
        assert abs(mu - float(z[f"{tag}_mu"])) < 1e-9
        assert rel_l2(S[::7], z[f"{tag}_S_sub"]) < 1e-11
    E, p, omega, S, mu = fa.fft_extract_E_p(z["case1_Htop"], site=2, times=times, n_peaks=3, pad=16)
    np.testing.assert_allclose(E, z["case1_E"], atol=1e-8)
    np.testing.assert_allclose(p, z["case1_p"], atol=1e-8)
    assert abs(mu - float(z["case1_mu"])) < 1e-9


def test_no_peak_and_thread_safety(fa):
    t = np.arange(64) * 0.1
    E, p, Eaxis, S_norm = fa.fft_from_timeseries(np.zeros(64, complex), t)
    assert E.size == 0 and p.size == 0 and len(Eaxis) == 64 * 16 and np.all(S_norm == 0.0)
    # concurrent sessions (Streamlit runs one script thread per session)
    import threading
    xs = [_rand(3000, s, np.complex128) for s in range(6)]
    refs = [np.fft.fft(x) for x in xs]
    errs = []

    def work(i):
        for _ in range(5):
            y = fa.fft(xs[i])
            errs.append(rel_l2(y, refs[i]))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert len(errs) == 30 and max(errs) < 1e-12


def test_return_amplitude_generator(fa, golden):
    """Device-side f(t) generator against the reference's per-time loop (golden) and the oracle."""
    z = golden("final2_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    for m in (0, 1, 2):
        f = fa.return_amplitude_block_general(z["K"], z["J"], m, 0, times)
        assert f.dtype == np.complex128 and f.shape == times.shape
        assert rel_l2(f, z[f"f_t_{m}"]) < 1e-13
        assert rel_l2(f, so.return_amplitude_block(z["K"], z["J"], m, 0, times)) < 1e-13
    zc = golden("combined_defaults.npz")
    tc = np.arange(0.0, 15.0 + 0.01, 0.01)
    assert rel_l2(fa.return_amplitude(zc["Htop"], 2, tc), zc["f_t"]) < 1e-13
    # non-uniform grid and hbar != 1
    tn = np.sort(np.random.default_rng(0).random(257)) * 30.0
    H = so.heff_block([1.0, 0.5, 2.0], [0.3, 1.1], 1)
    w, V = np.linalg.eigh(H)
    ref = (np.exp(-1j * np.outer(tn, w) / 0.7) * V[2, :].conj()[None, :]) @ V[2, :]
    assert rel_l2(fa.return_amplitude(H, 2, tn, hbar=0.7), ref) < 1e-13
    # end to end: generator -> estimator reproduces the golden lines of block m = 1
    kw = dict(n_levels=3, pad_factor=32, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    E, p, _, _ = fa.fft_from_timeseries(fa.return_amplitude_block_general(z["K"], z["J"], 1, 0, times), times, **kw)
    np.testing.assert_allclose(E, z["E_1"], atol=1e-9)
    np.testing.assert_allclose(p, z["p_1"], atol=1e-9)


def test_resident_block_pipeline_equals_the_loop_over_blocks(fa, golden):
    """fft_from_timeseries_blocks -- f(t) of every block generated on the device straight into the batched
    transform's input -- against the goldens of the reference's own loop (final2.py:752-758 at the sidebar
    defaults, blocks m = 0, 1, 2) and against the oracle, to the tolerances of the single-signal test."""
    z = golden("final2_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    kw = dict(n_levels=3, pad_factor=32, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    out = fa.fft_from_timeseries_blocks(z["K"], z["J"], [0, 1, 2], 0, times, **kw)
    assert len(out) == 3
    for m, (E, p, Eaxis, S_norm) in zip((0, 1, 2), out):
        np.testing.assert_allclose(E, z[f"E_{m}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, z[f"p_{m}"], rtol=0, atol=1e-9)
        assert int(np.argmax(S_norm)) == int(z[f"argmax_{m}"]) and S_norm.max() == 1.0
        assert rel_l2(S_norm[::61], z[f"Snorm_sub_{m}"]) < 1e-12
        assert np.array_equal(Eaxis[::61], z[f"Eaxis_sub_{m}"])
        Eo, po, _, Sn_o = so.fft_from_timeseries(so.return_amplitude_block(z["K"], z["J"], m, 0, times), times, **kw)
        np.testing.assert_allclose(E, Eo, atol=1e-9)
        np.testing.assert_allclose(p, po, atol=1e-9)
        assert rel_l2(S_norm, Sn_o) < 1e-12
    # any order / subset of blocks, another site, the function's own defaults, hbar != 1; and the empty list
    K, J = [1.0, 0.7, 1.3, 0.4], [0.5, 1.5, 0.9]
    t2 = np.arange(3000) * 0.02
    for ms, site, kw2 in (([3, 1], 2, {}), ([2], 0, dict(use_hann=False, pad_factor=4, post_prune=False)),
                          ([0, 1, 2, 3], 1, dict(hbar=0.7, n_levels=4, keep_top=4))):
        got = fa.fft_from_timeseries_blocks(K, J, ms, site, t2, **kw2)
        kw_ref = {k: v for k, v in kw2.items() if k != "hbar"}
        for m, (E, p, Eaxis, S_norm) in zip(ms, got):
            f = so.return_amplitude_block(K, J, m, site, t2 / kw2.get("hbar", 1.0))
            Eo, po, Eax_o, Sn_o = so.fft_from_timeseries(f, t2, **kw_ref)
            np.testing.assert_allclose(E, Eo, atol=1e-8)
            np.testing.assert_allclose(p, po, atol=1e-7)
            assert rel_l2(S_norm, Sn_o) < 1e-11 and np.array_equal(Eaxis, Eax_o)
    assert fa.fft_from_timeseries_blocks(K, J, [], 0, t2) == []
    with pytest.raises(ValueError):
        fa.fft_from_timeseries_blocks(K, J, [0], 0, np.array([0.0, 1.0, 3.0]))


def test_local_maxima_scan(fa):
    """Device-side candidate scan (final2.py:151, streamlit_app_combined.py:218) against the
    numpy expression, bit-exact on indices, including plateaus, ties, tiny arrays and overflow."""
    import ctypes
    import torch
    from fft_amd import _lib
    lib = _lib.load()

    def scan(S, capacity):
        d_S = torch.from_numpy(S).cuda()
        d_idx = torch.full((max(capacity, 1),), -1, dtype=torch.int64, device="cuda")
        d_cnt = torch.full((1,), 77, dtype=torch.int64, device="cuda")
        _lib.check(lib.mfft_local_maxima(d_S.data_ptr(), S.shape[0], _lib.C64 if S.dtype == np.float32 else _lib.C128,
                                         d_idx.data_ptr(), capacity, d_cnt.data_ptr(), None))
        torch.cuda.synchronize()
        return int(d_cnt.item()), d_idx.cpu().numpy()

    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float64):
        for n in (0, 1, 2, 3, 4, 255, 256, 257, 100003, 1 << 20):
            # small integers force plateaus and ties: the left-strict / right-weak rule matters
            S = rng.integers(0, 4, n).astype(dtype)
            want = np.flatnonzero((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:])) + 1 if n >= 3 else np.zeros(0, int)
            cnt, idx = scan(S, max(n, 1))
            assert cnt == want.size, (dtype, n)
            assert np.array_equal(np.sort(idx[:cnt]), want), (dtype, n)
    S = rng.random(10000)
    want = np.flatnonzero((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:])) + 1
    cnt, idx = scan(S, 100)                               # overflow: total reported, buffer not overrun
    assert cnt == want.size and cnt > 100
    assert set(idx[:100].tolist()) <= set(want.tolist())
    with pytest.raises(ValueError):
        _lib.check(lib.mfft_local_maxima(None, 8, _lib.C128, None, 8, None, None))


def test_fft_on_device_tensors(fa):
    """fft() on torch CUDA tensors: numpy semantics, result stays on the device."""
    import torch
    x = _rand((3, 5, 64), 1, np.complex128)
    d = torch.from_numpy(x).cuda()
    for axis in (-1, 0, 1):
        y = fa.fft(d, axis=axis)
        assert y.is_cuda and y.dtype == torch.complex128
        assert rel_l2(y.cpu().numpy(), np.fft.fft(x, axis=axis)) < 1e-13
    assert rel_l2(fa.fft(d, n=100).cpu().numpy(), np.fft.fft(x, n=100)) < 1e-12
    assert rel_l2(fa.fft(d, n=16, norm="ortho").cpu().numpy(), np.fft.fft(x, n=16, norm="ortho")) < 1e-13
    xr = torch.randn(4, 1000, device="cuda")
    yr = fa.fft(xr)
    assert yr.dtype == torch.complex64 and rel_l2(yr.cpu().numpy(), np.fft.fft(xr.cpu().numpy().astype(np.float64))) < 1e-5
    big = torch.view_as_complex(torch.randn(4, 1 << 20, 2, device="cuda"))
    assert rel_l2(fa.fft(big).cpu().numpy(), np.fft.fft(big.cpu().numpy().astype(np.complex128))) < 1e-5
    assert fa.fft(torch.zeros(0, 8, device="cuda")).shape == (0, 8)
    assert rel_l2(fa.fft(torch.from_numpy(x)), np.fft.fft(x)) < 1e-13        # CPU tensor: numpy path


def test_axis_transforms_and_fft2(fa):
    """Leading / middle axes run column-batched where the shape allows; fft2 is rows then columns."""
    import torch
    for shape, dtype, tol in (((64, 256), np.complex128, 1e-13), ((3, 128, 64), np.complex64, 1e-5),
                              ((2, 16, 4, 8), np.complex128, 1e-13), ((48, 32), np.complex128, 1e-12)):
        x = _rand(shape, sum(shape), dtype)
        for axis in range(len(shape)):
            for norm in (None, "ortho"):
                want = np.fft.fft(x.astype(np.complex128), axis=axis, norm=norm)
                got = fa.fft(x, axis=axis, norm=norm)
                assert got.dtype == dtype and rel_l2(got, want) < tol, (shape, axis, norm)
                got_d = fa.fft(torch.from_numpy(x).cuda(), axis=axis, norm=norm)
                assert rel_l2(got_d.cpu().numpy(), want) < tol, (shape, axis, norm)
    a = _rand((512, 1024), 9, np.complex64)
    assert rel_l2(fa.fft2(a), np.fft.fft2(a.astype(np.complex128))) < 1e-5
    assert rel_l2(fa.fft2(torch.from_numpy(a).cuda()).cpu().numpy(), np.fft.fft2(a.astype(np.complex128))) < 1e-5


def test_normal_equations_on_device(fa):
    """Gram matrix, right-hand side and target energy of the windowed line fit (final2.py:174-175,
    streamlit_app_combined.py:233-246) against numpy, including more than 16 columns (two launches)
    and the demodulated target."""
    from fft_amd.spectral import _normal_equations
    rng = np.random.default_rng(12)
    for n0, m, mu, hann in ((8001, 3, 0.0, True), (1501, 6, -0.81, True), (700, 20, 0.3, False), (50, 1, 0.0, True)):
        t = np.sort(rng.random(n0)) * 40.0 if m == 20 else np.arange(n0) * 0.01
        E = rng.uniform(-300, 300, m)
        x = rng.standard_normal(n0) + 1j * rng.standard_normal(n0)
        win = np.hanning(n0) if hann else None
        G, r, yy = _normal_equations(x, win, t, E, mu=mu)
        w = win if hann else np.ones(n0)
        B = np.exp(-1j * np.outer(t, E)) * w[:, None]
        y = x * w * np.exp(1j * mu * t)
        scale = np.linalg.norm(B.conj().T @ B)
        assert np.linalg.norm(G - B.conj().T @ B) < 1e-12 * scale
        assert np.linalg.norm(r - B.conj().T @ y) < 1e-12 * np.linalg.norm(B) * np.linalg.norm(y)
        assert abs(yy - np.vdot(y, y).real) < 1e-12 * yy
    with pytest.raises(ValueError):
        _normal_equations(x, None, t, np.zeros(65))


def test_peak_buffer_overflow_reruns_on_device(fa):
    """A noise-like spectrum has far more local maxima than samples: the scan is repeated with
    room for all of them, still on the device."""
    rng = np.random.default_rng(21)
    n0, pad = 300, 512
    x = rng.standard_normal(n0) + 1j * rng.standard_normal(n0)
    S, smax, ssum, cen, idx = fa.power_spectrum(x, n0 * pad, 0.01, use_hann=False, want_peaks=True)
    want = np.flatnonzero((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:])) + 1
    assert np.array_equal(idx, want)
    plan_S, stats, few = fa.npfft.cached_plan(n0 * pad, n0, 1, fa.C128, fa.POWER | fa.STATS).exec_host_peaks(
        x, scale=0.01, capacity=8)
    assert isinstance(few, int) and few == want.size


def test_non_finite_samples_behave_like_the_reference(fa):
    """A NaN (or Inf) sample poisons the whole spectrum in numpy's transform; the estimator then
    finds no peak and returns the NaN spectrum unnormalised (S.max() > 0 is False): same here."""
    t = np.arange(300) * 0.05
    f = np.exp(-1j * 1.3 * t)
    for bad in (np.nan, np.inf):
        g = f.copy()
        g[17] = bad
        y = fa.fft(g)
        assert np.all(np.isnan(y.real) | np.isnan(y.imag))
        with np.errstate(all="ignore"):
            Eo, po, axo, So = so.fft_from_timeseries(g, t, n_levels=3, pad_factor=4)
        E, p, ax, S = fa.fft_from_timeseries(g, t, n_levels=3, pad_factor=4)
        assert E.size == Eo.size == 0 and p.size == po.size == 0
        assert np.array_equal(ax, axo) and np.all(np.isnan(S)) and np.all(np.isnan(So))
    # a batch: only the poisoned signal is affected
    x = _rand((4, 1024), 3, np.complex64)
    x[2, 100] = np.nan
    y = fa.fft(x)
    assert np.all(np.isnan(y[2])) and rel_l2(y[[0, 1, 3]], np.fft.fft(x[[0, 1, 3]].astype(np.complex128))) < 1e-5


def test_plan_cache_eviction_while_in_use(fa, monkeypatch):
    """A plan evicted from the shared cache by another session must stay usable for the thread
    that is executing it (it is released when its last user lets go, not on eviction)."""
    import threading
    from fft_amd import npfft
    monkeypatch.setattr(npfft, "_CACHE_MAX", 2)
    npfft.clear_plan_cache()
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((64, 4096)) + 1j * rng.standard_normal((64, 4096)))
    ref = np.fft.fft(x)
    errors, stop = [], threading.Event()

    def user():
        try:
            plan = npfft.cached_plan(4096, 4096, 64, fa.C128, 0)
            for _ in range(60):
                if rel_l2(plan.exec_host(x), ref) > 1e-12:
                    errors.append("wrong result")
        except Exception as e:                                        # noqa: BLE001
            errors.append(repr(e))
        finally:
            stop.set()

    def churn():
        k = 0
        while not stop.is_set():
            npfft.cached_plan(256 + 16 * (k % 40), 256 + 16 * (k % 40), 1, fa.C128, 0)
            k += 1

    threads = [threading.Thread(target=user), threading.Thread(target=churn)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    npfft.clear_plan_cache()
    assert not errors, errors


def test_plan_cache_is_bounded_by_device_bytes(fa, monkeypatch):
    """The cache evicts by the device memory its plans hold, not only by their number: cycling
    through more large lengths than the count limit never holds more than the byte budget (plus
    the newest plan), every transform stays correct, and an out-of-memory build clears the cache
    and retries."""
    import ctypes
    from fft_amd import npfft, _lib
    npfft.clear_plan_cache()
    monkeypatch.setattr(npfft, "_budget", 192 << 20)               # 192 MiB
    free0, total = ctypes.c_int64(), ctypes.c_int64()
    lib = _lib.load()
    lib.mfft_device_memory(ctypes.byref(free0), ctypes.byref(total))
    assert 0 < free0.value <= total.value
    rng = np.random.default_rng(12)
    lengths = [1_000_003 + 1009 * i for i in range(30)]             # chirp-z plans of ~100 MiB each
    for n in lengths:
        x = rng.standard_normal(2000) + 1j * rng.standard_normal(2000)
        y = fa.fft(x, n=n)
        k = np.array([0, 1, n // 3, n - 1])
        want = (x[None, :] * np.exp(-2j * np.pi * ((k[:, None] * np.arange(2000)[None, :]) % n) / n)).sum(axis=1)
        assert np.max(np.abs(y[k] - want)) < 1e-10
        with npfft._cache_lock:
            held = sum(npfft._cache_bytes.values())
            newest = npfft._cache_bytes[next(reversed(npfft._cache))]
            assert held - newest <= (192 << 20) and len(npfft._cache) <= npfft._CACHE_MAX
    import gc
    gc.collect()
    free1 = ctypes.c_int64()
    lib.mfft_device_memory(ctypes.byref(free1), ctypes.byref(total))
    assert free0.value - free1.value < (1 << 30), "evicted plans must release their device memory"
    # an allocation failure drops the cache and retries once
    calls = {"n": 0}

    def flaky():
        calls["n"] += 1
        if calls["n"] == 1:
            raise _lib.MfftError("out of memory: hipMalloc failed")
        return "built"
    assert npfft._build(flaky) == "built" and calls["n"] == 2 and len(npfft._cache) == 0
    npfft.clear_plan_cache()


def test_batched_estimator_reproduces_goldens_in_one_call(fa, golden):
    """The app's loop over blocks m = 0, 1, 2 (final2.py:752-758) as one batched call: every
    block's (E, p, axis, S_norm) equals the reference's own output (golden) and what the
    single-signal function returns."""
    z = golden("final2_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    kw = dict(n_levels=3, pad_factor=32, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    res = fa.fft_from_timeseries_batch([z[f"f_t_{m}"] for m in (0, 1, 2)], times, **kw)
    assert len(res) == 3
    for m, (E, p, Eaxis, S_norm) in enumerate(res):
        np.testing.assert_allclose(E, z[f"E_{m}"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, z[f"p_{m}"], rtol=0, atol=1e-9)
        assert int(np.argmax(S_norm)) == int(z[f"argmax_{m}"]) and S_norm.max() == 1.0
        assert rel_l2(S_norm[::61], z[f"Snorm_sub_{m}"]) < 1e-12
        assert np.array_equal(Eaxis[::61], z[f"Eaxis_sub_{m}"])
        E1, p1, ax1, S1 = fa.fft_from_timeseries(z[f"f_t_{m}"], times, **kw)
        np.testing.assert_allclose(E, E1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(p, p1, rtol=0, atol=1e-12)
        # (a plan of several complex128 signals may split its chirp-z length differently from a plan
        # of one: same spectrum to rounding, not bit for bit)
        np.testing.assert_allclose(S_norm, S1, rtol=0, atol=1e-13)
        assert np.array_equal(Eaxis, ax1)
    # ragged outcomes inside one batch: a silent signal (no line), a real one, a noisy one
    t = np.arange(500) * 0.02
    sigs = [np.zeros(500, complex), np.cos(1.7 * t), _rand(500, 4, np.complex128) + 5 * np.exp(-1j * 2.2 * t)]
    got = fa.fft_from_timeseries_batch(sigs, t, n_levels=4, pad_factor=8)
    for f, (E, p, ax, Sn) in zip(sigs, got):
        Eo, po, axo, Sno = so.fft_from_timeseries(f, t, n_levels=4, pad_factor=8)
        assert E.shape == Eo.shape
        np.testing.assert_allclose(E, Eo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, po, rtol=0, atol=1e-9)
        assert rel_l2(Sn, Sno) < 1e-12 and np.array_equal(ax, axo)
    assert fa.fft_from_timeseries_batch([], t) == []
    with pytest.raises(ValueError):
        fa.fft_from_timeseries_batch([np.ones(499)], t)


def _reference_pick(S, omega, mode, want, guard, dc_thresh, include_dc):
    """The candidate / ranking / guard-spaced pick / refinement steps as the reference writes them
    (final2.py:151-171 for mode 0, streamlit_app_combined.py:218-227 for mode 1), on a given S."""
    loc = np.where((S[1:-1] > S[:-2]) & (S[1:-1] >= S[2:]))[0] + 1
    if mode == 0:
        if include_dc:
            if S[0] > 0:
                loc = np.unique(np.concatenate(([0], loc)))
        else:
            loc = loc[np.abs(omega[loc]) > dc_thresh]
    cand = loc[np.argsort(S[loc], kind="stable")[::-1]]
    picked = []
    for k in cand:
        if len(picked) >= want:
            break
        if mode == 0:
            ok = all(abs(omega[k] - omega[j]) > guard for j in picked)
        else:
            ok = all(abs(abs(omega[k]) - abs(omega[j])) > guard for j in picked)
        if ok:
            picked.append(int(k))
    return picked, np.array([so._log_parabola(omega, S, i) for i in picked])


@pytest.mark.parametrize("dtype", [0, 1])
def test_device_peak_pipeline_matches_reference_steps(fa, dtype):
    """mfft_lines_host's candidates, ranking, guard-spaced pick and refinement against the
    reference's own steps evaluated on the SAME spectrum (bit-exact bins, refined positions to
    1e-12), over both modes, DC handling, guards from none to wide, noise-like and line spectra,
    batches, and exact height ties."""
    rng = np.random.default_rng(17 + dtype)
    cdt = np.complex64 if dtype == 0 else np.complex128
    # n0, pad, batch.  The last two are noise without padding: ~n/3 local maxima, more than the pick kernel
    # holds in registers (8 per thread: 2048 below 16384 bins, 8192 from there), i.e. its global-memory form.
    cases = [(300, 16, 3), (1501, 16, 2), (257, 1, 4), (64, 4, 1), (1000, 3, 5), (12000, 1, 2), (40000, 1, 1)]
    for n0, pad, batch in cases:
        n, dt = n0 * pad, 0.01
        t = np.arange(n0) * dt
        x = 0.05 * (rng.standard_normal((batch, n0)) + 1j * rng.standard_normal((batch, n0)))
        for b in range(batch):
            for E, w in zip(rng.uniform(-40, 40, 4), rng.uniform(0.1, 1.0, 4)):
                x[b] += w * np.exp(-1j * E * t)
        x[0] = x[0].real                                   # symmetric spectrum: +-omega pairs of equal height
        x = x.astype(cdt)
        omega = 2.0 * np.pi * np.fft.fftfreq(n, d=dt)
        domega = 2 * np.pi / (t[-1] - t[0] + 1e-12)
        with fa.Plan(n, n_in=n0, batch=batch, dtype=dtype, flags=fa.POWER | fa.STATS | fa.HANN) as p:
            for mode, include_dc, gm, want in ((0, False, 1.2, 3), (0, True, 2.0, 6), (0, False, 0.0, 8),
                                               (0, False, 25.0, 64), (1, False, 0.6, 3), (1, False, 0.6, 9)):
                r = p.exec_host_lines(x, axis_dt=dt, guard=gm * domega, want=want, dc_guard=1.5 * domega,
                                      include_dc=include_dc, mode=mode, scale=dt)
                S = r["S"].astype(np.float64)
                for b in range(batch):
                    picked, refined = _reference_pick(S[b], omega, mode, want, gm * domega, 1.5 * domega, include_dc)
                    k = int(r["count"][b])
                    got = [int(i) for i in r["idx"][b, :k]]
                    if got != picked:
                        # exact ties (the +-omega bins of a real signal) may be taken in either order;
                        # the heights walked must still be the same non-increasing sequence
                        assert sorted(got) == sorted(picked) or np.array_equal(S[b][got], S[b][picked]), (n0, mode, b)
                        assert np.array_equal(S[b][got], S[b][picked])
                    else:
                        np.testing.assert_allclose(r["omega_hat"][b, :k], refined, rtol=0,
                                                   atol=1e-12 * max(1.0, np.abs(omega).max()))
                    assert np.all(r["idx"][b, k:] == -1)
                    assert np.array_equal(r["height"][b, :k], S[b][got])
    with fa.Plan(64, batch=1, dtype=1, flags=fa.POWER | fa.STATS) as p:
        with pytest.raises(ValueError):
            p.exec_host_lines(np.ones(64, complex), axis_dt=0.1, guard=1.0, want=65)
        with pytest.raises(ValueError):
            p.exec_host_lines(np.ones(64, complex), axis_dt=0.1, guard=1.0, want=0)
    with fa.Plan(64, batch=1, dtype=1) as p:
        with pytest.raises(ValueError):
            p.exec_host_lines(np.ones(64, complex), axis_dt=0.1, guard=1.0, want=2)


def test_repeated_calls_on_one_cached_plan(fa):
    """The estimators keep per-plan device state between calls (staging blocks, candidate counters
    that the pick kernel leaves at zero, pinned landing buffers).  Repeated calls -- same settings, new
    data; then other settings on the same plan (another number of lines reallocates the plan's
    buffers) and back -- must keep giving what the reference gives for each input."""
    n0, dt = 1501, 0.01
    t = np.arange(n0) * dt
    rng = np.random.default_rng(5)

    def signal():
        E = rng.uniform(-20, 20, 3)
        w = rng.uniform(0.2, 1.0, 3)
        return (w[None, :] / w.sum() * np.exp(-1j * E[None, :] * t[:, None])).sum(axis=1)

    def check(f, **kw):
        E, p, ax, S = fa.fft_from_timeseries(f, t, **kw)
        Eo, po, axo, So = so.fft_from_timeseries(f, t, **kw)
        assert E.shape == Eo.shape
        np.testing.assert_allclose(E, Eo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, po, rtol=0, atol=1e-8)
        np.testing.assert_allclose(S, So, rtol=0, atol=1e-12)
        assert np.array_equal(ax, axo)

    kw3 = dict(n_levels=3, pad_factor=16, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    kw4 = dict(n_levels=4, pad_factor=16, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=4)
    for _ in range(4):
        check(signal(), **kw3)
    for _ in range(3):
        check(signal(), **kw4)             # reallocates the line buffers
    for _ in range(3):
        check(signal(), **kw3)
    blocks = [signal() for _ in range(3)]
    for _ in range(3):
        outs = fa.fft_from_timeseries_batch(blocks, t, **kw3)
        for f, (E, p, _, S) in zip(blocks, outs):
            Eo, po, _, So = so.fft_from_timeseries(f, t, **kw3)
            np.testing.assert_allclose(E, Eo, rtol=0, atol=1e-9)
            np.testing.assert_allclose(S, So, rtol=0, atol=1e-12)


def test_close_lines_fit_matches_the_tall_least_squares(fa):
    """Lines 0.6 to 1.2 resolution cells apart (and guard_mult < 1): the weights come from the
    k x k normal equations reduced on the device, the reference solves the tall N0 x k system.  In
    this regime the Gram matrix is ill-conditioned (its condition number is the square of the
    basis'); the estimator must still return the reference's lines and weights."""
    n0, dt = 2000, 0.01
    t = np.arange(n0) * dt
    cell = 2 * np.pi / (t[-1] - t[0])
    for sep in (1.2, 0.9, 0.6):
        E_true = np.array([-2.0, -2.0 + sep * cell, 3.0, 3.0 + sep * cell])
        p_true = np.array([0.4, 0.3, 0.2, 0.1])
        f = (p_true[None, :] * np.exp(-1j * E_true[None, :] * t[:, None])).sum(axis=1)
        kw = dict(n_levels=4, pad_factor=16, guard_mult=0.5, dc_guard=0.5, min_weight=1e-9)
        E, p, _, _ = fa.fft_from_timeseries(f, t, **kw)
        Eo, po, _, _ = so.fft_from_timeseries(f, t, **kw)
        assert E.shape == Eo.shape, (sep, E, Eo)
        np.testing.assert_allclose(E, Eo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(p, po, rtol=0, atol=1e-7)


def test_estimators_on_random_line_sets(fa):
    """Both estimators against the CPU restatement on seeded random inputs the golden files do not
    hold: 1-6 lines (some closer than the guard band, some at DC), noise, real-valued signals,
    every keyword of final2.py:115-127 / streamlit_app_combined.py:190 at random values, single
    and batched calls (tools/fuzz_estimators.py).  E to 1e-8, weights to 1e-6, spectra to 1e-11.
    Inputs for which the reference's own answer flips under a 1e-13 perturbation (exact height
    ties of mirrored or side-lobe peaks) are set aside: parity is not defined there."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_estimators as fz
    del fz.failures[:], fz.unstable[:]
    count = fz.run(301, 304, 25, verbose=False)
    assert count >= 140
    assert not fz.failures, fz.failures[:3]
    assert len(fz.unstable) < count // 5          # the set-aside cases stay the exception


def test_fft_signature_corners_and_random_shapes(fa):
    """numpy.fft.fft's `out`, a non-integer length, half-precision norm factors, and a seeded sweep
    over dimensions / axes / lengths / norms / dtypes / views against numpy (tools/fuzz_npfft.py)."""
    import os, sys
    rng = np.random.default_rng(8)
    a = rng.standard_normal((5, 12)) + 1j * rng.standard_normal((5, 12))
    out = np.empty((5, 20), dtype=np.complex128)
    got = fa.fft(a, n=20, out=out)
    assert got is out and rel_l2(out, np.fft.fft(a, n=20)) < 1e-12
    with pytest.raises(ValueError):
        fa.fft(a, n=20, out=np.empty((5, 19), dtype=np.complex128))
    with pytest.raises(TypeError):
        fa.fft(a, n=2.5)
    h = rng.standard_normal(17).astype(np.float16)
    for norm in ("ortho", "forward"):
        ref = np.fft.fft(h, norm=norm)
        got = fa.fft(h, norm=norm)
        assert got.dtype == ref.dtype == np.complex64 and rel_l2(got, ref) < 1e-6
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_npfft as fz
    del fz.failures[:]
    assert fz.run(501, 503, 50, verbose=False) >= 150
    assert not fz.failures, fz.failures[:3]


def test_spectra_come_back_in_recycled_page_locked_arrays():
    """The spectrum fft_from_timeseries returns (final2.py:146, 188) lives in a page-locked block the device wrote
    directly; the block is recycled once the last view of it is gone, and never while one is alive."""
    import gc
    from fft_amd import hostmem, spectral as sp
    times = np.arange(8001) * 0.01
    f1 = np.exp(-1j * 1.3 * times) + 0.5 * np.exp(1j * 0.4 * times)
    f2 = np.exp(-1j * 2.1 * times)
    hostmem.release_free_blocks()
    E1, p1, ax1, S1 = sp.fft_from_timeseries(f1, times, n_levels=2, pad_factor=32)
    ref1 = so.fft_from_timeseries(f1, times, n_levels=2, pad_factor=32)
    assert np.allclose(E1, ref1[0], atol=1e-9) and rel_l2(S1, ref1[3]) < 1e-12
    keep = S1[1000:1010]                      # a view keeps the block out of the pool
    addr1 = S1.ctypes.data
    E2, p2, ax2, S2 = sp.fft_from_timeseries(f2, times, n_levels=1, pad_factor=32)
    assert S2.ctypes.data != addr1
    assert rel_l2(S1, ref1[3]) < 1e-12        # untouched by the second call
    assert rel_l2(S2, so.fft_from_timeseries(f2, times, n_levels=1, pad_factor=32)[3]) < 1e-12
    snapshot = keep.copy()
    del S1, E1, p1, ax1
    gc.collect()
    S3 = sp.fft_from_timeseries(f2, times, n_levels=1, pad_factor=32)[3]
    assert S3.ctypes.data != addr1 and np.array_equal(keep, snapshot)
    del keep
    gc.collect()
    addrs = {sp.fft_from_timeseries(f2, times, n_levels=1, pad_factor=32)[3].ctypes.data for _ in range(6)}
    assert addr1 in addrs                     # recycled now
    S4 = sp.fft_from_timeseries(f2, times, n_levels=1, pad_factor=32)[3]
    S4[:10] = 7.0                             # writable like any numpy array
    assert S4.flags.writeable and S4.dtype == np.float64 and S4.shape == (256032,)


def test_resident_extract_passes_against_the_separate_calls():
    """mfft_lines_extract_host / mfft_lines_gram_host (the device passes of fft_extract_E_p,
    streamlit_app_combined.py:197-246, on samples that stay on the device) against the same steps made one by one
    through the host: mu bit for bit, spectrum, picked bins, normal equations; and what the entry points refuse."""
    import fft_amd as fa
    from fft_amd import _lib, spectral as sp
    from fft_amd.npfft import cached_plan
    H = np.array([[0.3, 1.0, 0.0], [1.0, -0.2, 0.7], [0.0, 0.7, 0.5]])
    times = np.arange(0.0, 15.0 + 0.01, 0.01)
    n0, pad, dt = times.shape[0], 16, 0.01
    nfft = pad * n0
    flags = _lib.POWER | _lib.STATS | _lib.HANN
    with fa.Plan(nfft, n_in=n0, batch=1, dtype=fa.C128, flags=flags | _lib.DEMOD) as main, \
            fa.Plan(nfft, n_in=n0, batch=1, dtype=fa.C128, flags=flags) as rough, \
            fa.Plan(nfft, n_in=n0 - 1, batch=1, dtype=fa.C128, flags=flags) as other:
        w, c0, vrow = sp._eigen_data(H, 0)
        tol = sp.TWO_PI / (times[-1] - times[0] + 1e-12)
        gram0 = np.empty(2 * 4 + 2 * 2 + 1)
        E2 = np.array([0.4, -0.4])
        with pytest.raises((ValueError, fa.MfftError), match="no resident samples"):
            _lib.check(_lib.load().mfft_lines_gram_host(main._h, E2.ctypes.data, 2, 0.0, gram0.ctypes.data))
        with pytest.raises((ValueError, fa.MfftError), match="same one-signal plan"):
            main.exec_host_extract(other, w, c0, vrow, times, axis_dt=dt, guard=0.6 * tol, want=3, scale=dt, t0=0.0, dt=dt)
        res = main.exec_host_extract(rough, w, c0, vrow, times, axis_dt=dt, guard=0.6 * tol, want=3, scale=dt,
                                     t0=float(times[0]), dt=dt, columns_of=lambda r: np.array([1.1, 0.3, -1.1, -0.3]))
        # the same steps one by one
        f_t = sp.return_amplitude(H, 0, times)
        assert rel_l2(f_t, so.return_amplitude_top(H, 0, times)) < 1e-13
        _, _, ssum, cen = sp.power_spectrum(f_t, nfft, dt, use_hann=True, want_spectrum=False)
        mu = float((sp.TWO_PI / (nfft * dt)) * cen / ssum)
        assert res["mu"] == mu and mu != 0.0
        S, _, _, _ = sp.power_spectrum(f_t, nfft, dt, use_hann=True, mu=mu, t0=float(times[0]))
        assert rel_l2(res["S"][0], S) < 1e-13
        G, r, yy = sp._normal_equations(f_t, np.hanning(n0), times, np.array([1.1, 0.3, -1.1, -0.3]), mu=mu)
        g = res["gram"]
        assert rel_l2(g[:32].view(np.complex128).reshape(4, 4), G) < 1e-12
        assert rel_l2(g[32:40].view(np.complex128), r) < 1e-12 and abs(g[-1] - yy) < 1e-12 * yy
        # without demodulation: no rough plan, mu = 0
        res0 = rough.exec_host_extract(None, w, c0, vrow, times, axis_dt=dt, guard=0.6 * tol, want=3, scale=dt, t0=0.0, dt=dt)
        assert res0["mu"] == 0.0 and res0["gram"] is None
        S0, _, _, _ = sp.power_spectrum(f_t, nfft, dt, use_hann=True)
        assert rel_l2(res0["S"][0], S0) < 1e-13
