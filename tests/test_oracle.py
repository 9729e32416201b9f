"""Pins the CPU oracle (oracle/) before anything is compared against it:
against an O(n^2) long-double DFT, against numpy's pocketfft (the routine the
reference calls), and against the golden vectors produced by the reference's
own functions (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2
from oracle import spectral_oracle as so


def _rand(n, seed, dtype=np.complex128):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dtype)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 12, 16, 30, 64, 97, 100, 127, 128, 254, 509])
def test_c_engine_vs_direct_dft(n):
    x = _rand(n, n)
    ref = so.dft(x, engine="direct")
    assert rel_l2(so.dft(x, engine="c"), ref) < 2e-15
    assert rel_l2(so.dft(x, engine="bluestein"), ref) < 5e-15
    assert rel_l2(so.dft(x, engine="numpy"), ref) < 2e-15


def test_known_answers():
    n = 64
    imp = np.zeros(n, complex); imp[0] = 1
    assert np.allclose(so.dft(imp), np.ones(n), atol=0, rtol=0)
    const = np.ones(n, complex)
    X = so.dft(const)
    assert abs(X[0] - n) < 1e-13 and np.max(np.abs(X[1:])) < 1e-12
    k0 = 5
    tone = np.exp(2j * np.pi * k0 * np.arange(n) / n)
    X = so.dft(tone)
    assert abs(X[k0] - n) < 1e-12 and np.max(np.abs(np.delete(X, k0))) < 1e-12
    # shifted impulse -> forward sign convention exp(-2 pi i k/n)
    imp1 = np.zeros(n, complex); imp1[1] = 1
    assert np.allclose(so.dft(imp1), np.exp(-2j * np.pi * np.arange(n) / n), atol=1e-15)


@pytest.mark.parametrize("n_in,n", [(100, 128), (1024, 4096), (1501, 24016), (8001, 256032), (10, 4)])
def test_zero_pad_and_truncate(n_in, n):
    x = _rand(n_in, n_in)
    a = so.dft(x, n=n, engine="c")
    b = np.fft.fft(x, n=n)
    assert a.shape == (n,)
    assert rel_l2(a, b) < 1e-14


def test_raw_fft_golden(golden):
    z = golden("raw_fft.npz")
    for n in z["sizes"]:
        n = int(n)
        for tag, dtype in (("c128", np.complex128), ("c64", np.complex64)):
            x = _rand(n, n, dtype)
            assert np.array_equal(x[:4], z[f"{tag}_{n}_x_head"]), "seeded input drifted"
            X = so.dft(x, engine="c")            # complex128 oracle
            step = max(1, n // 64)
            g = z[f"{tag}_{n}_sub"]
            tol = 1e-13 if tag == "c128" else 2e-6   # c64 golden was computed in fp32 by numpy
            assert rel_l2(X[::step], g) < tol, (tag, n)
            assert abs(np.linalg.norm(X) - float(z[f"{tag}_{n}_l2"])) / np.linalg.norm(X) < tol
    for n_in, n in ((8001, 256032), (1501, 24016), (100, 128), (1024, 4096)):
        rng = np.random.default_rng(n_in)
        x = rng.standard_normal(n_in) + 1j * rng.standard_normal(n_in)
        X = so.dft(x, n=n, engine="c")
        assert rel_l2(X[::max(1, n // 64)], z[f"pad_{n_in}_{n}_sub"]) < 1e-13


@pytest.mark.parametrize("M", [0, 1, 2, 3, 8, 1501, 8001, 4096])
def test_hanning_matches_numpy(M):
    w = so.hanning(M)
    assert w.dtype == np.float64 and w.shape == (max(M, 0),)
    assert np.array_equal(w, np.hanning(M))
    if M >= 2:
        assert w[0] == 0.0 and w[-1] == 0.0
        n = np.arange(M)
        assert np.allclose(w, 0.5 - 0.5 * np.cos(2 * np.pi * n / (M - 1)), atol=1e-15)


@pytest.mark.parametrize("n,d", [(1, 1.0), (2, 0.5), (7, 0.01), (8, 0.01), (24016, 0.01), (256032, 0.01)])
def test_fftfreq_matches_numpy(n, d):
    assert np.array_equal(so.fftfreq(n, d), np.fft.fftfreq(n, d))


def test_final2_defaults_golden(golden):
    z = golden("final2_defaults.npz")
    K, J = z["K"], z["J"]
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    assert len(times) == int(z["times_len"]) == 8001
    kw = dict(n_levels=3, pad_factor=32, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3)
    a_lists, b_lists, ms = [], [], []
    for m in (0, 1, 2):
        f_t = so.return_amplitude_block(K, J, m, 0, times)
        assert rel_l2(f_t, z[f"f_t_{m}"]) < 1e-13
        for engine in ("numpy", "c"):
            E, p, Eaxis, S_norm = so.fft_from_timeseries(z[f"f_t_{m}"], times, engine=engine, **kw)
            assert len(S_norm) == int(z[f"Snorm_len_{m}"]) == 256032
            np.testing.assert_allclose(E, z[f"E_{m}"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(p, z[f"p_{m}"], rtol=0, atol=1e-9)
            assert int(np.argmax(S_norm)) == int(z[f"argmax_{m}"])
            assert rel_l2(S_norm[::61], z[f"Snorm_sub_{m}"]) < 1e-12
            lo = int(z[f"Snorm_peakwin_lo_{m}"]); win = z[f"Snorm_peakwin_{m}"]
            assert rel_l2(S_norm[lo:lo + len(win)], win) < 1e-12
            assert np.array_equal(Eaxis[::61], z[f"Eaxis_sub_{m}"])
        a, b = so.lanczos_ab(E, p)
        np.testing.assert_allclose(a, z[f"lanczos_a_{m}"], atol=1e-9)
        np.testing.assert_allclose(b, z[f"lanczos_b_{m}"], atol=1e-9)
    # the function's own defaults and the rectangular-window variant
    E, p, _, S_norm = so.fft_from_timeseries(z["f_t_1"], times)
    np.testing.assert_allclose(E, z["own_defaults_E_1"], atol=1e-9)
    np.testing.assert_allclose(p, z["own_defaults_p_1"], atol=1e-9)
    assert rel_l2(S_norm[::61], z["own_defaults_Snorm_sub_1"]) < 1e-12
    E, p, _, S_norm = so.fft_from_timeseries(z["f_t_1"], times, n_levels=3, pad_factor=4,
                                             use_hann=False, post_prune=False)
    np.testing.assert_allclose(E, z["rect_E_1"], atol=1e-9)
    np.testing.assert_allclose(p, z["rect_p_1"], atol=1e-9)
    assert rel_l2(S_norm[::61], z["rect_Snorm_sub_1"]) < 1e-12


def test_combined_defaults_golden(golden):
    z = golden("combined_defaults.npz")
    times = np.arange(0.0, float(z["tmax"]) + float(z["dt"]), float(z["dt"]))
    assert len(times) == int(z["times_len"]) == 1501
    Htop = z["Htop"]
    assert rel_l2(so.return_amplitude_top(Htop, 2, times), z["f_t"]) < 1e-13
    for engine in ("numpy", "c"):
        E, p, omega, S, mu = so.fft_extract_E_p(Htop, site=2, times=times, n_peaks=3, pad=16,
                                                use_hann=True, demod=True, engine=engine)
        np.testing.assert_allclose(E, z["E_est"], atol=1e-8)
        np.testing.assert_allclose(p, z["p_est"], atol=1e-8)
        assert len(S) == 24016 and rel_l2(S, z["S"]) < 1e-11
        assert abs(S.sum() - float(z["S_sum"])) < 1e-9 * float(z["S_sum"])
        assert np.array_equal(omega[::7], z["omega_sub"])
        assert abs(mu) < 1e-12            # ~ -6e-16 at defaults: still "!= 0.0", demod still applied
    for tag, kw in (("nodemod", dict(demod=False)), ("rect", dict(use_hann=False)),
                    ("site0_pad4", dict(site=0, pad=4))):
        args = dict(site=2, pad=16); args.update(kw)
        E, p, omega, S, mu = so.fft_extract_E_p(Htop, times=times, n_peaks=3, **args)
        np.testing.assert_allclose(E, z[f"{tag}_E"], atol=1e-8)
        np.testing.assert_allclose(p, z[f"{tag}_p"], atol=1e-8)
        assert abs(mu - float(z[f"{tag}_mu"])) < 1e-9
        assert rel_l2(S[::7], z[f"{tag}_S_sub"]) < 1e-11
    E, p, omega, S, mu = so.fft_extract_E_p(z["case1_Htop"], site=2, times=times, n_peaks=3, pad=16)
    np.testing.assert_allclose(E, z["case1_E"], atol=1e-8)
    np.testing.assert_allclose(p, z["case1_p"], atol=1e-8)
    assert abs(mu - float(z["case1_mu"])) < 1e-9


def test_cfg1_real4096_golden(golden):
    """BASELINE config 1: N = 4096 real float64 signal through fft_from_timeseries."""
    z = golden("cfg1_real4096.npz")
    t = np.arange(4096) * 0.01
    Etrue = np.array([-3.0, 0.5, 4.25]); ptrue = np.array([0.2, 0.5, 0.3])
    x = (ptrue[None, :] * np.exp(-1j * Etrue[None, :] * t[:, None])).sum(axis=1).real.copy()
    for engine in ("numpy", "c"):
        E, p, Eaxis, S_norm = so.fft_from_timeseries(x, t, n_levels=6, pad_factor=1,
                                                     include_dc=False, keep_top=6, engine=engine)
        np.testing.assert_allclose(E, z["E"], atol=1e-9)
        np.testing.assert_allclose(p, z["p"], atol=1e-9)
        assert rel_l2(S_norm, z["S_norm"]) < 1e-12
        assert np.array_equal(Eaxis, z["Eaxis"])


def test_validation_errors():
    with open(os.path.join(GOLDEN, "errors.json")) as fh:
        msgs = json.load(fh)
    cases = {"ndim": (np.ones(4), np.ones((2, 2))), "short": (np.ones(1), np.ones(1)),
             "len": (np.ones(5), np.arange(4.0)), "uniform": (np.ones(4), np.array([0.0, 1.0, 2.0, 4.0]))}
    for tag, args in cases.items():
        with pytest.raises(ValueError) as ei:
            so.fft_from_timeseries(*args)
        assert str(ei.value) == msgs[tag]
    with pytest.raises(ValueError) as ei:
        so.lanczos_ab([1.0, 2.0], [0.0, 0.0])
    assert str(ei.value) == msgs["lanczos_zero"]


def test_reference_numpy_chain_precision():
    """The chain bench.py times as the CPU baseline keeps the input precision."""
    x = _rand(1024, 0, np.complex64).reshape(1, 1024)
    S = so.reference_numpy_chain(x)
    assert S.dtype == np.float32
    _, S64 = so.windowed_power(x, engine="c")
    assert rel_l2(S, S64) < 1e-5
