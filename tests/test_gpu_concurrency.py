"""Several host threads inside the estimators at once -- Streamlit runs one script thread per browser
session, so fft_from_timeseries (final2.py:754-758) and fft_extract_E_p
(streamlit_app_combined.py:357) are reached concurrently, each session with its own lengths and
therefore its own plan -- plus one thread driving the entry point that allocates nothing on a stream
of its own, and executions of ONE plan from two streams.  Every result is compared with the CPU
oracle / numpy to the single-threaded tolerances."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import rel_l2
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


def _lines_signal(n0, dt, E, p, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n0) * dt
    f = (np.asarray(p)[None, :] * np.exp(-1j * np.asarray(E)[None, :] * t[:, None])).sum(axis=1)
    return t, f + 1e-4 * (rng.standard_normal(n0) + 1j * rng.standard_normal(n0))


def test_estimators_from_six_threads_and_the_static_entry_point_on_its_own_stream():
    import torch
    import fft_amd as fa
    from fft_amd import _lib
    torch.cuda.set_device(0)
    lib = _lib.load()
    loops = 4
    failures, done = [], []
    fail_lock = threading.Lock()

    def guarded(fn):
        def run():
            try:
                fn()
            except BaseException as exc:                      # noqa: BLE001 - reported by the main thread
                with fail_lock:
                    failures.append(repr(exc)[:500])
        return run

    # sessions of the first app: different N0 / padding each, so every thread owns different plans
    shapes = [(1501, 16, 3), (2000, 8, 2), (8001, 32, 3), (1024, 16, 3), (3001, 4, 2), (4096, 1, 3)]

    def batch_session(i):
        n0, pad, nl = shapes[i]
        E = [-2.1, 0.4, 3.3][:nl]
        p = [0.3, 0.5, 0.2][:nl]
        sigs, t = [], None
        for b in range(2):
            t, f = _lines_signal(n0, 0.01 * (1 + 0.1 * i), [e + 0.05 * b for e in E], p, 100 * i + b)
            sigs.append(f)
        kw = dict(n_levels=nl, pad_factor=pad, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=nl)
        want = [so.fft_from_timeseries(f, t, **kw) for f in sigs]
        for _ in range(loops):
            got = fa.fft_from_timeseries_batch(sigs, t, **kw)
            for (E1, p1, ax1, S1), (E0, p0, ax0, S0) in zip(got, want):
                np.testing.assert_allclose(E1, E0, rtol=0, atol=1e-8)
                np.testing.assert_allclose(p1, p0, rtol=0, atol=1e-6)
                assert np.array_equal(ax1, ax0) and rel_l2(S1, S0) < 1e-11
            with fail_lock:
                done.append(("A", i))

    # sessions of the second app (demodulating plans; the 2^k sign search on the device Gram matrix)
    def combined_session(i):
        K = [1.0, 0.6 + 0.1 * i, 1.4]
        H = so.heff_block(K, [0.8, 1.2], 1)
        times = np.arange(0.0, 12.0 + 0.005 * (i + 1), 0.01)
        want = so.fft_extract_E_p(H, 0, times, n_peaks=3, pad=8 + 4 * i)
        for _ in range(loops):
            E1, p1, om1, S1, mu1 = fa.fft_extract_E_p(H, 0, times, n_peaks=3, pad=8 + 4 * i)
            np.testing.assert_allclose(E1, want[0], rtol=0, atol=1e-8)
            np.testing.assert_allclose(p1, want[1], rtol=0, atol=1e-6)
            assert np.array_equal(om1, want[2]) and rel_l2(S1, want[3]) < 1e-11 and abs(mu1 - want[4]) < 1e-9
            with fail_lock:
                done.append(("B", i))

    # a caller of mfft_lsq_normal_static on its own non-blocking stream, checked against numpy's B^H B
    def static_caller():
        torch.cuda.set_device(0)
        rng = np.random.default_rng(9)
        n, batch, m = 6000, 4, 5
        times = np.arange(n) * 0.01
        x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
        win = np.hanning(n)
        E = np.ascontiguousarray(rng.uniform(-4, 4, size=(batch, m)))
        nres = 2 * m * m + 2 * m + 1
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            d_x, d_t, d_w, d_E = (torch.from_numpy(a).cuda() for a in (x, times, win, E))
            d_out = torch.empty((batch, nres), dtype=torch.float64, device="cuda")
        stream.synchronize()
        for _ in range(10 * loops):
            with torch.cuda.stream(stream):
                d_out.fill_(float("nan"))
            _lib.check(lib.mfft_lsq_normal_static(d_x.data_ptr(), d_w.data_ptr(), d_t.data_ptr(), n, batch, None,
                                                  d_E.data_ptr(), m, d_out.data_ptr(), ctypes.c_void_p(stream.cuda_stream)))
            stream.synchronize()
            out = d_out.cpu().numpy()
            for b in range(batch):
                B = np.exp(-1j * np.outer(times, E[b])) * win[:, None]
                y = x[b] * win
                G = out[b, :2 * m * m].view(np.complex128).reshape(m, m)
                r = out[b, 2 * m * m:2 * m * m + 2 * m].view(np.complex128)
                assert np.linalg.norm(G - B.conj().T @ B) <= 1e-12 * np.linalg.norm(B) ** 2
                assert np.linalg.norm(r - B.conj().T @ y) <= 1e-12 * np.linalg.norm(B) * np.linalg.norm(y)
                assert abs(out[b, -1] - np.vdot(y, y).real) <= 1e-12 * np.vdot(y, y).real
        with fail_lock:
            done.append(("S", 0))

    threads = [threading.Thread(target=guarded(lambda i=i: batch_session(i))) for i in range(4)]
    threads += [threading.Thread(target=guarded(lambda i=i: combined_session(i))) for i in range(2)]
    threads += [threading.Thread(target=guarded(static_caller))]
    [th.start() for th in threads]
    [th.join(timeout=100) for th in threads]
    assert not any(th.is_alive() for th in threads), "a session thread did not finish"
    assert not failures, failures
    assert len(done) == 6 * loops + 1


def test_one_plan_from_two_threads_and_streams_with_staged_and_shared_state():
    """Plans whose executions share device state of the plan -- a windowed, zero-padded transform over a
    segmented layout (dense staging copies), a chirp-z length, a multi-pass length -- driven from two
    threads on two streams at once, each with its own buffers; the library orders them."""
    import torch
    import fft_amd as fa
    from fft_amd import _lib
    torch.cuda.set_device(0)
    cases = []
    # (plan, numpy reference builder, input shape)
    n, n_in, batch, seg = 4096, 3000, 8, 256
    seg_plan = fa.Plan(n, n_in=n_in, batch=batch, dtype=_lib.C128, flags=_lib.HANN,
                       out_layout=(2 * n, seg, 2 * seg))

    def seg_ref(x):
        return np.fft.fft(x * np.hanning(n_in)[None, :], n=n, axis=-1)

    def seg_unpack(buf):
        # element j of signal b at b*2n + (j // seg) * 2*seg + j % seg
        v = buf.reshape(batch, n // seg, 2 * seg)[:, :, :seg]
        return v.reshape(batch, n)
    cases.append((seg_plan, seg_ref, (batch, n_in), batch * 2 * n, seg_unpack))
    cz_plan = fa.Plan(3000, n_in=1000, batch=4, dtype=_lib.C128)
    cases.append((cz_plan, lambda x: np.fft.fft(x, n=3000, axis=-1), (4, 1000), 4 * 3000, lambda b: b.reshape(4, 3000)))
    mp_plan = fa.Plan(1 << 16, batch=2, dtype=_lib.C128)
    cases.append((mp_plan, lambda x: np.fft.fft(x, axis=-1), (2, 1 << 16), 2 << 16, lambda b: b.reshape(2, 1 << 16)))

    failures = []

    def worker(plan, ref, shape, out_count, unpack, seed):
        try:
            torch.cuda.set_device(0)
            rng = np.random.default_rng(seed)
            stream = torch.cuda.Stream()
            for it in range(12):
                x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
                with torch.cuda.stream(stream):
                    d_x = torch.from_numpy(x).cuda()
                    d_y = torch.zeros(out_count, dtype=torch.complex128, device="cuda")
                stream.synchronize()
                plan.exec(d_x, d_y, stream=stream.cuda_stream)
                stream.synchronize()
                err = rel_l2(unpack(d_y.cpu().numpy()), ref(x))
                assert err < 1e-12, (shape, it, err)
        except BaseException as exc:                          # noqa: BLE001
            failures.append(repr(exc)[:500])

    threads = []
    for k, (plan, ref, shape, cnt, unpack) in enumerate(cases):
        for w in range(2):
            threads.append(threading.Thread(target=worker, args=(plan, ref, shape, cnt, unpack, 10 * k + w)))
    [th.start() for th in threads]
    [th.join(timeout=100) for th in threads]
    assert not any(th.is_alive() for th in threads)
    assert not failures, failures
    for plan, *_ in cases:
        plan.destroy()
