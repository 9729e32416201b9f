"""mfft_exec recorded into a graph (the replacement for np.fft.fft, final2.py:143, inside a caller's captured
step): single-kernel, multi-pass, packed-real and chirp-z plans replay with fresh inputs; a plan that would
have to allocate during the capture says so."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fft_amd
    return fft_amd


def _graph_of(torch, fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            fn(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    return g


@pytest.mark.parametrize("case", ["lds_hann_power", "three_pass_c128", "two_pass_c64_window", "packed_real", "chirpz_stats"])
def test_plans_replay_from_a_graph(fa, case):
    import torch
    torch.cuda.set_device(0)
    rng = np.random.default_rng(len(case))

    def cplx(shape, cdt):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(cdt)
    d_stats = None
    if case == "lds_hann_power":
        n, n_in, batch, dtype, flags = 1024, 1024, 64, 0, fa.HANN | fa.POWER
    elif case == "three_pass_c128":
        n, n_in, batch, dtype, flags = 1 << 20, 1 << 20, 2, 1, 0
    elif case == "two_pass_c64_window":
        n, n_in, batch, dtype, flags = 1 << 17, 100001, 3, 0, fa.HANN
    elif case == "packed_real":
        n, n_in, batch, dtype, flags = 1 << 18, 1 << 18, 2, 1, fa.REAL_IN
    else:
        n, n_in, batch, dtype, flags = 24016, 1501, 2, 1, fa.HANN | fa.POWER | fa.STATS
        d_stats = torch.zeros(3 * batch, dtype=torch.float64, device="cuda")
    cdt = np.complex64 if dtype == 0 else np.complex128
    rdt = np.float32 if dtype == 0 else np.float64
    real_in, power = bool(flags & fa.REAL_IN), bool(flags & fa.POWER)

    def sample():
        return rng.standard_normal((batch, n_in)).astype(rdt) if real_in else cplx((batch, n_in), cdt)

    def reference(x):
        xw = x.astype(np.float64 if real_in else np.complex128)
        if flags & fa.HANN:
            xw = xw * np.hanning(n_in)
        F = 0.5 * np.fft.fft(xw, n, axis=-1)
        return np.abs(F) ** 2 if power else F
    x0 = sample()
    d_x = torch.from_numpy(x0).cuda()
    d_y = torch.empty((batch, n), dtype=(torch.float32 if dtype == 0 else torch.float64) if power
                      else (torch.complex64 if dtype == 0 else torch.complex128), device="cuda")
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags) as p:
        p.exec(d_x, d_y, scale=0.5, d_stats=d_stats, stream=torch.cuda.current_stream().cuda_stream)     # allocates what it needs
        torch.cuda.synchronize()
        g = _graph_of(torch, lambda s: p.exec(d_x, d_y, scale=0.5, d_stats=d_stats, stream=s))
        for _ in range(3):
            x = sample()
            d_x.copy_(torch.from_numpy(x))
            d_y.zero_()
            g.replay()
            torch.cuda.synchronize()
            ref = reference(x)
            assert rel_l2(d_y.cpu().numpy(), ref) < (1e-5 if dtype == 0 else 1e-12)
            if d_stats is not None:
                st = d_stats.cpu().numpy().reshape(batch, 3)
                np.testing.assert_allclose(st[:, 0], ref.max(axis=1), rtol=1e-11)
                np.testing.assert_allclose(st[:, 1], ref.sum(axis=1), rtol=1e-11)
        del g


def test_a_capture_that_would_allocate_is_refused(fa):
    """A three-pass power-spectrum plan takes its second workspace on first use: recorded into a graph before any
    ordinary execution it returns an error (and records nothing); after one ordinary execution it records."""
    import torch
    torch.cuda.set_device(0)
    n = 1 << 22
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((1, n)) + 1j * rng.standard_normal((1, n))).astype(np.complex64)
    d_x = torch.from_numpy(x).cuda()
    d_y = torch.empty((1, n), dtype=torch.float32, device="cuda")
    with fa.Plan(n, dtype=0, flags=fa.POWER) as p:
        assert p.launches == 3
        with pytest.raises(Exception, match="outside stream capture"):
            _graph_of(torch, lambda s: p.exec(d_x, d_y, stream=s))
        torch.cuda.synchronize()
        p.exec(d_x, d_y, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        g = _graph_of(torch, lambda s: p.exec(d_x, d_y, stream=s))
        d_y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert rel_l2(d_y.cpu().numpy(), np.abs(np.fft.fft(x.astype(np.complex128), axis=-1)) ** 2) < 1e-5
