"""The multi-GPU decompositions behind the C ABI (mfft_plan_2d_dist / mfft_plan_1d_dist) on the
one-GPU box:

* over a real RCCL communicator of one rank (ncclCommInitRank, exchanges as RCCL send/recv groups:
  the library's RCCL binding, communicator set-up and stream ordering with real traffic);
* over "local" communicators -- 2, 4 and 8 ranks sharing the GPU inside this process, one host
  thread per rank, exchanges as device copies -- which run the multi-rank index algebra, chunked
  pipelines and event ordering unchanged.

Results are compared with numpy's transform of the assembled global input (the oracle's engine);
tolerances as in BASELINE.json: 1e-5 (complex64), 1e-12 (complex128), relative L2."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = {0: 1e-5, 1: 1e-12}


def _rand(shape, seed, dtype):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def _run_ranks(comms, fn):
    """fn(comm) on one host thread per rank (an exchange waits for every rank); returns the results."""
    out, errs = [None] * len(comms), []

    def work(i):
        try:
            out[i] = fn(comms[i])
        except Exception as e:                                          # noqa: BLE001
            errs.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(comms))]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert not any(t.is_alive() for t in threads), "a rank did not finish"
    assert not errs, errs
    return out


def _check_2d(fd, comms, n0, n1, dtype, chunks, groups=None):
    import torch
    P = len(comms)
    cdt = np.complex64 if dtype == 0 else np.complex128
    a = _rand((n0, n1), n0 + n1 + P, cdt)
    ref = np.fft.fft2(a.astype(np.complex128))
    R, W = n0 // P, n1 // P
    for restore in (False, True):
        def one(c):
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                loc = torch.from_numpy(a[c.rank * R:(c.rank + 1) * R].reshape(-1)).cuda()
                res = [fd.fft2_slab(loc, n0, n1, c, None, restore_layout=restore, chunks=chunks, col_groups=groups) for _ in range(2)]
                torch.cuda.current_stream().synchronize()
                assert torch.equal(res[0], res[1])                      # re-execution of the same plan
                return res[1].cpu().numpy()
        outs = _run_ranks(comms, one)
        for r, o in enumerate(outs):
            want = ref[r * R:(r + 1) * R] if restore else ref[:, r * W:(r + 1) * W]
            assert rel_l2(o.reshape(want.shape), want) < TOL[dtype], (P, n0, n1, restore, r, chunks, groups)


def _check_1d(fd, comms, n1, n2, dtype, chunks, groups=None):
    import torch
    P = len(comms)
    cdt = np.complex64 if dtype == 0 else np.complex128
    n = n1 * n2
    x = _rand(n, n1 + 3 * n2 + P, cdt)
    ref = np.fft.fft(x.astype(np.complex128))
    cnt, R1 = n // P, n1 // P
    for natural in (True, False):
        def one(c):
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                loc = torch.from_numpy(x[c.rank * cnt:(c.rank + 1) * cnt]).cuda()
                keep = loc.clone()
                res = fd.fft1d_fourstep(loc, n1, n2, c, None, natural_order=natural, chunks=chunks, col_groups=groups)
                torch.cuda.current_stream().synchronize()
                assert torch.equal(loc, keep), "the input must not be modified"
                return res.cpu().numpy()
        outs = _run_ranks(comms, one)
        for r, o in enumerate(outs):
            if natural:
                want = ref[r * cnt:(r + 1) * cnt]
            else:                                                       # out[i*n2 + k2] = X[(r*R1 + i) + n1*k2]
                k1 = r * R1 + np.arange(R1)
                want = ref[k1[:, None] + n1 * np.arange(n2)[None, :]].reshape(-1)
            assert rel_l2(o, want) < TOL[dtype], (P, n1, n2, natural, r, chunks, groups)


@pytest.mark.parametrize("dtype", [0, 1])
def test_rccl_communicator_of_one_rank(dtype):
    """RCCL itself: unique id, ncclCommInitRank, a send/recv group through mfft_comm_exchange, and both
    decompositions on that communicator."""
    import torch
    from fft_amd import _lib, dist as fd
    torch.cuda.set_device(0)
    comm = fd.CComm.from_unique_id(1, 0, fd.CComm.unique_id())
    try:
        lib = _lib.load()
        w, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.mfft_comm_info(comm._h, ctypes.byref(w), ctypes.byref(r)))
        assert (w.value, r.value) == (1, 0)
        src = torch.arange(1 << 16, dtype=torch.float64, device="cuda")
        dst = torch.zeros_like(src)
        sp, rp = (ctypes.c_void_p * 1)(src.data_ptr()), (ctypes.c_void_p * 1)(dst.data_ptr())
        _lib.check(lib.mfft_comm_exchange(comm._h, sp, rp, src.numel() * 8,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.equal(src, dst), "RCCL did not carry the block"
        _check_2d(fd, [comm], 256, 128, dtype, None)
        _check_1d(fd, [comm], 256, 64, dtype, None)
    finally:
        comm.close()


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("P", [2, 4, 8])
def test_local_ranks_run_the_multi_rank_paths(P, dtype):
    """Several ranks on the one GPU: chunked first exchange, column transforms, chunked return exchange
    with per-chunk unpack (2-D); pack, chunked exchanges around the row transforms, natural-order
    transpose (1-D).  Chunk counts 1, the default 4 and one that does not divide the rows evenly."""
    import torch
    from fft_amd import dist as fd
    torch.cuda.set_device(0)
    wmin = 32 if dtype == 0 else 16
    comms = fd.CComm.local(P)
    try:
        for chunks in (1, None, 3):
            _check_2d(fd, comms, 64 * P, wmin * P, dtype, chunks)
            _check_1d(fd, comms, 16 * P if 16 * P >= 64 else 64, wmin * P, dtype, chunks)
        _check_2d(fd, comms, 1024, 1024, dtype, None)
        _check_1d(fd, comms, 1024, 512, dtype, None)
        # column groups: wide enough for 1, 2 (the default), 3 (-> 2) and 4 groups of at least 64 / 128 columns
        wide = (128 if dtype == 0 else 64) * 4 * P
        for chunks, groups in ((1, 1), (None, 2), (2, 3), (None, None), (1, 4)):
            _check_2d(fd, comms, 64 * P, wide, dtype, chunks, groups)
            _check_1d(fd, comms, 256, wide, dtype, chunks, groups)
        lib_info = comms[0].plan_info(comms[0].plan("1d", 256, wide, dtype, True, None, None))
        assert lib_info[0] == 4 and lib_info[1] == 2, lib_info
    finally:
        for c in comms:
            c.close()


def test_dist_plan_argument_errors():
    import torch
    from fft_amd import _lib, dist as fd
    torch.cuda.set_device(0)
    lib = _lib.load()
    comms = fd.CComm.local(2)
    try:
        h = ctypes.c_void_p()
        assert lib.mfft_plan_2d_dist(ctypes.byref(h), comms[0]._h, 63, 64, 0, 0, 0, 0) == _lib.E_INVALID
        assert lib.mfft_plan_1d_dist(ctypes.byref(h), comms[0]._h, 64, 63, 0, 0, 0, 0) == _lib.E_INVALID
        assert lib.mfft_plan_2d_dist(ctypes.byref(h), comms[0]._h, 64, 64, 7, 0, 0, 0) == _lib.E_INVALID
        assert lib.mfft_plan_2d_dist(ctypes.byref(h), comms[0]._h, 64, 16, 0, 0, 0, 0) == _lib.E_UNSUPPORTED   # 8 local columns
        assert lib.mfft_plan_2d_dist(ctypes.byref(h), None, 64, 64, 0, 0, 0, 0) == _lib.E_INVALID
        _lib.check(lib.mfft_plan_2d_dist(ctypes.byref(h), comms[0]._h, 64, 64, 0, 1, 0, 0))
        ch, gr, ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        _lib.check(lib.mfft_dist_plan_info(h, ctypes.byref(ch), ctypes.byref(gr), ctypes.byref(ws)))
        assert ch.value == 4 and gr.value == 1 and ws.value == 2 * 32 * 64 * 8      # 32 local columns: one group
        d = torch.zeros(32 * 64, dtype=torch.complex64, device="cuda")
        assert lib.mfft_dist_exec(h, d.data_ptr(), d.data_ptr(), None) == _lib.E_INVALID          # in place
        lib.mfft_dist_plan_destroy(h)
    finally:
        for c in comms:
            c.close()


def test_local_ranks_on_random_shapes():
    """Seeded slice of tools/fuzz_dist_local.py: random rank counts, power-of-two shapes, chunk counts and
    precisions through both decompositions, both output layouts / orders, against numpy."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_dist_local as fz
    del fz.failures[:], fz.refused[:]
    assert fz.run(701, 702, 20, verbose=False) >= 30
    assert not fz.failures, fz.failures[:3]
