"""Multi-rank paths.  On CPU (`-m "not gpu"`): world_size 2 over gloo with a numpy
stand-in for the local kernels, which checks the exchange / index algebra of the
2-D slab and 1-D four-step decompositions.  On the GPU box: the same algorithms
with the real kernels, single rank, and a two-rank rehearsal sharing the one GPU."""
import os
import socket
import tempfile

import numpy as np
import pytest

import dist_helpers as dh


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(world, use_gpu):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dh.worker, args=(world, _free_port(), use_gpu, d), nprocs=world, join=True)
        for r in range(world):
            with open(os.path.join(d, f"rank{r}.txt")) as fh:
                assert fh.read() == "OK"


def test_shard_range_covers_everything():
    from fft_amd.dist import shard_range
    for total in (0, 1, 7, 8, 65536, 100003):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_layout_index_matches_definition():
    idx = dh.layout_index(8, 3, (4, 4, 12), 8)      # batch_stride 4, seg 4, seg_stride 12
    assert idx.tolist() == [[0, 1, 2, 3, 12, 13, 14, 15], [4, 5, 6, 7, 16, 17, 18, 19],
                            [8, 9, 10, 11, 20, 21, 22, 23]]


def test_two_ranks_gloo_cpu():
    _spawn(2, use_gpu=False)


def test_four_ranks_gloo_cpu():
    _spawn(4, use_gpu=False)


def test_single_rank_cpu_standin():
    """world_size 1 degenerates to local copies: same code path, no process group."""
    import torch
    from fft_amd import dist as fd
    comm = fd.Comm()
    for ops in (dh.NumpyOps(columns=True), dh.NumpyOps(columns=False)):
        A = dh.global_inputs(5, (16, 24))
        for restore in (True, False):          # one rank: both layouts are the plain (16 x 24) spectrum
            out = fd.fft2_slab(torch.from_numpy(A.reshape(-1).copy()), 16, 24, comm, ops, restore_layout=restore)
            assert dh.rel_l2(out.numpy().reshape(16, 24), np.fft.fft2(A)) < 1e-12
        x = dh.global_inputs(6, (8 * 32,))
        out = fd.fft1d_fourstep(torch.from_numpy(x.copy()), 8, 32, comm, ops)
        assert dh.rel_l2(out.numpy(), np.fft.fft(x)) < 1e-12
        out = fd.fft1d_fourstep(torch.from_numpy(x.copy()), 8, 32, comm, ops, natural_order=False)
        assert dh.rel_l2(out.numpy().reshape(8, 32), np.fft.fft(x).reshape(32, 8).T) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.complex64, 1e-5), (np.complex128, 1e-12)])
def test_single_gpu_real_kernels(dtype, tol):
    import torch
    from fft_amd import dist as fd
    torch.cuda.set_device(0)
    comm, ops = fd.Comm(), fd.HipOps()
    for n0, n1 in ((256, 512), (1024, 64), (2048, 2048), (8, 64), (48, 80)):
        A = dh.global_inputs(n0 + n1, (n0, n1), dtype)
        ref = np.fft.fft2(A.astype(np.complex128))
        d = torch.from_numpy(A.reshape(-1).copy()).cuda()
        out = fd.fft2_slab(d, n0, n1, comm, ops, restore_layout=True)
        assert dh.rel_l2(out.cpu().numpy().reshape(n0, n1), ref) < tol
        out = fd.fft2_slab(d, n0, n1, comm, ops, restore_layout=False)
        assert dh.rel_l2(out.cpu().numpy().reshape(n0, n1), ref) < tol
        assert np.array_equal(d.cpu().numpy(), A.reshape(-1))          # input untouched
    for n1, n2 in ((64, 128), (1024, 1024), (4096, 256), (8, 16), (1 << 17, 64)):
        x = dh.global_inputs(n1 * 3 + n2, (n1 * n2,), dtype)
        ref = np.fft.fft(x.astype(np.complex128))
        out = fd.fft1d_fourstep(torch.from_numpy(x.copy()).cuda(), n1, n2, comm, ops)
        assert dh.rel_l2(out.cpu().numpy(), ref) < tol
    ops.close()


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu():
    """Rehearsal of the N > 1 path with the real kernels: two ranks on cuda:0, the
    exchange staged through gloo (RCCL refuses two ranks on one device)."""
    _spawn(2, use_gpu=True)
