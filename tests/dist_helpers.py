"""Helpers for the multi-process tests: a CPU stand-in for the local operations
(test infrastructure: numpy on CPU tensors, same interface as fft_amd.dist.HipOps)
and the per-rank worker bodies that torch.multiprocessing spawns."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def layout_index(n, batch, layout, dense_stride):
    """Flat element positions (batch, n) described by a (batch_stride, seg, seg_stride) layout."""
    bs, seg, ss = layout if layout is not None else (0, 0, 0)
    bs = bs or dense_stride
    j = np.arange(n, dtype=np.int64)
    within = (j // seg) * ss + (j % seg) if seg else j
    return np.arange(batch, dtype=np.int64)[:, None] * bs + within[None, :]


class NumpyOps:
    """CPU stand-in: numpy's FFT as the local engine (the oracle's arithmetic).
    `columns=False` makes it decline column-batched transforms, which drives the
    decompositions down their transpose fallback."""

    def __init__(self, columns=True):
        self.columns = columns

    @staticmethod
    def min_group_cols(like):
        return 1                                   # the stand-in has no minimum segment length

    def columns_ok(self, n, inner, like, batch=1, out_ld=None):
        return self.columns

    def fft_columns(self, src, dst, n, inner, batch=1, twiddle=None, out_ld=None):
        assert self.columns
        x = src.numpy()[: batch * n * inner].reshape(batch, n, inner)
        y = np.fft.fft(x, axis=1)
        if twiddle is not None:
            n_total, c0 = twiddle
            k = np.arange(n, dtype=np.int64)[:, None]
            c = (c0 + np.arange(inner, dtype=np.int64))[None, :]
            y = y * np.exp(-2j * np.pi * ((k * c) % n_total) / n_total)[None]
        if out_ld is None or out_ld == inner:
            dst.numpy()[: batch * n * inner] = y.reshape(-1)
        else:                                      # rows of the result out_ld elements apart
            assert batch == 1
            d = dst.numpy()
            rows = np.arange(n, dtype=np.int64)[:, None] * out_ld + np.arange(inner, dtype=np.int64)[None, :]
            d[rows] = y[0]

    def swap_axes(self, src, dst, a, b, w, strides=None):
        sa, sb = strides if strides is not None else (b * w, w)
        idx = (np.arange(b, dtype=np.int64)[:, None, None] * sb + np.arange(a, dtype=np.int64)[None, :, None] * sa
               + np.arange(w, dtype=np.int64)[None, None, :])
        dst.numpy()[: a * b * w] = src.numpy()[idx].reshape(-1)

    def empty(self, count, like):
        import torch
        return torch.empty(int(count), dtype=like.dtype)

    def fft(self, src, dst, n, batch, in_layout=None, out_layout=None):
        s, d = src.numpy(), dst.numpy()
        x = s[layout_index(n, batch, in_layout, n)]
        d[layout_index(n, batch, out_layout, n)] = np.fft.fft(x, axis=-1)

    def transpose(self, src, dst, rows, cols, batch=1):
        s = src.numpy()[: batch * rows * cols].reshape(batch, rows, cols)
        dst.numpy()[: batch * rows * cols] = s.transpose(0, 2, 1).reshape(-1)

    def twiddle(self, x, rows, cols, r0, c0, n_total):
        v = x.numpy()[: rows * cols].reshape(rows, cols)
        r = (r0 + np.arange(rows, dtype=np.int64))[:, None]
        c = (c0 + np.arange(cols, dtype=np.int64))[None, :]
        v *= np.exp(-2j * np.pi * ((r * c) % n_total) / n_total)


def global_inputs(seed, shape, dtype=np.complex128):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def rel_l2(a, b):
    return float(np.linalg.norm((np.asarray(a) - np.asarray(b)).ravel()) / np.linalg.norm(np.asarray(b).ravel()))


def worker(rank, world, port, use_gpu, results_dir):
    """One rank: runs every distributed algorithm and writes its max error to a file."""
    import torch
    import torch.distributed as dist
    from fft_amd import dist as fd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if use_gpu:
            torch.cuda.set_device(0)                 # rehearsal: every rank shares the one GPU
            ops, dev = fd.HipOps(), "cuda"
            tol = 1e-12
        else:
            ops, dev = NumpyOps(), "cpu"
            tol = 1e-12
        comm = fd.Comm()
        assert comm.world == world and comm.rank == rank
        errs = {}

        # on the CPU stand-in run both the column-batched route and the transpose fallback; on
        # the GPU the shapes pick the route: (64 x 128), 32*64 -> tile passes, (8 x 128), 8*64 -> fallback
        variants = [("col", NumpyOps(columns=True)), ("tr", NumpyOps(columns=False))] if not use_gpu \
            else [("gpu", ops)]
        for tag, vops in variants:
            for n0, n1 in ((64, 96 if not use_gpu else 128), (8, 128)):
                # ---- 2-D slab ---------------------------------------------------------
                A = global_inputs(1, (n0, n1))
                F = np.fft.fft2(A)
                R, W = n0 // world, n1 // world
                mine = torch.from_numpy(A[rank * R:(rank + 1) * R].reshape(-1).copy()).to(dev)
                # row chunks x column groups of the pipelined schedule (col_groups 1: chunked by rows only)
                for chunks, groups in ((1, 1), (2, 1), (4, 1), (None, 1), (1, 2), (2, 2), (2, 3), (None, None), (1, None)):
                    out = fd.fft2_slab(mine, n0, n1, comm, vops, restore_layout=False, chunks=chunks, col_groups=groups)
                    out = out.cpu().numpy().reshape(n0, W)
                    errs[f"fft2_cols_{tag}_{n0}_{chunks}_{groups}"] = rel_l2(out, F[:, rank * W:(rank + 1) * W])
                for chunks, groups in ((1, 1), (2, 1), (3, 1), (None, 1), (1, 2), (2, 2), (2, 3), (None, None), (1, None)):
                    out = fd.fft2_slab(mine, n0, n1, comm, vops, restore_layout=True, chunks=chunks, col_groups=groups)
                    errs[f"fft2_rows_{tag}_{n0}_{chunks}_{groups}"] = rel_l2(out.cpu().numpy().reshape(R, n1), F[rank * R:(rank + 1) * R])
                assert np.array_equal(mine.cpu().numpy(), A[rank * R:(rank + 1) * R].reshape(-1)), "input clobbered"

            # ---- 1-D four-step ---------------------------------------------------------
            for n1f, n2f in ((32, 64), (8, 64)):
                n = n1f * n2f
                x = global_inputs(2, (n,))
                X = np.fft.fft(x)
                chunk = n // world
                mine = torch.from_numpy(x[rank * chunk:(rank + 1) * chunk].copy()).to(dev)
                combos = ((1, 1), (2, 1), (3, 1), (None, 1), (1, 2), (2, 2), (2, 4), (None, None), (1, None))
                for chunks, groups in combos:
                    out = fd.fft1d_fourstep(mine, n1f, n2f, comm, vops, natural_order=True, chunks=chunks, col_groups=groups).cpu().numpy()
                    errs[f"fft1d_N_{tag}_{n1f}_{chunks}_{groups}"] = rel_l2(out, X[rank * chunk:(rank + 1) * chunk])
                R1 = n1f // world
                want = X.reshape(n2f, n1f).T[rank * R1:(rank + 1) * R1]      # [k1][k2] = X[k1 + n1*k2]
                for chunks, groups in combos:
                    out = fd.fft1d_fourstep(mine, n1f, n2f, comm, vops, natural_order=False, chunks=chunks, col_groups=groups).cpu().numpy()
                    errs[f"fft1d_T_{tag}_{n1f}_{chunks}_{groups}"] = rel_l2(out.reshape(R1, n2f), want)
                assert np.array_equal(mine.cpu().numpy(), x[rank * chunk:(rank + 1) * chunk]), "input clobbered"

        # ---- batch sharding: no collective ------------------------------------------
        lo, hi = fd.shard_range(37, world, rank)
        sizes = [fd.shard_range(37, world, r) for r in range(world)]
        assert sizes[0][0] == 0 and sizes[-1][1] == 37 and all(a[1] == b[0] for a, b in zip(sizes, sizes[1:]))
        errs["shard"] = 0.0 if (hi - lo) in (37 // world, 37 // world + 1) else 1.0

        bad = {k: v for k, v in errs.items() if not v < tol}
        with open(os.path.join(results_dir, f"rank{rank}.txt"), "w") as fh:
            fh.write("OK" if not bad else f"FAIL {bad}")
    finally:
        dist.destroy_process_group()
