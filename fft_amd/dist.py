"""Multi-GPU decompositions: one process per GPU, torch.distributed (RCCL) for the
exchange steps, libmfft kernels for everything local.

The reference never leaves one process; these are the build-side answers to the
BASELINE configurations that exceed one GPU:

* independent signals (a batch)      -> ``shard_range``: contiguous split, no collective
* 2-D transform of an n0 x n1 array  -> ``fft2_slab``: row FFTs, one all-to-all, column FFTs
                                        (a second all-to-all restores the row-slab layout)
* one long 1-D transform n = n1*n2   -> ``fft1d_fourstep``: transpose, all-to-all, n1-point
                                        FFTs, all-to-all, twiddle, n2-point FFTs
                                        (a third all-to-all gives natural order)

xGMI is a full mesh of point-to-point links, and all-to-all is the one collective
that drives every link at once with a single hop per message, so each exchange is
ONE ``all_to_all_single`` over the whole slab, with the pack / unpack folded into
the transform kernels' own load and store (segmented layouts of
``mfft_plan_1d_layout``) instead of separate copy passes.

Local work goes through an ``ops`` object.  ``HipOps`` (the product) drives
libmfft on torch CUDA tensors; tests inject a CPU stand-in to exercise the index
algebra over ``gloo`` without a GPU.
"""
from __future__ import annotations

import ctypes
import threading

from . import _lib
from .engine import Plan


# --------------------------------------------------------------------------- sharding

def shard_range(total: int, world: int, rank: int):
    """Contiguous [lo, hi) of `total` independent units owned by `rank`: the first
    total % world ranks take one extra unit."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# --------------------------------------------------------------------------- communication

class Comm:
    """The exchange step.  With a single rank it degenerates to handing the buffer back."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            self.backend = dist.get_backend(group)
        else:
            self.world, self.rank, self.backend = 1, 0, "none"

    def all_to_all(self, dst, src):
        """Exchange equal chunks: the r-th chunk of the result is rank r's src[my chunk].
        Returns the tensor that holds the result -- `dst`, or `src` itself when there is only
        one rank (no copy); the other one is free for reuse by the caller."""
        import torch
        if self.world == 1:
            return src
        d, s = torch.view_as_real(dst), torch.view_as_real(src)
        if self.backend == "gloo" and src.is_cuda:
            # rehearsal mode (several ranks sharing one GPU box): stage through the host
            sc = s.cpu()
            dc = torch.empty_like(sc)
            self._dist.all_to_all_single(dc, sc, group=self.group)
            d.copy_(dc)
        else:
            self._dist.all_to_all_single(d, s, group=self.group)
        return dst


# --------------------------------------------------------------------------- local operations

class HipOps:
    """Local transforms / re-layouts on torch CUDA tensors through the C ABI."""

    def __init__(self):
        self._plans = {}
        self._lock = threading.Lock()

    @staticmethod
    def _dtype_of(t):
        import torch
        if t.dtype == torch.complex64:
            return _lib.C64
        if t.dtype == torch.complex128:
            return _lib.C128
        raise TypeError(f"complex tensor expected, got {t.dtype}")

    @staticmethod
    def _stream():
        import torch
        return torch.cuda.current_stream().cuda_stream

    def empty(self, count, like):
        import torch
        return torch.empty(int(count), dtype=like.dtype, device=like.device)

    def fft(self, src, dst, n, batch, in_layout=None, out_layout=None):
        dtype = self._dtype_of(src)
        key = (n, batch, dtype, in_layout, out_layout, src.device.index)
        with self._lock:
            plan = self._plans.get(key)
            if plan is None:
                plan = Plan(n, batch=batch, dtype=dtype, in_layout=in_layout, out_layout=out_layout)
                self._plans[key] = plan
        plan.exec(src, dst, stream=self._stream())

    def transpose(self, src, dst, rows, cols, batch=1):
        _lib.check(_lib.load().mfft_transpose(src.data_ptr(), dst.data_ptr(), rows, cols, batch,
                                              self._dtype_of(src), self._stream() or None))

    def twiddle(self, x, rows, cols, r0, c0, n_total):
        _lib.check(_lib.load().mfft_twiddle_tile(x.data_ptr(), rows, cols, r0, c0, n_total,
                                                 self._dtype_of(x), self._stream() or None))

    def close(self):
        with self._lock:
            for p in self._plans.values():
                p.destroy()
            self._plans.clear()


# --------------------------------------------------------------------------- 2-D slab / pencil

def fft2_slab(a_local, n0, n1, comm: Comm, ops, restore_layout=False):
    """2-D forward DFT of an n0 x n1 array distributed by row slabs.

    ``a_local``: this rank's n0/P consecutive rows, flat complex tensor of (n0/P)*n1
    elements (row-major).  Returns a flat tensor of the same size:

    * ``restore_layout=False`` (one all-to-all): the TRANSPOSED spectrum in column
      slabs -- out[c*n0 + k0] = F[k0][rank*(n1/P) + c];
    * ``restore_layout=True`` (two all-to-alls): the spectrum in the input's own row-slab
      layout -- out[i*n1 + k1] = F[rank*(n0/P) + i][k1].
    """
    P = comm.world
    if n0 % P or n1 % P:
        raise ValueError(f"both dimensions must divide by the number of ranks ({P})")
    R, W = n0 // P, n1 // P
    if a_local.numel() != R * n1:
        raise ValueError("a_local has the wrong size for this rank's slab")
    x = ops.empty(R * n1, a_local)
    y = ops.empty(R * n1, a_local)

    def other(t):
        return y if t is x else x

    # row FFTs; bin k of local row i is stored at [k // W][i][k % W]: already packed per destination
    ops.fft(a_local, x, n=n1, batch=R, out_layout=(W, W, R * W))
    cur = comm.all_to_all(y, x)                    # cur[r][i][c]: an (n0 x W) matrix [global row][local col]
    nxt = other(cur)
    ops.transpose(cur, nxt, rows=n0, cols=W)       # -> (W x n0): each local column contiguous
    cur, nxt = nxt, cur
    if not restore_layout:
        ops.fft(cur, nxt, n=n0, batch=W)
        return nxt
    # column FFTs; bin k0 of local column c goes to [k0 // R][c][k0 % R], packed for the way back
    ops.fft(cur, nxt, n=n0, batch=W, out_layout=(R, R, W * R))
    cur = comm.all_to_all(cur, nxt)                # cur[s][c][i]: an (n1 x R) matrix [global col][local row]
    nxt = other(cur)
    ops.transpose(cur, nxt, rows=n1, cols=R)       # -> (R x n1)
    return nxt


# --------------------------------------------------------------------------- large 1-D four-step

def fft1d_fourstep(x_local, n1, n2, comm: Comm, ops, natural_order=True):
    """Forward DFT of ONE signal of n = n1*n2 points distributed in contiguous chunks.

    ``x_local``: this rank's n/P consecutive samples (flat complex tensor).  With
    j = j1*n2 + j2 and k = k1 + n1*k2:

        X[k1 + n1*k2] = sum_j2 W_n2^(j2 k2) * W_n^(j2 k1) * sum_j1 W_n1^(j1 k1) x[j1*n2 + j2]

    Returns a flat tensor of n/P elements:

    * ``natural_order=True`` (three all-to-alls): X[rank*n/P + m];
    * ``natural_order=False`` (two all-to-alls): out[i*n2 + k2] = X[(rank*(n1/P) + i) + n1*k2].
    """
    P = comm.world
    if n1 % P or n2 % P:
        raise ValueError(f"both factors must divide by the number of ranks ({P})")
    n = n1 * n2
    R1, W2 = n1 // P, n2 // P
    if x_local.numel() != R1 * n2:
        raise ValueError("x_local has the wrong size for this rank's chunk")
    x = ops.empty(R1 * n2, x_local)
    y = ops.empty(R1 * n2, x_local)

    def other(t):
        return y if t is x else x

    # local rows j1 (R1 x n2) -> (n2 x R1): destination s owns rows [s*W2, (s+1)*W2), contiguous
    ops.transpose(x_local, x, rows=R1, cols=n2)
    cur = comm.all_to_all(y, x)                    # cur[r][c][i] = x[(r*R1 + i)*n2 + rank*W2 + c]
    nxt = other(cur)
    # n1-point FFT over j1 for each local column c.  Sample j1 sits at [j1 // R1][c][j1 % R1];
    # bin k1 is stored at [k1 // R1][c][k1 % R1], i.e. packed for the owner of that k1 block.
    seg = (R1, R1, W2 * R1)
    ops.fft(cur, nxt, n=n1, batch=W2, in_layout=seg, out_layout=seg)
    cur = comm.all_to_all(cur, nxt)                # cur[s][c][i] = A[j2 = s*W2 + c][k1 = rank*R1 + i]: (n2 x R1)
    nxt = other(cur)
    ops.twiddle(cur, rows=n2, cols=R1, r0=0, c0=comm.rank * R1, n_total=n)
    ops.transpose(cur, nxt, rows=n2, cols=R1)      # -> (R1 x n2): each local k1 row contiguous in j2
    cur, nxt = nxt, cur
    if not natural_order:
        ops.fft(cur, nxt, n=n2, batch=R1)
        return nxt
    # n2-point FFT over j2; bin k2 of local row i goes to [k2 // W2][i][k2 % W2] (owner of natural index k2*n1 + k1)
    ops.fft(cur, nxt, n=n2, batch=R1, out_layout=(W2, W2, R1 * W2))
    cur = comm.all_to_all(cur, nxt)                # cur[r][i][c] = X[(rank*W2 + c)*n1 + r*R1 + i]: (n1 x W2)
    nxt = other(cur)
    ops.transpose(cur, nxt, rows=n1, cols=W2)      # -> (W2 x n1) = natural-order chunk
    return nxt
