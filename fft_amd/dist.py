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
from .engine import ColumnPlan, Plan


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
        # When set to a list, every exchange appends (label, start event, end event) recorded on
        # the current stream (device tensors only): per-exchange times for the benchmark.
        self.timings = None

    def _mark(self, like):
        if self.timings is None or not getattr(like, "is_cuda", False):
            return None
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def all_to_all(self, dst, src, label="all_to_all"):
        """Exchange equal chunks: the r-th chunk of the result is rank r's src[my chunk].
        Returns the tensor that holds the result -- `dst`, or `src` itself when there is only
        one rank (no copy); the other one is free for reuse by the caller."""
        import torch
        if self.world == 1:
            return src
        e0 = self._mark(src)
        try:
            return self._all_to_all(dst, src)
        finally:
            if e0 is not None:
                self.timings.append((label, e0, self._mark(src)))

    def _all_to_all(self, dst, src):
        import torch
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


    def start_exchange(self, send_blocks, recv_blocks, label="exchange"):
        """Begin a personalised exchange of contiguous blocks: send_blocks[s] goes to rank s and
        recv_blocks[r] receives what rank r sends here (the own block is copied locally).  Returns
        a handle whose wait() completes it.  Point-to-point sends and receives in one group (what an
        all-to-all is made of), so a caller can start the exchange of one chunk and keep
        transforming the next while it is on the links."""
        import torch
        dist = self._dist
        e0 = self._mark(send_blocks[0])
        recv_blocks[self.rank].copy_(send_blocks[self.rank])
        if self.world == 1:
            return _Exchange([], [])
        staged = self.backend == "gloo" and send_blocks[0].is_cuda       # rehearsal: through the host
        ops, post = [], []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            dst_rank = peer if self.group is None else dist.get_global_rank(self.group, peer)
            snd = torch.view_as_real(send_blocks[peer])
            rcv = torch.view_as_real(recv_blocks[peer])
            if staged:
                snd = snd.cpu()
                tmp = torch.empty(rcv.shape, dtype=rcv.dtype)
                post.append((rcv, tmp))
                rcv = tmp
            ops.append(dist.P2POp(dist.isend, snd, dst_rank, self.group))
            ops.append(dist.P2POp(dist.irecv, rcv, dst_rank, self.group))
        done = None
        if e0 is not None:
            done = lambda: self.timings.append((label, e0, self._mark(send_blocks[0])))   # noqa: E731
        return _Exchange(dist.batch_isend_irecv(ops), post, done)


class _Exchange:
    def __init__(self, works, post, done=None):
        self._works, self._post, self._done = works, post, done

    def wait(self):
        for w in self._works:
            w.wait()
        for dst, src in self._post:
            dst.copy_(src)
        if self._done is not None:
            self._done()
        self._works, self._post, self._done = [], [], None


def _chunk_count(requested, rows, world):
    """How many row chunks the first exchange is cut into: the largest divisor of `rows` not
    above the request (default 4 with several ranks, MFFT_DIST_CHUNKS overrides), 1 for one rank."""
    import os
    if world == 1:
        return 1
    if requested is None:
        requested = int(os.environ.get("MFFT_DIST_CHUNKS", "4"))
    c = max(1, min(int(requested), rows))
    while rows % c:
        c -= 1
    return c


# --------------------------------------------------------------------------- C ABI route

class CComm:
    """Communicator of the C ABI (mfft_comm): RCCL inside libmfft.so, no torch.distributed on the
    data path.  `fft2_slab` / `fft1d_fourstep` given a CComm run as one mfft_dist_exec call."""

    def __init__(self, handle, world, rank):
        self._h, self.world, self.rank, self.backend = handle, int(world), int(rank), "mfft"
        self._plans = {}
        self.timings = None

    @classmethod
    def from_unique_id(cls, world, rank, uid: bytes):
        """Join an RCCL communicator (collective); `uid` comes from CComm.unique_id() on rank 0."""
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(uid), 128)
        _lib.check(_lib.load().mfft_comm_init_rank(ctypes.byref(h), int(world), int(rank), buf))
        return cls(h, world, rank)

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().mfft_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch(cls, group=None):
        """One rank per process under torch.distributed: rank 0's id is broadcast through the
        process group (any backend), then every rank joins with its current device."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls.from_unique_id(1, 0, cls.unique_id())        # a single process: one rank
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(box, src=src, group=group)
        return cls.from_unique_id(world, rank, box[0])

    @classmethod
    def local(cls, nranks):
        """`nranks` ranks sharing the current device inside this process (one host thread each)."""
        arr = (ctypes.c_void_p * int(nranks))()
        _lib.check(_lib.load().mfft_comm_init_local(arr, int(nranks)))
        return [cls(ctypes.c_void_p(arr[i]), nranks, i) for i in range(int(nranks))]

    def plan(self, kind, a0, a1, dtype, option, chunks):
        key = (kind, int(a0), int(a1), int(dtype), int(bool(option)), chunks)
        h = self._plans.get(key)
        if h is None:
            h = ctypes.c_void_p()
            fn = _lib.load().mfft_plan_2d_dist if kind == "2d" else _lib.load().mfft_plan_1d_dist
            _lib.check(fn(ctypes.byref(h), self._h, int(a0), int(a1), int(dtype), int(bool(option)),
                          int(chunks) if chunks else 0))
            self._plans[key] = h
        return h

    def run(self, kind, src, a0, a1, option, chunks):
        import torch
        dtype = HipOps._dtype_of(src)
        h = self.plan(kind, a0, a1, dtype, option, chunks)
        out = torch.empty_like(src)
        _lib.check(_lib.load().mfft_dist_exec(h, src.data_ptr(), out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream or None))
        return out

    def close(self):
        lib = _lib.load()
        for h in self._plans.values():
            lib.mfft_dist_plan_destroy(h)
        self._plans.clear()
        if self._h is not None:
            lib.mfft_comm_destroy(self._h)
            self._h = None


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

    def _column_plan(self, n, inner, batch, dtype, twiddle, device):
        key = ("col", n, inner, batch, dtype, twiddle, device)
        with self._lock:
            if key not in self._plans:
                try:
                    self._plans[key] = ColumnPlan(n, inner, batch=batch, dtype=dtype, twiddle=twiddle)
                except _lib.MfftError:
                    self._plans[key] = None          # shape outside the tile passes' range
            return self._plans[key]

    def columns_ok(self, n, inner, like, batch=1):
        """Whether fft_columns serves an (n x inner) array of this dtype."""
        return self._column_plan(n, inner, batch, self._dtype_of(like), None, like.device.index) is not None

    def fft_columns(self, src, dst, n, inner, batch=1, twiddle=None):
        """dst[b][k][c] = sum_j src[b][j][c] exp(-2 pi i j k / n), times exp(-2 pi i k (c0 + c) / n_total)
        when twiddle = (n_total, c0); the columns are transformed where they lie."""
        plan = self._column_plan(n, inner, batch, self._dtype_of(src), twiddle, src.device.index)
        if plan is None:
            raise _lib.MfftError(f"no column-batched plan for n={n}, inner={inner}")
        plan.exec(src, dst, stream=self._stream())

    def transpose(self, src, dst, rows, cols, batch=1):
        _lib.check(_lib.load().mfft_transpose(src.data_ptr(), dst.data_ptr(), rows, cols, batch,
                                              self._dtype_of(src), self._stream() or None))

    def swap_axes(self, src, dst, a, b, w):
        """dst viewed as (b, a, w) = src viewed as (a, b, w) with the two leading axes swapped."""
        _lib.check(_lib.load().mfft_swap_axes(src.data_ptr(), dst.data_ptr(), a, b, w,
                                              self._dtype_of(src), self._stream() or None))

    def twiddle(self, x, rows, cols, r0, c0, n_total):
        _lib.check(_lib.load().mfft_twiddle_tile(x.data_ptr(), rows, cols, r0, c0, n_total,
                                                 self._dtype_of(x), self._stream() or None))

    def close(self):
        with self._lock:
            for p in self._plans.values():
                if p is not None:
                    p.destroy()
            self._plans.clear()


class _Buffers:
    """Two scratch buffers of the local size; `free(*busy)` hands out one that is none of `busy`."""

    def __init__(self, ops, count, like):
        self._bufs = [ops.empty(count, like), ops.empty(count, like)]

    def free(self, *busy):
        for b in self._bufs:
            if all(b is not t for t in busy):
                return b
        raise RuntimeError("no free scratch buffer")


# --------------------------------------------------------------------------- 2-D slab / pencil

def fft2_slab(a_local, n0, n1, comm: Comm, ops, restore_layout=False, chunks=None):
    """2-D forward DFT of an n0 x n1 array distributed by row slabs.

    ``a_local``: this rank's n0/P consecutive rows, flat complex tensor of (n0/P)*n1
    elements (row-major).  Returns a flat tensor of the same size:

    * ``restore_layout=False`` (one all-to-all): the spectrum in column slabs, all n0 rows
      of this rank's n1/P columns -- out[k0*(n1/P) + c] = F[k0][rank*(n1/P) + c];
    * ``restore_layout=True`` (two all-to-alls): the spectrum in the input's own row-slab
      layout -- out[i*n1 + k1] = F[rank*(n0/P) + i][k1].

    Local passes over the slab: row FFTs (stored already packed per destination), then the
    column FFTs on the strided columns where the exchange left them (no transpose), then --
    only with P > 1 and restore_layout -- one block copy that unpacks the way back.

    With several ranks the first exchange is pipelined: the rows are transformed in `chunks`
    groups, and a group's blocks travel (grouped point-to-point sends, each landing as a
    contiguous row range of the column matrix) while the next group is transformed.
    """
    P = comm.world
    if n0 % P or n1 % P:
        raise ValueError(f"both dimensions must divide by the number of ranks ({P})")
    R, W = n0 // P, n1 // P
    if a_local.numel() != R * n1:
        raise ValueError("a_local has the wrong size for this rank's slab")
    if isinstance(comm, CComm):                    # the whole decomposition inside libmfft.so, RCCL exchanges
        return comm.run("2d", a_local, n0, n1, restore_layout, chunks)
    bufs = _Buffers(ops, R * n1, a_local)

    # row FFTs; bin k of local row i is stored at [k // W][i][k % W]: already packed per destination
    x = bufs.free(a_local)
    C = _chunk_count(chunks, R, P)
    if C == 1:
        ops.fft(a_local, x, n=n1, batch=R, out_layout=(W, W, R * W))
        cur = comm.all_to_all(bufs.free(x), x, "rows->columns")     # cur[r][i][c]: (n0 x W), row r*R + i, local column c
    else:
        Rc = R // C
        cur = bufs.free(x)
        pending = []
        for c in range(C):
            sb = x[c * Rc * n1:(c + 1) * Rc * n1]  # this group's send buffer: [s][i][col], P x Rc x W
            ops.fft(a_local[c * Rc * n1:(c + 1) * Rc * n1], sb, n=n1, batch=Rc, out_layout=(W, W, Rc * W))
            sends = [sb[s * Rc * W:(s + 1) * Rc * W] for s in range(P)]
            recvs = [cur[(r * R + c * Rc) * W:(r * R + (c + 1) * Rc) * W] for r in range(P)]
            pending.append(comm.start_exchange(sends, recvs, f"rows->columns[{c}/{C}]"))
        for h in pending:
            h.wait()
    nxt = bufs.free(cur)
    if ops.columns_ok(n0, W, cur):
        ops.fft_columns(cur, nxt, n=n0, inner=W)
        cur = nxt
    else:                                          # shapes outside the tile passes: go through a transpose
        ops.transpose(cur, nxt, rows=n0, cols=W)   # (W x n0)
        ops.fft(nxt, cur, n=n0, batch=W)
        ops.transpose(cur, nxt, rows=W, cols=n0)   # (n0 x W)
        cur = nxt
    # cur[k0][c] = F[k0][rank*W + c]
    if not restore_layout:
        return cur
    if P == 1:
        return cur
    # way back: rows [s*R, (s+1)*R) belong to rank s and are contiguous; they travel in row chunks, and
    # a chunk that has arrived ([s][i][c], P x Rc x W) is unpacked into its Rc output rows while the
    # next one is on the links
    C2 = _chunk_count(chunks, R, P)
    Rc, piece = R // C2, (R // C2) * W
    res = bufs.free(cur)
    out = ops.empty(R * n1, a_local)
    pending = []
    for c in range(C2):
        sends = [cur[(s * R + c * Rc) * W:(s * R + c * Rc) * W + piece] for s in range(P)]
        recvs = [res[(c * P + r) * piece:(c * P + r + 1) * piece] for r in range(P)]
        pending.append(comm.start_exchange(sends, recvs, f"columns->rows[{c}/{C2}]"))
    for c, h in enumerate(pending):
        h.wait()
        ops.swap_axes(res[c * P * piece:], out[c * Rc * n1:], a=P, b=Rc, w=W)     # out[i][s][c]: Rc rows of n1
    return out


def fourstep_factors(log2n: int, complex128: bool = True):
    """(n1, n2) with n1*n2 = 2^log2n for `fft1d_fourstep`: n2 as long as ONE LDS-resident kernel
    transforms (8192 complex128 / 16384 complex64 points), so the row step is a single pass over the
    data and the strided column step takes the rest in two; an even split makes both steps two-pass
    (measured at 2^31 complex128 on 8 ranks' shapes: 8.2 -> 6.4 ms of local work per GPU)."""
    cap = 13 if complex128 else 14
    l2 = min(cap, log2n // 2)
    return 1 << (log2n - l2), 1 << l2


# --------------------------------------------------------------------------- large 1-D four-step

def fft1d_fourstep(x_local, n1, n2, comm: Comm, ops, natural_order=True, chunks=None):
    """Forward DFT of ONE signal of n = n1*n2 points distributed in contiguous chunks.

    ``x_local``: this rank's n/P consecutive samples (flat complex tensor).  With
    j = j1*n2 + j2 and k = k1 + n1*k2:

        X[k1 + n1*k2] = sum_j2 W_n2^(j2 k2) * W_n^(j2 k1) * sum_j1 W_n1^(j1 k1) x[j1*n2 + j2]

    Returns a flat tensor of n/P elements:

    * ``natural_order=True`` (three all-to-alls): X[rank*n/P + m];
    * ``natural_order=False`` (two all-to-alls): out[i*n2 + k2] = X[(rank*(n1/P) + i) + n1*k2].

    Local passes: one block copy that packs the first exchange, the n1-point FFTs on the
    strided columns with the inter-step twiddle folded into their last store, the n2-point
    FFTs reading the second exchange's blocks in place (segmented load), and for natural
    order one transpose after the third exchange.  With several ranks the first exchange is
    pipelined in `chunks` row groups like fft2_slab's.
    """
    P = comm.world
    if n1 % P or n2 % P:
        raise ValueError(f"both factors must divide by the number of ranks ({P})")
    n = n1 * n2
    R1, W2 = n1 // P, n2 // P
    if x_local.numel() != R1 * n2:
        raise ValueError("x_local has the wrong size for this rank's chunk")
    if isinstance(comm, CComm):
        return comm.run("1d", x_local, n1, n2, natural_order, chunks)
    bufs = _Buffers(ops, R1 * n2, x_local)

    # pack: local rows j1 (R1 x P x W2) -> [s][i][c]; destination s owns columns [s*W2, (s+1)*W2)
    if P == 1:
        cur = x_local
    else:
        x = bufs.free(x_local)
        C = _chunk_count(chunks, R1, P)
        if C == 1:
            ops.swap_axes(x_local, x, a=R1, b=P, w=W2)
            cur = comm.all_to_all(bufs.free(x), x, "samples->columns")     # cur[r][i][c] = x[(r*R1 + i)*n2 + rank*W2 + c]: (n1 x W2)
        else:
            Rc = R1 // C
            cur = bufs.free(x)
            pending = []
            for c in range(C):
                sb = x[c * Rc * n2:(c + 1) * Rc * n2]      # [s][i][col], P x Rc x W2
                ops.swap_axes(x_local[c * Rc * n2:(c + 1) * Rc * n2], sb, a=Rc, b=P, w=W2)
                sends = [sb[s * Rc * W2:(s + 1) * Rc * W2] for s in range(P)]
                recvs = [cur[(r * R1 + c * Rc) * W2:(r * R1 + (c + 1) * Rc) * W2] for r in range(P)]
                pending.append(comm.start_exchange(sends, recvs, f"samples->columns[{c}/{C}]"))
            for h in pending:
                h.wait()
    nxt = bufs.free(cur)
    if ops.columns_ok(n1, W2, cur):
        # A[k1][c] = W_n^(k1 * (rank*W2 + c)) * sum_j1 W_n1^(j1 k1) cur[j1][c]
        ops.fft_columns(cur, nxt, n=n1, inner=W2, twiddle=(n, comm.rank * W2))
        cur = nxt
    else:
        ops.transpose(cur, nxt, rows=n1, cols=W2)  # (W2 x n1): each local column contiguous
        tmp = bufs.free(nxt, x_local)
        ops.fft(nxt, tmp, n=n1, batch=W2)
        ops.twiddle(tmp, rows=W2, cols=n1, r0=comm.rank * W2, c0=0, n_total=n)
        cur = bufs.free(tmp, x_local)
        ops.transpose(tmp, cur, rows=W2, cols=n1)  # (n1 x W2)
    seg_in = (W2, W2, R1 * W2)      # sample j2 of local row i sits at [j2 // W2][i][j2 % W2]
    if P == 1:
        out = bufs.free(cur, x_local)
        ops.fft(cur, out, n=n2, batch=R1)
        if not natural_order:
            return out
        res = bufs.free(out, x_local)
        ops.transpose(out, res, rows=n1, cols=W2)
        return res
    # Second exchange + n2-point transforms, chunked over this rank's rows k1: rows k1 in [s*R1, (s+1)*R1)
    # belong to rank s and are contiguous; the blocks of chunk c (rows [c*Rc, +Rc) from every peer) are
    # transformed while chunk c + 1 travels.  For natural order bin k2 of row i goes to
    # [k2 // W2][i][k2 % W2] (the owner of natural index k2*n1 + k1) and chunk c leaves for its owners
    # -- into the regions of `res` that the transform has just consumed -- while chunk c + 1 is transformed.
    C2 = _chunk_count(chunks, R1, P)
    Rc, piece = R1 // C2, (R1 // C2) * W2
    res = bufs.free(cur, x_local)
    extra = ops.empty(R1 * n2, x_local)

    def second(c):
        sends = [cur[(s * R1 + c * Rc) * W2:(s * R1 + c * Rc) * W2 + piece] for s in range(P)]
        recvs = [res[r * R1 * W2 + c * piece:r * R1 * W2 + (c + 1) * piece] for r in range(P)]
        return comm.start_exchange(sends, recvs, f"columns->rows[{c}/{C2}]")

    handle, third = second(0), []
    for c in range(C2):
        ahead = second(c + 1) if c + 1 < C2 else None
        handle.wait()
        if natural_order:
            ops.fft(res[c * piece:], extra[c * piece:], n=n2, batch=Rc, in_layout=seg_in, out_layout=seg_in)
            sends = [extra[s * R1 * W2 + c * piece:s * R1 * W2 + (c + 1) * piece] for s in range(P)]
            recvs = [res[r * R1 * W2 + c * piece:r * R1 * W2 + (c + 1) * piece] for r in range(P)]
            third.append(comm.start_exchange(sends, recvs, f"natural order[{c}/{C2}]"))
        else:
            ops.fft(res[c * piece:], extra[c * Rc * n2:], n=n2, batch=Rc, in_layout=seg_in)
        handle = ahead
    if not natural_order:
        return extra                               # out[i*n2 + k2] = X[(rank*R1 + i) + n1*k2]
    for h in third:
        h.wait()
    # res[r][i][c] = X[(rank*W2 + c)*n1 + r*R1 + i]: an (n1 x W2) matrix whose transpose is the natural-order chunk
    ops.transpose(res, cur, rows=n1, cols=W2)
    return cur
