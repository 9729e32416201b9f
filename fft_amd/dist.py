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

    def plan(self, kind, a0, a1, dtype, option, chunks, col_groups=None):
        key = (kind, int(a0), int(a1), int(dtype), int(bool(option)), chunks, col_groups)
        h = self._plans.get(key)
        if h is None:
            h = ctypes.c_void_p()
            fn = _lib.load().mfft_plan_2d_dist if kind == "2d" else _lib.load().mfft_plan_1d_dist
            _lib.check(fn(ctypes.byref(h), self._h, int(a0), int(a1), int(dtype), int(bool(option)),
                          int(chunks) if chunks else 0, int(col_groups) if col_groups else 0))
            self._plans[key] = h
        return h

    def plan_info(self, h):
        """(row chunks, column groups, workspace bytes) of a distributed plan"""
        ch, gr, ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        _lib.check(_lib.load().mfft_dist_plan_info(h, ctypes.byref(ch), ctypes.byref(gr), ctypes.byref(ws)))
        return ch.value, gr.value, ws.value

    def run(self, kind, src, a0, a1, option, chunks, col_groups=None):
        import torch
        dtype = HipOps._dtype_of(src)
        h = self.plan(kind, a0, a1, dtype, option, chunks, col_groups)
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

    def _column_plan(self, n, inner, batch, dtype, twiddle, device, out_ld=None):
        if out_ld == inner:
            out_ld = None
        key = ("col", n, inner, batch, dtype, twiddle, device, out_ld)
        with self._lock:
            if key not in self._plans:
                try:
                    self._plans[key] = ColumnPlan(n, inner, batch=batch, dtype=dtype, twiddle=twiddle, out_ld=out_ld)
                except _lib.MfftError:
                    self._plans[key] = None          # shape outside the tile passes' range
            return self._plans[key]

    @staticmethod
    def min_group_cols(like):
        """Narrowest column group of the pipelined schedules: the shortest segments the contiguous-axis
        kernels read and write in place (64 complex128 / 128 complex64 elements)."""
        return 64 if HipOps._dtype_of(like) == _lib.C128 else 128

    def columns_ok(self, n, inner, like, batch=1, out_ld=None):
        """Whether fft_columns serves an (n x inner) array of this dtype."""
        return self._column_plan(n, inner, batch, self._dtype_of(like), None, like.device.index, out_ld) is not None

    def fft_columns(self, src, dst, n, inner, batch=1, twiddle=None, out_ld=None):
        """dst[b][k][c] = sum_j src[b][j][c] exp(-2 pi i j k / n), times exp(-2 pi i k (c0 + c) / n_total)
        when twiddle = (n_total, c0); the columns are transformed where they lie.  out_ld: elements
        between consecutive rows of dst (default inner: dense)."""
        plan = self._column_plan(n, inner, batch, self._dtype_of(src), twiddle, src.device.index, out_ld)
        if plan is None:
            raise _lib.MfftError(f"no column-batched plan for n={n}, inner={inner}")
        plan.exec(src, dst, stream=self._stream())

    def transpose(self, src, dst, rows, cols, batch=1):
        _lib.check(_lib.load().mfft_transpose(src.data_ptr(), dst.data_ptr(), rows, cols, batch,
                                              self._dtype_of(src), self._stream() or None))

    def swap_axes(self, src, dst, a, b, w, strides=None):
        """dst viewed as (b, a, w): dst[j][i][:] = src[i*sa + j*sb + (0..w)], (sa, sb) = strides
        (default (b*w, w): src viewed as (a, b, w) with the two leading axes swapped)."""
        sa, sb = strides if strides is not None else (b * w, w)
        _lib.check(_lib.load().mfft_swap_axes_strided(src.data_ptr(), dst.data_ptr(), a, b, w, sa, sb,
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


def _group_count(requested, cols, world, floor):
    """Column groups of the pipelined schedules: the largest divisor of this rank's `cols` columns not
    above the request (default 2, MFFT_DIST_GROUPS overrides) that leaves groups of at least `floor`
    columns; 1 for one rank.  (The step behind the last exchange hides only behind the last group's
    exchange, so more groups shorten the window it has: tools/local_work_bench.py.)"""
    import os
    if world == 1:
        return 1
    if requested is None:
        requested = int(os.environ.get("MFFT_DIST_GROUPS", "2"))
    g = max(1, min(int(requested), cols))
    while g > 1 and (cols % g or cols // g < floor):
        g -= 1
    return g


def _wait_all(handles):
    for h in handles:
        h.wait()


# --------------------------------------------------------------------------- 2-D slab / pencil

def fft2_slab(a_local, n0, n1, comm: Comm, ops, restore_layout=False, chunks=None, col_groups=None):
    """2-D forward DFT of an n0 x n1 array distributed by row slabs.

    ``a_local``: this rank's n0/P consecutive rows, flat complex tensor of (n0/P)*n1
    elements (row-major).  Returns a flat tensor of the same size:

    * ``restore_layout=False`` (one exchange): the spectrum in column slabs, all n0 rows
      of this rank's n1/P columns -- out[k0*(n1/P) + c] = F[k0][rank*(n1/P) + c];
    * ``restore_layout=True`` (two exchanges): the spectrum in the input's own row-slab
      layout -- out[i*n1 + k1] = F[rank*(n0/P) + i][k1].

    Schedule with several ranks (mfft_plan_2d_dist runs the same one inside libmfft.so).  The
    columns are cut into P*G segments of Wc columns; rank q transforms, as its group g, segment
    q*G + g (column slabs out: its columns must be one contiguous range) or g*P + q (row slabs
    out: any assignment serves, and this one frees a group's buffers as soon as it has moved on).

    * rows: bin k of local row i is stored at [k // Wc][i][k % Wc], so every (peer, group) block is
      contiguous; the rows are transformed in `chunks` chunks;
    * first exchange, group-major: the first group's chunks leave as their rows are transformed,
      the other groups follow whole;
    * columns: a group that has arrived is transformed along the strided axis (no transpose) while
      the next one travels -- straight into its column block of the result, or, with
      ``restore_layout``, into the send buffer of the way back, which leaves while the next group
      is transformed;
    * unpack: the last group travels back in row chunks, and a chunk's rows are unpacked into the
      result while the next chunk is on the links.

    ``col_groups=1`` is the schedule that chunks by rows only.
    """
    P = comm.world
    if n0 % P or n1 % P:
        raise ValueError(f"both dimensions must divide by the number of ranks ({P})")
    R, W = n0 // P, n1 // P
    if a_local.numel() != R * n1:
        raise ValueError("a_local has the wrong size for this rank's slab")
    if isinstance(comm, CComm):                    # the whole decomposition inside libmfft.so, RCCL exchanges
        return comm.run("2d", a_local, n0, n1, restore_layout, chunks, col_groups)
    bufs = _Buffers(ops, R * n1, a_local)
    X = bufs.free(a_local)
    Y = bufs.free(X)

    def columns(src, dst, wc, out_ld=None):
        """n0-point transforms down the wc columns of src (n0 x wc) into dst (rows out_ld apart)"""
        if ops.columns_ok(n0, wc, src, out_ld=out_ld):
            ops.fft_columns(src, dst, n=n0, inner=wc, out_ld=out_ld)
            return
        assert wc == W and out_ld in (None, W)     # shapes outside the tile passes: one group, through a transpose
        ops.transpose(src, X, rows=n0, cols=W)     # (W x n0)
        ops.fft(X, src, n=n0, batch=W)
        ops.transpose(src, dst, rows=W, cols=n0)   # (n0 x W)

    if P == 1:
        ops.fft(a_local, Y, n=n1, batch=R)
        out = ops.empty(R * n1, a_local)
        columns(Y, out, W)
        return out
    G = _group_count(col_groups, W, P, ops.min_group_cols(a_local))
    while G > 1 and not ops.columns_ok(n0, W // G, a_local, out_ld=None if restore_layout else W):
        G -= 1
        while G > 1 and W % G:
            G -= 1
    C = _chunk_count(chunks, R, P)
    Rc, Wc = R // C, W // G
    blk, gsz = R * Wc, P * R * Wc                  # one (peer, group) block; one group's (n0 x Wc) column matrix
    seg_of = (lambda q, g: g * P + q) if restore_layout else (lambda q, g: q * G + g)

    def unit(src_of, g, c, nck, label):
        """exchange of chunk c (of nck) of group g: rows [c*R/nck, (c+1)*R/nck) of every (peer, group) block"""
        piece = blk // nck
        sends = [src_of(q)[c * piece:(c + 1) * piece] for q in range(P)]
        recvs = [Y[g * gsz + q * blk + c * piece:g * gsz + q * blk + (c + 1) * piece] for q in range(P)]
        return comm.start_exchange(sends, recvs, label)

    arriving = {g: [] for g in range(G)}
    for c in range(C):
        ops.fft(a_local[c * Rc * n1:(c + 1) * Rc * n1], X[c * Rc * Wc:], n=n1, batch=Rc, out_layout=(Wc, Wc, R * Wc))
        arriving[0].append(unit(lambda q: X[seg_of(q, 0) * blk:], 0, c, C, f"rows->columns[group 0/{G}, chunk {c}/{C}]"))
    for g in range(1, G):
        arriving[g].append(unit(lambda q, g=g: X[seg_of(q, g) * blk:], g, 0, 1, f"rows->columns[group {g}/{G}]"))
    out = ops.empty(R * n1, a_local)
    back = []
    for g in range(G):
        _wait_all(arriving[g])
        if not restore_layout:                     # out[k0][rank's column g*Wc + w]: rows W elements apart
            columns(Y[g * gsz:(g + 1) * gsz], out[g * Wc:], Wc, out_ld=W)
            continue
        # group g's send blocks have left: their region takes the transformed group, whose rows
        # [q*R, (q+1)*R) belong to rank q and are contiguous
        columns(Y[g * gsz:(g + 1) * gsz], X[g * gsz:(g + 1) * gsz], Wc)
        nck = C if g == G - 1 else 1
        for c in range(nck):
            h = unit(lambda q, g=g: X[g * gsz + q * blk:], g, c, nck, f"columns->rows[group {g}/{G}, chunk {c}/{nck}]")
            back.append(h)
    if not restore_layout:
        return out
    # Y[s][i][w] = F[rank*R + i][s*Wc + w]: out[i][s][w], Rc rows at a time as the last group's chunks arrive
    _wait_all(back[:G - 1])
    for c in range(C):
        back[G - 1 + c].wait()
        ops.swap_axes(Y[c * Rc * Wc:], out[c * Rc * n1:], a=P * G, b=Rc, w=Wc, strides=(R * Wc, Wc))
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

def fft1d_fourstep(x_local, n1, n2, comm: Comm, ops, natural_order=True, chunks=None, col_groups=None):
    """Forward DFT of ONE signal of n = n1*n2 points distributed in contiguous chunks.

    ``x_local``: this rank's n/P consecutive samples (flat complex tensor).  With
    j = j1*n2 + j2 and k = k1 + n1*k2:

        X[k1 + n1*k2] = sum_j2 W_n2^(j2 k2) * W_n^(j2 k1) * sum_j1 W_n1^(j1 k1) x[j1*n2 + j2]

    Returns a flat tensor of n/P elements:

    * ``natural_order=True`` (three exchanges): X[rank*n/P + m];
    * ``natural_order=False`` (two exchanges): out[i*n2 + k2] = X[(rank*(n1/P) + i) + n1*k2].

    Schedule with several ranks (mfft_plan_1d_dist runs the same one inside libmfft.so).  The n2
    columns j2 are cut into P*G segments of Wc columns; rank q transforms segment g*P + q as its
    group g.

    * pack + first exchange, group-major: one strided block copy packs group g of the local rows
      per destination; the first group is packed and sent in `chunks` row chunks;
    * columns: the n1-point transforms of a group (inter-step twiddle folded into their last store)
      run while the next group travels; their rows [q*R1, (q+1)*R1) belong to rank q, are
      contiguous, and leave while the next group is transformed -- into the region the group
      arrived in, so sample j2 of local row i ends up at [j2 // Wc][i][j2 % Wc];
    * rows: the last group travels in row chunks, and the n2-point transforms of a chunk's rows read
      those segments in place while the next chunk is on the links;
    * natural order: bin k2 of row i goes to [k2 // W2][i][k2 % W2] (the owner of natural index
      k2*n1 + k1), a chunk leaves for its owners while the next one is transformed, and one
      transpose ends the transform.
    """
    P = comm.world
    if n1 % P or n2 % P:
        raise ValueError(f"both factors must divide by the number of ranks ({P})")
    n = n1 * n2
    R1, W2 = n1 // P, n2 // P
    if x_local.numel() != R1 * n2:
        raise ValueError("x_local has the wrong size for this rank's chunk")
    if isinstance(comm, CComm):
        return comm.run("1d", x_local, n1, n2, natural_order, chunks, col_groups)
    bufs = _Buffers(ops, R1 * n2, x_local)
    X = bufs.free(x_local)
    Y = bufs.free(X)

    def columns(src, dst, wc, c0, scratch):
        """dst[k1][c] = W_n^(k1 * (c0 + c)) * sum_j1 W_n1^(j1 k1) src[j1][c] on an (n1 x wc) matrix"""
        if ops.columns_ok(n1, wc, src):
            ops.fft_columns(src, dst, n=n1, inner=wc, twiddle=(n, c0))
            return
        assert wc == W2                            # shapes outside the tile passes: one group, through transposes
        ops.transpose(src, dst, rows=n1, cols=W2)          # (W2 x n1): each local column contiguous
        ops.fft(dst, scratch, n=n1, batch=W2)
        ops.twiddle(scratch, rows=W2, cols=n1, r0=c0, c0=0, n_total=n)
        ops.transpose(scratch, dst, rows=W2, cols=n1)      # (n1 x W2)

    if P == 1:
        columns(x_local, X, W2, 0, Y)
        ops.fft(X, Y, n=n2, batch=R1)
        if not natural_order:
            return Y
        ops.transpose(Y, X, rows=n1, cols=W2)
        return X
    G = _group_count(col_groups, W2, P, ops.min_group_cols(x_local))
    while G > 1 and not ops.columns_ok(n1, W2 // G, x_local):
        G -= 1
        while G > 1 and W2 % G:
            G -= 1
    C = _chunk_count(chunks, R1, P)
    Rc, Wc = R1 // C, W2 // G
    blk, gsz = R1 * Wc, P * R1 * Wc                # one (peer, group) block; one group's (n1 x Wc) column matrix

    def unit(src_of, g, c, nck, label):
        piece = blk // nck
        sends = [src_of(q)[:piece] for q in range(P)]
        recvs = [Y[g * gsz + q * blk + c * piece:g * gsz + q * blk + (c + 1) * piece] for q in range(P)]
        return comm.start_exchange(sends, recvs, label)

    arriving = {g: [] for g in range(G)}
    for g in range(G):
        nck = C if g == 0 else 1
        rows, piece = R1 // nck, blk // nck
        for c in range(nck):
            pack = X[g * gsz + c * P * piece:g * gsz + (c + 1) * P * piece]     # [q][i][w], P x rows x Wc
            ops.swap_axes(x_local[c * rows * n2 + g * P * Wc:], pack, a=rows, b=P, w=Wc, strides=(n2, Wc))
            arriving[g].append(unit(lambda q, pack=pack, piece=piece: pack[q * piece:], g, c, nck,
                                    f"samples->columns[group {g}/{G}, chunk {c}/{nck}]"))
    out = ops.empty(R1 * n2, x_local)
    landing = []
    for g in range(G):
        _wait_all(arriving[g])
        columns(Y[g * gsz:(g + 1) * gsz], X[g * gsz:(g + 1) * gsz], Wc, (g * P + comm.rank) * Wc, out)
        nck = C if g == G - 1 else 1
        piece = blk // nck
        for c in range(nck):
            landing.append(unit(lambda q, g=g, c=c, piece=piece: X[g * gsz + q * blk + c * piece:], g, c, nck,
                                f"columns->rows[group {g}/{G}, chunk {c}/{nck}]"))
    _wait_all(landing[:G - 1])
    # The natural-order exchange lands in a buffer of its own here: X is still being sent from by the last
    # group's later chunks, and only a transport that runs its transfers in issue order (one RCCL stream,
    # as inside libmfft.so) could reuse it.
    T = ops.empty(R1 * n2, x_local) if natural_order else None
    seg_in = (Wc, Wc, R1 * Wc)                     # sample j2 of local row i sits at [j2 // Wc][i][j2 % Wc]
    seg_out = (W2, W2, R1 * W2)
    run = Rc * W2
    third = []
    for c in range(C):
        landing[G - 1 + c].wait()
        if not natural_order:
            ops.fft(Y[c * Rc * Wc:], out[c * Rc * n2:], n=n2, batch=Rc, in_layout=seg_in)
            continue
        ops.fft(Y[c * Rc * Wc:], out[c * run:], n=n2, batch=Rc, in_layout=seg_in, out_layout=seg_out)
        sends = [out[q * R1 * W2 + c * run:q * R1 * W2 + (c + 1) * run] for q in range(P)]
        recvs = [T[q * R1 * W2 + c * run:q * R1 * W2 + (c + 1) * run] for q in range(P)]
        third.append(comm.start_exchange(sends, recvs, f"natural order[{c}/{C}]"))
    if not natural_order:
        return out                                 # out[i*n2 + k2] = X[(rank*R1 + i) + n1*k2]
    _wait_all(third)
    # T[r][i][c] = X[(rank*W2 + c)*n1 + r*R1 + i]: an (n1 x W2) matrix whose transpose is the natural-order chunk
    ops.transpose(T, out, rows=n1, cols=W2)
    return out
