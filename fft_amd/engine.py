"""Plans over libmfft.so: thin, allocation-free wrappers around the C ABI.

``Plan`` is the Python face of ``mfft_plan_1d``.  Buffers are passed either as
integers (device addresses), as objects exposing ``data_ptr()`` (torch tensors:
torch is used by callers only to own device memory and streams), or -- through
``exec_host`` -- as numpy arrays that the library copies in and out itself.
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np

from . import _lib
from . import hostmem


def _addr(buf) -> int:
    if buf is None:
        return 0
    if isinstance(buf, int):
        return buf
    if hasattr(buf, "data_ptr"):
        return int(buf.data_ptr())
    raise TypeError(f"cannot take a device address from {type(buf).__name__}")


def _check_buffer(buf, count: int, np_dtype, what: str, device=None):
    """Shape / dtype / layout check of a tensor-like device buffer (anything with numel(),
    dtype, is_contiguous()); plain integer addresses cannot be checked and pass through.  A wrong
    buffer would otherwise be read or written out of bounds on the GPU."""
    if buf is None or isinstance(buf, int) or not hasattr(buf, "numel"):
        return
    if not buf.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    if int(buf.numel()) < count:
        raise ValueError(f"{what} holds {int(buf.numel())} elements, the plan needs {count}")
    name = str(buf.dtype).replace("torch.", "")
    if name != np.dtype(np_dtype).name:
        raise ValueError(f"{what} has dtype {name}, the plan needs {np.dtype(np_dtype).name}")
    if not getattr(buf, "is_cuda", True):
        raise ValueError(f"{what} must live in device memory")


class Plan:
    """One batched 1-D transform configuration (see include/mfft.h: mfft_plan_1d)."""

    def __init__(self, n: int, n_in: int | None = None, batch: int = 1, dtype: int = _lib.C64,
                 flags: int = 0, in_layout=None, out_layout=None):
        """in_layout / out_layout: optional (batch_stride, seg, seg_stride) triples in
        elements (see mfft_plan_1d_layout); plans with a layout run on device buffers only."""
        lib = _lib.load()
        self.n = int(n)
        self.n_in = self.n if n_in is None else int(n_in)
        self.batch = int(batch)
        self.dtype = int(dtype)
        self.flags = int(flags)
        self._h = ctypes.c_void_p()
        self._has_layout = in_layout is not None or out_layout is not None
        lin = _lib.Layout(*[int(v) for v in in_layout]) if in_layout is not None else None
        lout = _lib.Layout(*[int(v) for v in out_layout]) if out_layout is not None else None
        _lib.check(lib.mfft_plan_1d_layout(ctypes.byref(self._h), self.n, self.n_in, self.batch,
                                           self.dtype, self.flags,
                                           ctypes.byref(lin) if lin is not None else None,
                                           ctypes.byref(lout) if lout is not None else None))
        self._lock = threading.Lock()

    # -- introspection ------------------------------------------------------
    @property
    def real_dtype(self):
        return np.float32 if self.dtype == _lib.C64 else np.float64

    @property
    def complex_dtype(self):
        return np.complex64 if self.dtype == _lib.C64 else np.complex128

    @property
    def in_dtype(self):
        return self.real_dtype if self.flags & _lib.REAL_IN else self.complex_dtype

    @property
    def out_dtype(self):
        return self.real_dtype if self.flags & _lib.POWER else self.complex_dtype

    @property
    def workspace_bytes(self) -> int:
        v = ctypes.c_int64()
        _lib.check(_lib.load().mfft_plan_workspace_bytes(self._h, ctypes.byref(v)))
        return v.value

    @property
    def launches(self) -> int:
        v = ctypes.c_int()
        _lib.check(_lib.load().mfft_plan_launches(self._h, ctypes.byref(v)))
        return v.value

    # -- execution ------------------------------------------------------------
    def exec(self, d_in, d_out, scale: float = 1.0, mu: float = 0.0, t0: float = 0.0, dt: float = 1.0,
             d_stats=None, stream: int = 0):
        """Asynchronous execution on device buffers (addresses or tensors).  Tensors are checked
        against the plan (dense plans: batch*n_in elements in, batch*n out; plans with a layout
        describe their own extents and are not checked)."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        if not getattr(self, "_has_layout", False):
            inner = getattr(self, "inner", 1)
            _check_buffer(d_in, self.batch * self.n_in * inner, self.in_dtype, "d_in")
            _check_buffer(d_out, self.batch * self.n * inner, self.out_dtype, "d_out")
        if self.flags & _lib.STATS:
            _check_buffer(d_stats, 3 * self.batch, np.float64, "d_stats")
        _lib.check(_lib.load().mfft_exec(self._h, _addr(d_in), _addr(d_out), float(scale), float(mu),
                                         float(t0), float(dt), _addr(d_stats), int(stream) or None))

    def exec_host(self, x: np.ndarray, scale: float = 1.0, mu: float = 0.0, t0: float = 0.0,
                  dt: float = 1.0, stats_only: bool = False):
        """Synchronous execution on a numpy array of shape (batch, n_in) (or (n_in,)
        when batch == 1).  Returns the output array, plus a (batch, 3) float64 array
        {max S, sum S, sum f_k S} when the plan has the STATS flag."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        x = np.ascontiguousarray(x, dtype=self.in_dtype)
        if x.size != self.batch * self.n_in:
            raise ValueError(f"input has {x.size} elements, plan expects {self.batch}x{self.n_in}")
        lead = x.shape[:-1] if x.ndim > 1 else ()
        stats = np.zeros((self.batch, 3), dtype=np.float64) if self.flags & _lib.STATS else None
        if stats_only and stats is None:
            raise ValueError("stats_only needs a plan with the STATS flag")
        out = None if stats_only else np.empty(lead + (self.n,), dtype=self.out_dtype)
        with self._lock:
            _lib.check(_lib.load().mfft_exec_host(
                self._h, x.ctypes.data, out.ctypes.data if out is not None else None, float(scale),
                float(mu), float(t0), float(dt), stats.ctypes.data if stats is not None else None))
        if stats_only:
            return stats
        return (out, stats) if stats is not None else out

    def exec_host_peaks(self, x: np.ndarray, scale: float = 1.0, mu: float = 0.0, t0: float = 0.0,
                        dt: float = 1.0, capacity: int = 65536):
        """exec_host for a one-signal POWER|STATS plan plus the device-side scan for interior
        local maxima.  Returns (S, stats, idx) with idx ascending; when more than `capacity`
        maxima exist idx is their total count instead (an int: call again with that capacity)."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        if not (self.flags & _lib.POWER and self.flags & _lib.STATS) or self.batch != 1:
            raise ValueError("exec_host_peaks needs a one-signal POWER|STATS plan")
        x = np.ascontiguousarray(x, dtype=self.in_dtype)
        if x.size != self.n_in:
            raise ValueError(f"input has {x.size} elements, plan expects {self.n_in}")
        out = np.empty(self.n, dtype=self.out_dtype)
        stats = np.zeros((1, 3), dtype=np.float64)
        idx = np.empty(capacity, dtype=np.int64)
        count = ctypes.c_int64(0)
        with self._lock:
            _lib.check(_lib.load().mfft_exec_host_peaks(
                self._h, x.ctypes.data, out.ctypes.data, float(scale), float(mu), float(t0), float(dt),
                stats.ctypes.data, idx.ctypes.data, int(capacity), ctypes.byref(count)))
        if count.value > capacity:
            return out, stats, int(count.value)
        return out, stats, np.sort(idx[:count.value])

    def exec_host_lines(self, x: np.ndarray, times=None, *, axis_dt: float, guard: float, want: int,
                        dc_guard: float = 0.0, include_dc: bool = False, mode: int = 0, scale: float = 1.0,
                        mu: float = 0.0, t0: float = 0.0, dt: float = 1.0, want_spectrum: bool = True,
                        want_fit: bool = False, normalize: bool = False):
        """Spectra, statistics and the picked / refined spectral lines of every signal of the batch in
        one call (mfft_lines_host).  Returns a dict: S (batch, n) or None (each spectrum divided by its
        guarded maximum on the device when `normalize`), stats (batch, 3) of the unnormalised spectra, idx /
        omega_hat / height (batch, want), count (batch,), gram (batch, 2*want*want + 2*want + 1) or
        None.  `times` (n_in doubles) is needed for the fit only."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        x = np.ascontiguousarray(x, dtype=self.in_dtype)
        if x.size != self.batch * self.n_in:
            raise ValueError(f"input has {x.size} elements, plan expects {self.batch}x{self.n_in}")
        want = int(want)
        prm = _lib.PeakParams(float(axis_dt), float(dc_guard), float(guard), int(bool(include_dc)), int(mode), want)
        B = self.batch
        S = hostmem.empty((B, self.n), self.out_dtype) if want_spectrum else None
        stats = np.zeros((B, 3), dtype=np.float64)
        idx = np.empty((B, max(want, 1)), dtype=np.int64)
        om = np.empty((B, max(want, 1)), dtype=np.float64)
        hgt = np.empty((B, max(want, 1)), dtype=np.float64)
        cnt = np.zeros(B, dtype=np.int32)
        gram = np.empty((B, 2 * want * want + 2 * want + 1), dtype=np.float64) if want_fit else None
        t = None
        if want_fit:
            t = np.ascontiguousarray(times, dtype=np.float64)
            if t.shape[0] != self.n_in:
                raise ValueError("`times` must have one entry per sample")
        with self._lock:
            _lib.check(_lib.load().mfft_lines_host(
                self._h, x.ctypes.data, float(scale), float(mu), float(t0), float(dt),
                t.ctypes.data if t is not None else None, ctypes.byref(prm), int(bool(normalize)),
                S.ctypes.data if S is not None else None, stats.ctypes.data, idx.ctypes.data, om.ctypes.data,
                hgt.ctypes.data, cnt.ctypes.data, gram.ctypes.data if gram is not None else None))
        return {"S": S, "stats": stats, "idx": idx, "omega_hat": om, "height": hgt, "count": cnt, "gram": gram}

    def exec_host_lines_blocks(self, w, c0, vrow, times, *, hbar: float = 1.0, axis_dt: float, guard: float, want: int,
                               dc_guard: float = 0.0, include_dc: bool = False, mode: int = 0, scale: float = 1.0,
                               want_spectrum: bool = True, want_fit: bool = True, normalize: bool = False):
        """exec_host_lines with the samples generated on the device (mfft_lines_blocks_host): signal b is the
        return amplitude sum_k vrow[b, k] exp(-i w[b, k] t / hbar) c0[b, k] on the grid `times`.  w: (batch, nk)
        float64; c0, vrow: (batch, nk) complex128.  Returns the same dict."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        w = np.ascontiguousarray(w, dtype=np.float64)
        c0 = np.ascontiguousarray(c0, dtype=np.complex128)
        vrow = np.ascontiguousarray(vrow, dtype=np.complex128)
        t = np.ascontiguousarray(times, dtype=np.float64)
        B = self.batch
        if w.ndim != 2 or w.shape[0] != B or c0.shape != w.shape or vrow.shape != w.shape:
            raise ValueError(f"block data must be (batch, nk) arrays for a batch of {B}")
        if t.shape[0] != self.n_in:
            raise ValueError("`times` must have one entry per sample")
        want = int(want)
        prm = _lib.PeakParams(float(axis_dt), float(dc_guard), float(guard), int(bool(include_dc)), int(mode), want)
        S = hostmem.empty((B, self.n), self.out_dtype) if want_spectrum else None
        stats = np.zeros((B, 3), dtype=np.float64)
        idx = np.empty((B, max(want, 1)), dtype=np.int64)
        om = np.empty((B, max(want, 1)), dtype=np.float64)
        hgt = np.empty((B, max(want, 1)), dtype=np.float64)
        cnt = np.zeros(B, dtype=np.int32)
        gram = np.empty((B, 2 * want * want + 2 * want + 1), dtype=np.float64) if want_fit else None
        with self._lock:
            _lib.check(_lib.load().mfft_lines_blocks_host(
                self._h, w.ctypes.data, c0.ctypes.data, vrow.ctypes.data, int(w.shape[1]), float(hbar), float(scale),
                t.ctypes.data, ctypes.byref(prm), int(bool(normalize)),
                S.ctypes.data if S is not None else None, stats.ctypes.data, idx.ctypes.data, om.ctypes.data,
                hgt.ctypes.data, cnt.ctypes.data, gram.ctypes.data if gram is not None else None))
        return {"S": S, "stats": stats, "idx": idx, "omega_hat": om, "height": hgt, "count": cnt, "gram": gram}

    def exec_host_extract(self, rough, w, c0, vrow, times, *, hbar: float = 1.0, axis_dt: float, guard: float, want: int,
                          scale: float = 1.0, t0: float = 0.0, dt: float = 1.0, columns_of=None):
        """The device passes of fft_extract_E_p in two calls on samples that stay on the device
        (mfft_lines_extract_host, mfft_lines_gram_host): the return amplitude sum_k vrow[k] exp(-i w[k] t / hbar) c0[k]
        on the grid `times`, the centroid mu from a rough pass through the plan `rough` (None: no demodulation), the
        demodulated spectrum and its |omega|-deduplicated peaks; then, if ``columns_of(result)`` returns line positions,
        the normal equations of the fit on those columns.  Returns the dict of exec_host_lines plus ``mu`` and ``gram``
        (2 m m + 2 m + 1 doubles, or None).  One signal per plan."""
        if self._h is None or (rough is not None and rough._h is None):
            raise ValueError("plan already destroyed")
        w = np.ascontiguousarray(w, dtype=np.float64)
        c0 = np.ascontiguousarray(c0, dtype=np.complex128)
        vrow = np.ascontiguousarray(vrow, dtype=np.complex128)
        t = np.ascontiguousarray(times, dtype=np.float64)
        if self.batch != 1 or w.ndim != 1 or c0.shape != w.shape or vrow.shape != w.shape:
            raise ValueError("block data must be three vectors of one length, for a plan of one signal")
        if t.shape[0] != self.n_in:
            raise ValueError("`times` must have one entry per sample")
        want = int(want)
        prm = _lib.PeakParams(float(axis_dt), 0.0, float(guard), 0, 1, want)
        S = hostmem.empty((1, self.n), self.out_dtype)
        stats = np.zeros((1, 3), dtype=np.float64)
        idx = np.empty((1, max(want, 1)), dtype=np.int64)
        om = np.empty((1, max(want, 1)), dtype=np.float64)
        hgt = np.empty((1, max(want, 1)), dtype=np.float64)
        cnt = np.zeros(1, dtype=np.int32)
        mu = ctypes.c_double(0.0)
        lib = _lib.load()
        with self._lock:
            _lib.check(lib.mfft_lines_extract_host(
                self._h, rough._h if rough is not None else None, w.ctypes.data, c0.ctypes.data, vrow.ctypes.data,
                int(w.shape[0]), float(hbar), float(scale), float(t0), float(dt), t.ctypes.data, ctypes.byref(prm),
                S.ctypes.data, stats.ctypes.data, idx.ctypes.data, om.ctypes.data, hgt.ctypes.data, cnt.ctypes.data,
                ctypes.byref(mu)))
            res = {"S": S, "stats": stats, "idx": idx, "omega_hat": om, "height": hgt, "count": cnt, "mu": float(mu.value),
                   "gram": None}
            cols = columns_of(res) if columns_of is not None else None
            if cols is not None:
                E = np.ascontiguousarray(cols, dtype=np.float64)
                m = int(E.shape[0])
                gram = np.empty(2 * m * m + 2 * m + 1, dtype=np.float64)
                _lib.check(lib.mfft_lines_gram_host(self._h, E.ctypes.data, m, float(mu.value), gram.ctypes.data))
                res["gram"] = gram
        return res

    # -- lifetime ---------------------------------------------------------------
    def destroy(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().mfft_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()


class ColumnPlan(Plan):
    """Column-batched transform (mfft_plan_1d_columns): np.fft.fft(a, axis=-2) on a
    (batch, n, inner) array, on the strided columns where they lie.  ``twiddle=(n_total, c0)``
    also multiplies output (k, c) by exp(-2 pi i k (c0 + c) / n_total).  Raises MfftError
    for shapes the tile passes do not serve (callers then transpose and use Plan)."""

    def __init__(self, n: int, inner: int, batch: int = 1, dtype: int = _lib.C64, twiddle=None, out_ld=None):
        """out_ld: elements between consecutive rows of the result (default inner: dense); a plan with
        out_ld > inner writes a block of columns of a wider array and runs on device buffers only."""
        lib = _lib.load()
        self.n = self.n_in = int(n)
        self.inner = int(inner)
        self.batch = int(batch)
        self.dtype = int(dtype)
        self.flags = 0
        self._h = ctypes.c_void_p()
        tw_n, tw_c0 = (int(twiddle[0]), int(twiddle[1])) if twiddle is not None else (0, 0)
        self.out_ld = int(out_ld) if out_ld else self.inner
        self._has_layout = self.out_ld != self.inner          # the output extent is the caller's: not checked
        _lib.check(lib.mfft_plan_1d_columns_ld(ctypes.byref(self._h), self.n, self.inner, self.batch,
                                               self.dtype, tw_n, tw_c0, self.out_ld))
        self._lock = threading.Lock()

    def exec_host(self, x: np.ndarray):
        """Synchronous execution on a numpy array of batch*n*inner complex elements."""
        if self._h is None:
            raise ValueError("plan already destroyed")
        x = np.ascontiguousarray(x, dtype=self.complex_dtype)
        if x.size != self.batch * self.n * self.inner:
            raise ValueError(f"input has {x.size} elements, plan expects {self.batch}x{self.n}x{self.inner}")
        out = np.empty(x.shape, dtype=self.complex_dtype)
        with self._lock:
            _lib.check(_lib.load().mfft_exec_host(self._h, x.ctypes.data, out.ctypes.data, 1.0, 0.0, 0.0, 1.0, None))
        return out

    def exec_host_peaks(self, *a, **k):
        raise ValueError("column-batched plans have no spectrum scan")
