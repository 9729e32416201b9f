"""numpy-compatible entry points served by the GPU engine.

``fft`` has the signature and semantics of ``numpy.fft.fft`` -- the one call the
reference makes on its spectral path (final2.py:143,
streamlit_app_combined.py:204, :213) -- and ``fftfreq`` those of
``numpy.fft.fftfreq`` (final2.py:144).  Arrays go in and come out as numpy
arrays; the transform itself always runs on the GPU (no host fallback).
"""
from __future__ import annotations

import ctypes
import gc
import operator
import os
import threading
from collections import OrderedDict

import numpy as np

from . import _lib
from .engine import ColumnPlan, Plan

_CACHE_MAX = 24
_cache: "OrderedDict[tuple, Plan]" = OrderedDict()
_cache_bytes: "dict[tuple, int]" = {}
_cache_lock = threading.Lock()
_budget = None


def _cache_budget() -> int:
    """Device bytes the cached plans may hold together: MFFT_CACHE_BYTES, else half of the device's
    memory (a chirp-z plan at the apps' largest length holds ~18 GiB, so a count alone does not
    bound the cache)."""
    global _budget
    if _budget is None:
        env = os.environ.get("MFFT_CACHE_BYTES")
        if env:
            _budget = int(env)
        else:
            free, total = ctypes.c_int64(), ctypes.c_int64()
            rc = _lib.load().mfft_device_memory(ctypes.byref(free), ctypes.byref(total))
            _budget = total.value // 2 if rc == _lib.OK else 1 << 62
    return _budget


def _held_bytes(plan) -> int:
    """Device memory a plan holds or will hold: tables and workspace plus the staging buffers its
    host entry points allocate on first use (input, output)."""
    if plan is None:
        return 0
    csz = 8 if plan.dtype == _lib.C64 else 16
    inner = getattr(plan, "inner", 1)
    in_sz = csz // 2 if plan.flags & _lib.REAL_IN else csz
    out_sz = csz // 2 if plan.flags & _lib.POWER else csz
    return plan.workspace_bytes + plan.batch * inner * (plan.n_in * in_sz + plan.n * out_sz)


def _build(factory):
    """Build a plan; if the device is out of memory, drop every cached plan (each frees its memory
    when its last user lets go) and try once more."""
    try:
        return factory()
    except _lib.MfftError as exc:
        if "out of memory" not in str(exc):
            raise
    clear_plan_cache()
    gc.collect()
    return factory()


def _insert(key, plan):
    """Cache lock held.  Adds the plan, then drops least recently used entries while the cache
    exceeds its entry count or its byte budget (the newest entry always stays).  A dropped plan is
    not destroyed here: another thread may be executing it; it frees its device memory when the
    last reference goes (Plan.__del__)."""
    _cache[key] = plan
    _cache_bytes[key] = _held_bytes(plan)
    budget = _cache_budget()
    while len(_cache) > 1 and (len(_cache) > _CACHE_MAX or sum(_cache_bytes.values()) > budget):
        old, _ = _cache.popitem(last=False)
        _cache_bytes.pop(old, None)


def cached_plan(n: int, n_in: int, batch: int, dtype: int, flags: int, device=None) -> Plan:
    """Plans own device tables and workspace, so they are kept and reused.  The
    cache is shared by all threads (Streamlit runs one script thread per session);
    a plan serialises its own executions.  A plan belongs to the HIP device that is current
    when it is built; `device` only keys the cache for callers that switch devices."""
    key = (int(n), int(n_in), int(batch), int(dtype), int(flags), device)
    with _cache_lock:
        plan = _cache.get(key)
        if plan is not None:
            _cache.move_to_end(key)
            return plan
    plan = _build(lambda: Plan(n, n_in=n_in, batch=batch, dtype=dtype, flags=flags))
    with _cache_lock:
        other = _cache.get(key)
        if other is not None:          # another thread built the same plan meanwhile
            plan.destroy()
            return other
        _insert(key, plan)
    return plan


def cached_column_plan(n: int, inner: int, batch: int, dtype: int, device=None):
    """Column-batched plan for (batch, n, inner) arrays, or None when the tile passes do not
    serve the shape (the caller then moves the axis last)."""
    key = ("columns", int(n), int(inner), int(batch), int(dtype), device)
    with _cache_lock:
        if key in _cache:
            _cache.move_to_end(key)
            return _cache[key]
    try:
        plan = _build(lambda: ColumnPlan(n, inner, batch=batch, dtype=dtype))
    except _lib.MfftError as exc:
        if "out of memory" in str(exc):
            raise
        plan = None
    with _cache_lock:
        other = _cache.get(key)
        if other is not None:
            if plan is not None:
                plan.destroy()
            return other
        _insert(key, plan)
    return plan


def _column_shape(shape, axis, n):
    """(batch, inner) when a transform of the full axis `axis` of a C-contiguous array of this
    shape can run as a column-batched plan: a middle or leading axis, power-of-two extents."""
    nd = len(shape)
    ax = axis % nd
    if ax == nd - 1 or n != shape[ax] or n < 16 or n & (n - 1):
        return None
    inner = int(np.prod(shape[ax + 1:]))
    batch = int(np.prod(shape[:ax])) if ax else 1
    if inner < 16 or inner & (inner - 1) or batch < 1:
        return None
    return batch, inner


def clear_plan_cache():
    """Forget every cached plan; each frees its device memory once no caller holds it any more.
    Also releases the blocks the library's helper entry points keep between calls."""
    with _cache_lock:
        _cache.clear()
        _cache_bytes.clear()
    _lib.load().mfft_release_cached()


def engine_dtype(a: np.ndarray):
    """(library dtype, input is real) for an array, following numpy 2.x promotion:
    float32 / complex64 stay single precision, everything else runs in double."""
    if a.dtype == np.complex64:
        return _lib.C64, False
    if a.dtype == np.float32 or a.dtype == np.float16:
        return _lib.C64, True
    if np.iscomplexobj(a):
        return _lib.C128, False
    return _lib.C128, True


def _norm_factor(n: int, norm, real_dtype) -> float:
    """The factor numpy applies for `norm`, formed the way numpy forms it: reciprocal and square root
    in the input's own real precision (numpy/fft/_pocketfft.py, _raw_fft) -- so half- and
    single-precision inputs carry that precision's rounding of 1/n or 1/sqrt(n), as numpy's do."""
    if norm is None or norm == "backward":
        return 1.0
    rdt = np.result_type(real_dtype, 1.0)
    if norm == "ortho":
        return float(np.reciprocal(np.sqrt(n, dtype=rdt)))
    return float(np.reciprocal(n, dtype=rdt))


def _fft_device(a, n, axis, norm):
    """`fft` on a torch CUDA tensor: same semantics, the data never leaves the GPU (torch owns
    the memory and the stream; the transform is libmfft's)."""
    import torch
    if a.dim() == 0:
        raise ValueError("fft requires at least a 1-D array")
    n_have = a.shape[axis]
    n = n_have if n is None else operator.index(n)
    if n < 1:
        raise ValueError(f"Invalid number of FFT data points ({n}) specified.")
    if norm not in (None, "backward", "ortho", "forward"):
        raise ValueError(f'Invalid norm value {norm}; should be "backward", "ortho" or "forward".')
    if a.dtype in (torch.complex64, torch.float32, torch.float16):
        dtype, rdt, cdt = _lib.C64, torch.float32, torch.complex64
    else:
        dtype, rdt, cdt = _lib.C128, torch.float64, torch.complex128
    real_in = not a.is_complex()
    cs = None if real_in else _column_shape(tuple(a.shape), axis, n)
    if cs is not None and a.numel() > 0:
        # a middle / leading axis: transform the strided columns where they lie (no transposes)
        with torch.cuda.device(a.device):
            plan = cached_column_plan(n, cs[1], cs[0], dtype, device=a.device.index)
            if plan is not None:
                x = a.to(cdt).contiguous()
                y = torch.empty_like(x)
                with plan._lock:
                    plan.exec(x, y, stream=torch.cuda.current_stream().cuda_stream)
                if norm in ("ortho", "forward"):
                    y *= _norm_factor(n, norm, np.float32 if dtype == _lib.C64 else np.float64)
                return y
    x = a.movedim(axis, -1)
    if n < n_have:
        x = x[..., :n]
    n_in = x.shape[-1]
    lead = tuple(x.shape[:-1])
    batch = 1
    for d in lead:
        batch *= int(d)
    y = torch.empty(lead + (n,), dtype=cdt, device=a.device)
    if batch == 0:
        return y.movedim(-1, axis)
    x = x.to(rdt if real_in else cdt).contiguous()
    scale = _norm_factor(n, norm, np.float16 if a.dtype == torch.float16 else np.float32 if dtype == _lib.C64 else np.float64)
    with torch.cuda.device(a.device):
        plan = cached_plan(n, n_in, batch, dtype, _lib.REAL_IN if real_in else 0, device=a.device.index)
        with plan._lock:
            plan.exec(x, y, scale=scale, stream=torch.cuda.current_stream().cuda_stream)
    return y.movedim(-1, axis)


def fft(a, n=None, axis=-1, norm=None, out=None):
    """One-dimensional forward DFT along ``axis``, as ``numpy.fft.fft``.  Array-likes go in and
    come out as numpy arrays; a torch CUDA tensor is transformed in place on its device and a
    CUDA tensor comes back (no PCIe traffic)."""
    if out is not None:
        y = fft(a, n=n, axis=axis, norm=norm)
        if hasattr(y, "is_cuda"):                       # device result into a device tensor
            if tuple(out.shape) != tuple(y.shape):
                raise ValueError("output array has wrong shape.")
            out.copy_(y)
            return out
        if not isinstance(out, np.ndarray) or out.shape != y.shape:
            raise ValueError("output array has wrong shape.")
        if out.dtype.kind != "c":
            raise TypeError(f"Cannot cast the transform's {y.dtype} output to {out.dtype}")
        out[...] = y
        return out
    if hasattr(a, "is_cuda") and hasattr(a, "data_ptr"):
        if not a.is_cuda:
            a = a.numpy()
        else:
            return _fft_device(a, n, axis, norm)
    a = np.asarray(a)
    if a.ndim == 0:
        raise ValueError("fft requires at least a 1-D array")
    n_have = a.shape[axis]
    n = n_have if n is None else operator.index(n)     # numpy: a float length is a TypeError
    if n < 1:
        raise ValueError(f"Invalid number of FFT data points ({n}) specified.")
    if norm not in (None, "backward", "ortho", "forward"):
        raise ValueError(f'Invalid norm value {norm}; should be "backward", "ortho" or "forward".')
    dtype, real_in = engine_dtype(a)
    rdt = np.float32 if dtype == _lib.C64 else np.float64
    cdt = np.complex64 if dtype == _lib.C64 else np.complex128
    cs = None if real_in else _column_shape(a.shape, axis, n)
    if cs is not None and a.size > 0:
        plan = cached_column_plan(n, cs[1], cs[0], dtype)
        if plan is not None:
            y = plan.exec_host(np.ascontiguousarray(a, dtype=cdt)).reshape(a.shape)
            if norm in ("ortho", "forward"):
                y *= _norm_factor(n, norm, a.real.dtype)
            return y
    x = np.moveaxis(a, axis, -1)
    if n < n_have:
        x = x[..., :n]                         # numpy crops when n is shorter than the axis
    n_in = x.shape[-1]
    lead = x.shape[:-1]
    batch = int(np.prod(lead)) if lead else 1
    if batch == 0:
        return np.moveaxis(np.zeros(lead + (n,), dtype=cdt), -1, axis)
    x = np.ascontiguousarray(x, dtype=rdt if real_in else cdt).reshape(batch, n_in)
    scale = _norm_factor(n, norm, a.real.dtype)
    plan = cached_plan(n, n_in, batch, dtype, _lib.REAL_IN if real_in else 0)
    y = plan.exec_host(x, scale=scale)
    return np.moveaxis(y.reshape(lead + (n,)), -1, axis)


def fft2(a, norm=None):
    """Two-dimensional forward DFT over the last two axes, as ``numpy.fft.fft2(a)``: row
    transforms, then the column transforms on the strided columns (no transpose pass when the
    shape allows).  numpy arrays and torch CUDA tensors, like `fft`."""
    if not (hasattr(a, "is_cuda") and hasattr(a, "data_ptr")):
        a = np.asarray(a)
        if a.ndim < 2:
            raise ValueError("fft2 requires at least a 2-D array")
        try:
            import torch
        except ImportError:
            torch = None
        if torch is not None and a.size > 0 and a.dtype.kind in "fc" and a.dtype.itemsize >= 4:
            # one trip over the bus: both transforms run on the device copy
            d = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return fft(fft(d, axis=-1, norm=norm), axis=-2, norm=norm).cpu().numpy()
    return fft(fft(a, axis=-1, norm=norm), axis=-2, norm=norm)


def fftfreq(n, d=1.0):
    """Sample frequencies in FFT order: [0, 1, ..., (n-1)//2, -(n//2), ..., -1] / (d*n)."""
    if not isinstance(n, (int, np.integer)):
        raise ValueError("n should be an integer")
    n = int(n)
    k = np.arange(n, dtype=np.int64)
    k[(n - 1) // 2 + 1:] -= n
    return k * (1.0 / (n * d))
