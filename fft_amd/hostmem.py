"""Page-locked host memory for the estimators' spectra.

The estimators return full-length spectra (final2.py:146, 188: 2 MB at the app's defaults, one per block).  Into a
fresh numpy array the download goes through a staging buffer and a second copy that faults every page in; a block
from ``mfft_host_alloc`` is written by the device directly.  ``empty`` hands out numpy arrays over such blocks and
takes a block back when the last view of it has been collected (three blocks per call: 0.57 -> 0.46 ms).  Past the
pool's limits, for small arrays and without the library callers get plain ``np.empty`` arrays.
"""
import ctypes
import threading
import weakref

import numpy as np

from . import _lib

MIN_BYTES = 256 << 10            # smaller arrays are cheaper from numpy
MAX_HELD = 1 << 30               # page-locked bytes in the hands of callers plus the free lists
MAX_FREE_PER_CLASS = 4

_lock = threading.Lock()
_free = {}                       # size class -> [address]
_held = 0


def _release(addr: int) -> None:
    try:
        _lib.load().mfft_host_free(ctypes.c_void_p(addr))
    except Exception:
        pass


def _give_back(addr: int, size: int) -> None:
    global _held
    with _lock:
        lst = _free.setdefault(size, [])
        if len(lst) < MAX_FREE_PER_CLASS:
            lst.append(addr)
            return
        _held -= size
    _release(addr)


def empty(shape, dtype) -> np.ndarray:
    """np.empty(shape, dtype) over a recycled page-locked block where that pays."""
    global _held
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    nbytes = count * dtype.itemsize
    if nbytes < MIN_BYTES:
        return np.empty(shape, dtype)
    size = MIN_BYTES
    while size < nbytes:
        size <<= 1
    addr = None
    with _lock:
        lst = _free.get(size)
        if lst:
            addr = lst.pop()
        elif _held + size > MAX_HELD:
            return np.empty(shape, dtype)
        else:
            _held += size
    if addr is None:
        p = ctypes.c_void_p()
        try:
            _lib.check(_lib.load().mfft_host_alloc(ctypes.byref(p), size))
            addr = int(p.value)
        except Exception:
            with _lock:
                _held -= size
            return np.empty(shape, dtype)
    buf = (ctypes.c_char * nbytes).from_address(addr)
    fin = weakref.finalize(buf, _give_back, addr, size)
    fin.atexit = False                                  # the runtime may be gone by then; the process is ending anyway
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def release_free_blocks() -> None:
    """Return the free list to the runtime (blocks still referenced by arrays stay)."""
    global _held
    with _lock:
        blocks = [(a, s) for s, lst in _free.items() for a in lst]
        _free.clear()
        _held -= sum(s for _, s in blocks)
    for a, _ in blocks:
        _release(a)
