"""ctypes binding of libmfft.so (the C ABI declared in include/mfft.h).

The library is built in-tree by ``make -C fft_amd/csrc`` (or
``__graft_entry__.build()``) into ``fft_amd/lib/libmfft.so``.  There is no CPU
fallback: if the shared object is missing or no GPU is visible, every call
raises -- the product path never computes a transform on the host.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmfft.so")

C64, C128 = 0, 1
HANN, POWER, DEMOD, REAL_IN, STATS, INVERSE = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20

OK, E_INVALID, E_UNSUPPORTED, E_NOMEM, E_HIP, E_NODEVICE = 0, -1, -2, -3, -4, -5

# every symbol include/mfft.h declares: (name, restype, argtypes)
_vp, _i64, _dbl, _int, _u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_uint32
_dblp = ctypes.POINTER(ctypes.c_double)


class Layout(ctypes.Structure):
    """mfft_layout: element j of signal b sits at
    b*batch_stride + (seg ? (j // seg)*seg_stride + j % seg : j); zeros = dense."""
    _fields_ = [("batch_stride", ctypes.c_int64), ("seg", ctypes.c_int64), ("seg_stride", ctypes.c_int64)]


class PeakParams(ctypes.Structure):
    """mfft_peak_params"""
    _fields_ = [("dt", ctypes.c_double), ("dc_guard", ctypes.c_double), ("guard", ctypes.c_double),
                ("include_dc", ctypes.c_int), ("mode", ctypes.c_int), ("want", ctypes.c_int)]


SIGNATURES = {
    "mfft_version": (_int, []),
    "mfft_device_count": (_int, []),
    "mfft_device_memory": (_int, [ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "mfft_release_cached": (_int, []),
    "mfft_strerror": (ctypes.c_char_p, [_int]),
    "mfft_last_error": (ctypes.c_char_p, []),
    "mfft_plan_1d": (_int, [ctypes.POINTER(_vp), _i64, _i64, _i64, _int, _u32]),
    "mfft_plan_1d_layout": (_int, [ctypes.POINTER(_vp), _i64, _i64, _i64, _int, _u32,
                                   ctypes.POINTER(Layout), ctypes.POINTER(Layout)]),
    "mfft_plan_1d_columns": (_int, [ctypes.POINTER(_vp), _i64, _i64, _i64, _int, _i64, _i64]),
    "mfft_plan_1d_columns_ld": (_int, [ctypes.POINTER(_vp), _i64, _i64, _i64, _int, _i64, _i64, _i64]),
    "mfft_plan_destroy": (_int, [_vp]),
    "mfft_plan_workspace_bytes": (_int, [_vp, ctypes.POINTER(_i64)]),
    "mfft_plan_launches": (_int, [_vp, ctypes.POINTER(_int)]),
    "mfft_exec": (_int, [_vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp]),
    "mfft_exec_host": (_int, [_vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp]),
    "mfft_lsq_normal_host": (_int, [_vp, _vp, _vp, _i64, _dbl, _vp, _int, _vp, _vp, _vp]),
    "mfft_lines_host": (_int, [_vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, ctypes.POINTER(PeakParams), _int, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp]),
    "mfft_lines_blocks_host": (_int, [_vp, _vp, _vp, _vp, _int, _dbl, _dbl, _vp, ctypes.POINTER(PeakParams), _int, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp]),
    "mfft_lines_extract_host": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _dbl, _dbl, _dbl, _dbl, _vp, ctypes.POINTER(PeakParams),
                                        _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_dbl)]),
    "mfft_lines_gram_host": (_int, [_vp, _vp, _int, _dbl, _vp]),
    "mfft_lsq_normal_static": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "mfft_local_maxima": (_int, [_vp, _i64, _int, _vp, _i64, _vp, _vp]),
    "mfft_exec_host_peaks": (_int, [_vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp]),
    "mfft_return_amplitude_host": (_int, [_vp, _i64, _vp, _dbl, _dbl, _vp, _vp, _vp, _int, _dbl]),
    "mfft_return_amplitude": (_int, [_vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _int, _dbl, _vp]),
    "mfft_normalize_by_max": (_int, [_vp, _vp, _i64, _vp, _int, _vp]),
    "mfft_transpose": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "mfft_host_alloc": (_int, [ctypes.POINTER(_vp), _i64]),
    "mfft_host_free": (_int, [_vp]),
    "mfft_swap_axes": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "mfft_swap_axes_strided": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
    "mfft_twiddle_tile": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _int, _vp]),
    "mfft_comm_unique_id": (_int, [_vp]),
    "mfft_comm_init_rank": (_int, [ctypes.POINTER(_vp), _int, _int, _vp]),
    "mfft_comm_init_local": (_int, [ctypes.POINTER(_vp), _int]),
    "mfft_comm_info": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "mfft_comm_destroy": (_int, [_vp]),
    "mfft_comm_exchange": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "mfft_plan_2d_dist": (_int, [ctypes.POINTER(_vp), _vp, _i64, _i64, _int, _int, _int, _int]),
    "mfft_plan_1d_dist": (_int, [ctypes.POINTER(_vp), _vp, _i64, _i64, _int, _int, _int, _int]),
    "mfft_dist_exec": (_int, [_vp, _vp, _vp, _vp]),
    "mfft_dist_plan_info": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_i64)]),
    "mfft_dist_plan_destroy": (_int, [_vp]),
}

_lib = None


class MfftError(RuntimeError):
    """A libmfft call failed for a reason other than a bad argument."""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (soname
    libamdhip64.so.7) and finds it by file name; libmfft.so asks for the soname and would
    otherwise bring in /opt/rocm's copy.  Two runtimes in one process cannot both own the
    device ("No HIP GPUs are available" from whichever comes second), so when torch is
    installed its runtime is loaded first -- without importing torch -- and libmfft.so binds
    to it through the shared soname.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                      # torch already loaded its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for base in spec.submodule_search_locations:
        libdir = os.path.join(base, "lib")
        hip = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(hip):
            for dep in ("libhsa-runtime64.so",):
                p = os.path.join(libdir, dep)
                if os.path.exists(p):
                    ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
            ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
            return


def load():
    """Load libmfft.so (once) and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfftError(
                f"{LIB_PATH} is missing: build it with `make -C fft_amd/csrc` "
                "(there is no CPU fallback for the spectral path)")
        _share_hip_runtime_with_torch()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int):
    """Translate a return code: bad arguments -> ValueError, the rest -> MfftError."""
    if rc == OK:
        return
    lib = load()
    detail = lib.mfft_last_error().decode("utf-8", "replace")
    what = lib.mfft_strerror(rc).decode()
    if rc == E_INVALID:
        raise ValueError(detail or what)
    raise MfftError(f"{what}: {detail}" if detail else what)


def device_count() -> int:
    n = load().mfft_device_count()
    if n < 0:
        check(n)
    return n
