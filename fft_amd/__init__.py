"""fft_amd -- MI355X (gfx950) engine for the spectral hot path of xavlolo/FFT.

Layers (bottom up):
  csrc/        hand-written HIP kernels + the C ABI of include/mfft.h -> lib/libmfft.so
  _lib.py      ctypes binding of that ABI
  engine.py    Plan objects over device or host buffers
"""
from . import _lib
from .engine import Plan
from ._lib import C64, C128, HANN, POWER, DEMOD, REAL_IN, STATS, INVERSE, MfftError, device_count

__all__ = ["Plan", "C64", "C128", "HANN", "POWER", "DEMOD", "REAL_IN", "STATS", "INVERSE",
           "MfftError", "device_count"]
