"""fft_amd -- MI355X (gfx950) engine for the spectral hot path of xavlolo/FFT.

Layers (bottom up):
  csrc/        hand-written HIP kernels + the C ABI of include/mfft.h -> lib/libmfft.so
  _lib.py      ctypes binding of that ABI
  engine.py    Plan objects over device or host buffers
  npfft.py     numpy-compatible fft / fftfreq served by the GPU
  spectral.py  drop-ins for the reference's fft_from_timeseries / fft_extract_E_p
  dist.py      one-process-per-GPU sharding: batches, 2-D pencils, large 1-D (torch.distributed / RCCL)
"""
from . import _lib
from .engine import Plan, ColumnPlan
from ._lib import C64, C128, HANN, POWER, DEMOD, REAL_IN, STATS, INVERSE, MfftError, device_count
from .npfft import fft, fft2, fftfreq, clear_plan_cache
from .spectral import (fft_from_timeseries, fft_from_timeseries_batch, fft_from_timeseries_blocks, fft_extract_E_p, power_spectrum, return_amplitude,
                       return_amplitude_block_general, Heff_block_general)

__all__ = ["Plan", "ColumnPlan", "C64", "C128", "HANN", "POWER", "DEMOD", "REAL_IN", "STATS", "INVERSE",
           "MfftError", "device_count", "fft", "fft2", "fftfreq", "clear_plan_cache",
           "fft_from_timeseries", "fft_from_timeseries_batch", "fft_from_timeseries_blocks", "fft_extract_E_p", "power_spectrum", "return_amplitude",
           "return_amplitude_block_general", "Heff_block_general"]
