#!/usr/bin/env python3
"""Run one plan a few times (for rocprofv3 --kernel-trace --stats):
    python tools/one_plan.py N BATCH [DTYPE] [FLAGS] [LIBTAG] [LAUNCHES]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import _lib as L
n, batch = int(sys.argv[1]), int(sys.argv[2])
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 0
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tag = sys.argv[5] if len(sys.argv) > 5 else "default"
lib = ctypes.CDLL(os.path.join(ROOT, "fft_amd", "lib", "libmfft.so" if tag == "default" else f"libmfft_{tag}.so"))
for fn, (res, args) in L.SIGNATURES.items():
    getattr(lib, fn).restype = res; getattr(lib, fn).argtypes = args
torch.cuda.set_device(0)
rdt = torch.float32 if dtype == 0 else torch.float64
x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=rdt, device="cuda"))
y = torch.empty((batch, n), dtype=x.dtype if not flags & 2 else rdt, device="cuda")
h = ctypes.c_void_p()
assert lib.mfft_plan_1d(ctypes.byref(h), n, n, batch, dtype, flags) == 0, lib.mfft_last_error()
for _ in range(int(sys.argv[6]) if len(sys.argv) > 6 else 30):
    assert lib.mfft_exec(h, x.data_ptr(), y.data_ptr(), 1.0, 0.0, 0.0, 1.0, None, None) == 0
torch.cuda.synchronize()
