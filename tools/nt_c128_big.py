#!/usr/bin/env python3
"""Cache policy of the passes of one large complex128 plan, alternated inside one process (MFFT_TILE_NT_PASS is
read when a plan is built):   python tools/nt_c128_big.py [LOG2N[,LOG2N...] [POLICY ...]]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, fft_amd
torch.cuda.set_device(0)
for lg in ([int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (27, 26)):
    n = 1 << lg
    x = torch.empty(n, dtype=torch.complex128, device="cuda")
    for lo in range(0, n, 1 << 27): torch.view_as_real(x)[lo:lo + (1 << 27)].normal_()
    y = torch.empty_like(x); s = torch.cuda.current_stream().cuda_stream
    res = {}
    for rnd in range(3):
        for pol in (sys.argv[2:] or ("default", "3,3,1", "3,3,0", "1,1,1", "0,0,0")):
            if pol == "default": os.environ.pop("MFFT_TILE_NT_PASS", None)
            else: os.environ["MFFT_TILE_NT_PASS"] = pol
            p = fft_amd.Plan(n, dtype=fft_amd.C128)
            for _ in range(2): p.exec(x, y, stream=s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): p.exec(x, y, stream=s)
            e1.record(); e1.synchronize()
            res.setdefault(pol, []).append(e0.elapsed_time(e1) / 8 * 1e3); p.destroy()
    print(f"2^{lg} c128: " + " | ".join(f"{k} " + "/".join(f"{v:.0f}" for v in vs) for k, vs in res.items()), flush=True)
