import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fft_amd
fft_amd._lib.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "fft_amd/lib/libmfft_r10.so")
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
for k, batch in ((19, 1), (19, 16), (20, 8)):
    n = 1 << k
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float64, device="cuda")); y = torch.empty_like(x)
    ref = torch.fft.fft(x[0])
    for tag in ["default", f"10,{k-10}", f"{k-10},10"] + (["10,10"] if k == 20 else []):
        if tag == "default": os.environ.pop("MFFT_SPLIT", None)
        else: os.environ["MFFT_SPLIT"] = tag
        try:
            with fft_amd.Plan(n, batch=batch, dtype=1) as p:
                for _ in range(5): p.exec(x, y, stream=stream)
                torch.cuda.synchronize()
                err = (torch.linalg.vector_norm(y[0] - ref) / torch.linalg.vector_norm(ref)).item()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): p.exec(x, y, stream=stream)
                    e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
                print(f"2^{k} x {batch} c128 split {tag} launches {p.launches}: {best:7.1f} us err {err:.1e}", flush=True)
        except Exception as exc:
            print(f"2^{k} split {tag}: {str(exc)[:80]}", flush=True)
os.environ.pop("MFFT_SPLIT", None)
