import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fft_amd
torch.cuda.set_device(0)
rng = np.random.default_rng(1)
for lg, splits in ((20, ["10,10"]), (21, ["11,10", "10,11"]), (22, ["11,11"]), (26, ["10,8,8", "8,10,8", "8,8,10", "11,8,7", "7,8,11", "8,11,7"])):
    n = 1 << lg
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = np.fft.fft(x)
    d = torch.from_numpy(x).cuda()
    for sp in splits:
        os.environ["MFFT_SPLIT"] = sp
        p = fft_amd.Plan(n, dtype=fft_amd.C128)
        y = torch.empty_like(d)
        p.exec(d, y); torch.cuda.synchronize()
        err = np.linalg.norm(y.cpu().numpy() - ref) / np.linalg.norm(ref)
        print(lg, sp, p.launches, err, flush=True)
        assert err < 1e-12
        # power + stats and windowed first pass through the same radices
        p.destroy()
        p = fft_amd.Plan(n, n_in=n - 1000, dtype=fft_amd.C128, flags=fft_amd.HANN | fft_amd.POWER)
        S = torch.empty(n, dtype=torch.float64, device="cuda")
        p.exec(d, S, scale=0.5); torch.cuda.synchronize()
        r2 = np.abs(0.5 * np.fft.fft(x[:n - 1000] * np.hanning(n - 1000), n=n)) ** 2
        err = np.linalg.norm(S.cpu().numpy() - r2) / np.linalg.norm(r2)
        print(lg, sp, "hann+power", err, flush=True)
        assert err < 1e-12
        p.destroy()
os.environ.pop("MFFT_SPLIT")
print("ok")
