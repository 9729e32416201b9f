import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import fft_amd as fa
rng = np.random.default_rng(5)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300 + np.linalg.norm(a) * 0 + (1.0 if not np.any(b) else 0.0)))
worst = 0
for split, lg, batch, n_in, power in [("9,9", 18, 3, 1 << 18, False), ("9,9", 18, 24, 200001, True), ("9,9", 18, 17, 131072, False),
                                     ("9,9", 18, 2, 2, False), ("9,9", 18, 2, 3, True), ("9,9", 18, 5, 513, False), ("9,9", 18, 5, (1 << 18) - 1, True),
                                     ("9,9,4", 22, 1, 3000001, True), ("9,4,9", 22, 2, 1 << 21, False)]:
    os.environ["MFFT_SPLIT"] = split
    n = 1 << lg
    x = (rng.standard_normal((batch, n_in)) + 1j * rng.standard_normal((batch, n_in)))
    flags = fa.HANN | (fa.POWER if power else 0)
    with fa.Plan(n, n_in=n_in, batch=batch, dtype=1, flags=flags) as p:
        y = p.exec_host(x)
    ref = np.fft.fft(x * np.hanning(n_in), n, axis=-1)
    if power: ref = np.abs(ref) ** 2
    e = rel(y, ref); worst = max(worst, e)
    print(split, lg, batch, n_in, power, "%.2e" % e, flush=True)
    assert e < 1e-13
print("ok worst %.2e" % worst)
# timing A/B: evaluated window vs table, alternated in one process
os.environ["MFFT_SPLIT"] = "9,9"
for lg, batch in ((18, 256), (17, 512)):
    os.environ["MFFT_SPLIT"] = "9,9" if lg == 18 else "9,8"
    n = 1 << lg
    x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float64, device="cuda"))
    S = torch.empty((batch, n), dtype=torch.float64, device="cuda")
    plans = {}
    for k in ("eval", "table"):                    # the variable is read when a plan is built
        if k == "table": os.environ["MFFT_TABLE_WINDOW"] = "1"
        else: os.environ.pop("MFFT_TABLE_WINDOW", None)
        plans[k] = fa.Plan(n, batch=batch, dtype=1, flags=fa.HANN | fa.POWER)
    os.environ.pop("MFFT_TABLE_WINDOW", None)
    s = torch.cuda.current_stream().cuda_stream
    res = {"eval": [], "table": []}
    for r in range(7):
        for k, p in plans.items():
            for _ in range(3): p.exec(x, S, stream=s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): p.exec(x, S, stream=s)
            e1.record(); e1.synchronize()
            res[k].append(e0.elapsed_time(e1) / 10)
    print("2^%d x %d hann+power c128: evaluated %.3f ms, table %.3f ms" % (lg, batch, sorted(res["eval"])[3], sorted(res["table"])[3]), flush=True)
    for p in plans.values(): p.destroy()
