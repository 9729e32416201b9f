#!/usr/bin/env python3
"""Headline benchmark: BASELINE config 2 -- batched 1-D complex64 FFT, N = 1024 x
65536 signals, fused Hann window + power spectrum, per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (starts N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` without a launcher's WORLD_SIZE in the environment starts the N ranks as fresh child
processes (torch.distributed.run on 127.0.0.1) before this process makes any GPU call, relays
rank 0's line and returns the children's status; `--gpus` that disagrees with WORLD_SIZE is refused.

One step = one pass of the hot path over one batch (already resident in HBM).
With N > 1 every rank transforms its own batch of 65536 signals (independent
signals shard with no data-path collective): weak scaling, value = whole-job
GFLOP/s.  Rank 0 prints one JSON line.  `roofline` prices the dominant kernel
(the only kernel of this workload) by its algorithmic bytes -- 8 B read + 4 B
written per point -- over its measured average launch time; `cpu_baseline` is the
reference's own numpy chain (np.hanning, multiply, np.fft.fft, np.abs()**2) timed
on the host, single thread, for about ten seconds of CPU work over the same signals
(`cpu_baseline_multicore`: the same chain through scipy.fft's worker pool).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
N_POINTS = 1024
BATCH = 65536
DT = 0.01


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks of the job, one per GPU (default: the launcher's WORLD_SIZE, else 1); "
                         "N > 1 without a launcher starts N ranks")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle", type=float, default=1.5,
                    help="seconds of untimed launches before the warmup steps: the GPU needs ~0.1-0.3 s of "
                         "sustained load after idle before its clocks settle (first launches run 135 us, "
                         "then up to 185 us, then settle)")
    ap.add_argument("--n", type=int, default=N_POINTS, help="transform length (default: the headline 1024)")
    ap.add_argument("--batch", type=int, default=BATCH, help="signals per GPU (default: the headline 65536)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short timings of BASELINE configs 3-5 (one GPU) appended as `other_configs`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--no-dist-configs", action="store_true",
                    help="with several ranks: skip the sharded 2-D / long 1-D transforms (BASELINE configs 4, 5) "
                         "that follow the batch-sharded headline")
    ap.add_argument("--dist-configs", action="store_true",
                    help="run the sharded-transform section even with one rank (both routes; a plumbing check)")
    ap.add_argument("--no-dist-c-abi", action="store_true",
                    help="with several ranks over RCCL: skip the second run of configs 4 and 5 through the C ABI's own "
                         "distributed plans (a watchdog abandons that leg after --c-abi-timeout seconds and still "
                         "prints the line)")
    ap.add_argument("--c-abi-timeout", type=float, default=240.0)
    ap.add_argument("--dist-timeout", type=float, default=240.0,
                    help="watchdog of the sharded-transform section over torch.distributed (seconds)")
    ap.add_argument("--dist-n2d", type=int, default=16384, help="side of the sharded 2-D transform (config 4)")
    ap.add_argument("--dist-log2n1d", type=int, default=0,
                    help="log2 length of the sharded 1-D transform (config 5); 0 = 31 on 8 ranks, else 28 + log2(ranks)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="all ranks share cuda:0 (use with --backend gloo on a one-GPU box; not a measurement)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="every rank joins the process group, the ranks count themselves with one all-reduce and "
                         "rank 0 prints a line without a measurement; nothing touches a GPU (checks the launch path)")
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port when this process starts the ranks itself (0 = a free one)")
    return ap.parse_args(argv)


EXIT_REFUSED = 2        # --gpus disagrees with the launcher's WORLD_SIZE
EXIT_ABANDONED = 3      # a sharded leg did not finish in time: the line is printed, the run was cut short


def resolve_world(args, env):
    """(world size, must this process start the ranks itself?) from --gpus and the launcher's environment.
    A launcher (torch.distributed.run) sets WORLD_SIZE; --gpus must agree with it when both are given."""
    env_world = env.get("WORLD_SIZE")
    if env_world is not None:
        env_world = int(env_world)
        if args.gpus is not None and args.gpus != env_world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world} of the launcher; "
                             "refusing to report a line for a job of another size")
        return env_world, False
    gpus = 1 if args.gpus is None else args.gpus
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    return gpus, gpus > 1


def launch_ranks(gpus, argv, port=0):
    """Start `gpus` fresh ranks of this script under torch.distributed.run (rendezvous on 127.0.0.1) and return
    their status.  Called before this process has imported torch or made any GPU call; the children inherit
    stdout, so rank 0's single JSON line is this command's output."""
    import socket
    import subprocess
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    # the ranks' stdout is relayed: JSON lines to stdout, anything else a library prints there (gloo's
    # connection notes, RCCL debug output) to stderr, so that stdout carries the one line and nothing else
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    for raw in proc.stdout:
        text = raw.decode("utf-8", "replace")
        if text.lstrip().startswith("{"):
            sys.stdout.write(text)
            sys.stdout.flush()
        else:
            sys.stderr.write(text)
    return proc.wait()


def rendezvous_only(args):
    """The launch path without a GPU: join the group, count the ranks, rank 0 prints a line."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    counted = 1
    if world > 1:
        dist.init_process_group(args.backend)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        counted = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "FFT GFLOP/s (5N·log2N) + achieved HBM GB/s vs roofline", "value": None,
                          "unit": "GFLOP/s", "n_gpus": counted, "steps": args.steps, "warmup": args.warmup,
                          "rendezvous_only": True, "ranks_counted": counted, "backend": args.backend}), flush=True)
    return 0


def fenced(line, key, seconds, fn, rank, reporter):
    """Run one sharded leg under a watchdog.  A collective that never returns must not cost the job its
    headline: when the leg has not finished after `seconds`, rank 0 hands the line with what has been measured
    to the reporter and every rank leaves with EXIT_ABANDONED, so the launcher and whoever reads the exit
    status see that the run was cut short."""
    import threading

    def abandon():
        if rank == 0:
            line[key] = {"error": f"abandoned after {seconds:.0f} s"}
            reporter.update(line)
            reporter.close()
        else:
            time.sleep(2.0)            # the launcher ends the job when the first rank leaves: let rank 0 report
        os._exit(EXIT_ABANDONED)
    if rank == 0:
        line[key] = {"error": "the process ended inside this leg"}    # what is printed if it does
        reporter.update(line)
    timer = threading.Timer(seconds, abandon)
    timer.daemon = True
    timer.start()
    try:
        res = fn()
    except Exception as exc:                          # noqa: BLE001 - a side measurement
        res = {"error": repr(exc)[:300]}
    timer.cancel()
    if rank == 0:
        line[key] = res


def synth(batch, n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    x = np.empty((batch, n), dtype=np.complex64)
    chunk = 8192
    for lo in range(0, batch, chunk):
        hi = min(batch, lo + chunk)
        x[lo:hi].real = rng.standard_normal((hi - lo, n), dtype=np.float32)
        x[lo:hi].imag = rng.standard_normal((hi - lo, n), dtype=np.float32)
    return x


def cpu_baseline(x_host, n, budget_s=10.0):
    """The reference's numpy chain over the same batch, 1 thread: whole passes over all the
    signals (in slabs of 4096 rows) until about `budget_s` seconds of CPU work are done."""
    import numpy as np
    from oracle import spectral_oracle as so
    rows = len(x_host)
    slab = 4096
    done, passes = 0, 0
    t0 = time.perf_counter()
    while True:
        for lo in range(0, rows, slab):
            so.reference_numpy_chain(x_host[lo:lo + slab])
            done += min(slab, rows - lo)
        passes += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    elapsed = time.perf_counter() - t0
    flops = 5.0 * n * math.log2(n) * done
    return {"value": flops / elapsed / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"{passes} passes over all {rows} signals (N={n}, complex64) = {elapsed:.1f} s of CPU work; "
                      "np.hanning * x -> np.fft.fft -> np.abs()**2 as final2.py:139-145; "
                      f"numpy {np.__version__} pocketfft is single-threaded"}


def cpu_multicore(x_host, n, budget_s=5.0):
    """Best-effort multicore CPU figure beside it: the same chain with scipy.fft's worker
    pool (the reference itself never uses more than one thread)."""
    import numpy as np
    try:
        import scipy
        import scipy.fft as sfft
    except ImportError:
        return None
    cores = max(1, min(16, os.cpu_count() or 1))      # the job's CPU share on a one-GPU box
    win = np.hanning(n).astype(np.float32)
    rows = len(x_host)
    done, passes = 0, 0
    t0 = time.perf_counter()
    while True:
        F = sfft.fft(x_host * win, axis=-1, workers=cores)
        S = F.real * F.real + F.imag * F.imag
        del F, S
        done += rows
        passes += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    elapsed = time.perf_counter() - t0
    flops = 5.0 * n * math.log2(n) * done
    return {"value": flops / elapsed / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{passes} passes over all {rows} signals = {elapsed:.1f} s; window * x -> "
                      f"scipy.fft.fft(workers={cores}) -> re^2+im^2, scipy {scipy.__version__}"}


def _direct_bins_1d(x, ks, chunk=1 << 26):
    """X[k] = sum_j x[j] exp(-2 pi i j k / n) for a few bins, evaluated directly in float64 on the device
    (exact integer reduction of j*k mod n): the checker of the long transforms' sampled bins."""
    import torch
    n = x.numel()
    out = []
    for k in ks:
        acc = torch.zeros((), dtype=torch.complex128, device=x.device)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            j = torch.arange(lo, hi, dtype=torch.int64, device=x.device)
            ang = (-2.0 * math.pi) * (((j * int(k)) % n).to(torch.float64) / n)
            acc = acc + torch.sum(x[lo:hi].to(torch.complex128) * torch.complex(torch.cos(ang), torch.sin(ang)))
            del j, ang
        out.append(acc)
    return torch.stack(out)


def _energy(t, chunk=1 << 27):
    import torch
    r = torch.view_as_real(t).reshape(-1)
    return sum(float(torch.sum(r[lo:lo + chunk].double() ** 2).item()) for lo in range(0, r.numel(), chunk))


def _check_1d(x, y, ks):
    """max |y[k] - X[k]| over the sampled bins, relative to the root-mean-square bin magnitude"""
    import torch
    got = y[torch.tensor(ks, device=y.device)].to(torch.complex128)
    want = _direct_bins_1d(x, ks)
    return float(torch.max(torch.abs(got - want)).item()) / math.sqrt(_energy(x))


def secondary_configs(with_cpu):
    """Short single-GPU timings of the other BASELINE configurations (parity-test cases, not the
    headline): config 3 (2^24 complex64), config 4 (16384^2 complex64 through the slab
    decomposition, one rank) and config 5 (2^31 complex128 through the four-step decomposition,
    one rank, and as one plan).  Each entry: ms per transform, GFLOP/s (5 N log2 N) and algorithmic GB/s (one read +
    one write of the data); failures are recorded, never raised."""
    import numpy as np
    import torch
    import fft_amd
    from fft_amd import dist as fd
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
        return best

    def entry(t, points, log2n, esz, **extra):
        d = {"ms": t * 1e3, "gflops": 5.0 * points * log2n / t / 1e9, "gbs_algorithmic": 2.0 * esz * points / t / 1e9,
             "roofline_frac": 2.0 * esz * points / t / 1e9 / HBM_PEAK_GBS}
        d.update(extra)
        return d

    stream = torch.cuda.current_stream().cuda_stream
    try:
        n = 1 << 24
        x = torch.view_as_complex(torch.randn((n, 2), dtype=torch.float32, device="cuda"))
        y = torch.empty_like(x)
        plan = fft_amd.Plan(n, dtype=fft_amd.C64)
        t = timed(lambda: plan.exec(x, y, stream=stream), 30)
        extra = {"launches": plan.launches,
                 "verified_max_err_over_rms": _check_1d(x, y, [0, 1, 255, 65536, n // 2, n - 1, 12345678 % n]), "verified_bins": 7}
        if with_cpu:
            xh = x.cpu().numpy()
            t0 = time.perf_counter()
            np.fft.fft(xh)
            extra["cpu_numpy_ms_1core"] = (time.perf_counter() - t0) * 1e3
        out["config3_1d_2pow24_c64"] = entry(t, n, 24, 8, **extra)
        plan.destroy()
        del x, y
    except Exception as exc:                                  # noqa: BLE001 - a side measurement must not sink the headline
        out["config3_1d_2pow24_c64"] = {"error": repr(exc)[:200]}
    try:
        # the reference's own call shapes next to the plain transform (parity-test cases with a clock on them): a real
        # signal (np.fft.fft of real f_t, final2.py:142-143) above one LDS residency -- half-length plan, split fused into
        # its last pass -- and a Hann-windowed, zero-padded complex64 batch (final2.py:139-143) against the plain one
        routes = {}
        n, batch = 1 << 22, 4
        xr = torch.randn((batch, n), dtype=torch.float64, device="cuda")
        xc = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float64, device="cuda"))
        y = torch.empty((batch, n), dtype=torch.complex128, device="cuda")
        with fft_amd.Plan(n, batch=batch, dtype=fft_amd.C128, flags=fft_amd.REAL_IN) as pr, fft_amd.Plan(n, batch=batch, dtype=fft_amd.C128) as pc:
            t_r = timed(lambda: pr.exec(xr, y, stream=stream), 10)
            ks = [0, 1, 255, n // 4 + 3, n // 2 - 1, n // 2, n // 2 + 1, n - 1]
            err = _check_1d(xr[1].to(torch.complex128), y[1], ks)
            t_c = timed(lambda: pc.exec(xc, y, stream=stream), 10)
        routes["real_input_2pow22_x4_f64"] = {"ms": t_r * 1e3, "ms_complex_input_same_length": t_c * 1e3, "ratio": t_r / t_c,
                                              "verified_max_err_over_rms": err, "verified_bins": len(ks)}
        del xr, xc, y
        n, n_in, batch = 1 << 18, 200000, 256          # (an odd count, whose signals start off the 16-byte grid, costs ~10 %)
        x = torch.view_as_complex(torch.randn((batch, n, 2), dtype=torch.float32, device="cuda"))
        y = torch.empty((batch, n), dtype=torch.complex64, device="cuda")
        xin = x[:, :n_in].contiguous()
        with fft_amd.Plan(n, n_in=n_in, batch=batch, dtype=fft_amd.C64, flags=fft_amd.HANN) as pw, fft_amd.Plan(n, batch=batch, dtype=fft_amd.C64) as pp:
            t_w = timed(lambda: pw.exec(xin, y, stream=stream), 10)
            xw = torch.zeros(n, dtype=torch.complex128, device="cuda")
            xw[:n_in] = xin[3].to(torch.complex128) * torch.from_numpy(np.hanning(n_in)).cuda()
            err = _check_1d(xw, y[3], ks=[0, 1, 777, n // 3, n // 2, n - 2, n - 1])
            t_p = timed(lambda: pp.exec(x, y, stream=stream), 10)
        routes["hann_zero_padded_2pow18_x256_c64"] = {"ms": t_w * 1e3, "ms_plain_full_length": t_p * 1e3, "ratio": t_w / t_p,
                                                     "n_in": n_in, "verified_max_err_over_rms": err, "verified_bins": 7}
        del x, y, xin
        out["reference_call_shapes"] = routes
    except Exception as exc:                                  # noqa: BLE001
        out["reference_call_shapes"] = {"error": repr(exc)[:200]}
    torch.cuda.empty_cache()
    comm, ops = fd.Comm(), fd.HipOps()
    try:
        n = 16384
        a = torch.view_as_complex(torch.randn((n * n, 2), dtype=torch.float32, device="cuda"))
        t = timed(lambda: fd.fft2_slab(a, n, n, comm, ops, restore_layout=True), 3)
        f = fd.fft2_slab(a, n, n, comm, ops, restore_layout=True).view(n, n)
        a2 = a.view(n, n)
        j = torch.arange(n, dtype=torch.int64, device="cuda")
        worst = 0.0
        bins = [(0, 0), (1, 0), (0, 1), (127, 128), (8191, 8192), (16383, 16383), (12345, 6789)]
        for k0, k1 in bins:
            w1 = torch.exp(-2j * math.pi * ((j * k1) % n).double() / n)
            w0 = torch.exp(-2j * math.pi * ((j * k0) % n).double() / n)
            acc = torch.zeros((), dtype=torch.complex128, device="cuda")
            for lo in range(0, n, 2048):
                acc = acc + torch.dot(w0[lo:lo + 2048], a2[lo:lo + 2048].to(torch.complex128) @ w1)
            worst = max(worst, abs(complex(f[k0, k1].item()) - complex(acc.item())))
        out["config4_2d_16384x16384_c64_one_gpu"] = entry(t, n * n, 28, 8, local_passes=3,
                                                         verified_max_err_over_rms=worst / math.sqrt(_energy(a)),
                                                         verified_bins=len(bins))
        del f, a2
        del a
    except Exception as exc:                                  # noqa: BLE001
        out["config4_2d_16384x16384_c64_one_gpu"] = {"error": repr(exc)[:200]}
    ops.close()
    torch.cuda.empty_cache()
    try:
        free, _ = torch.cuda.mem_get_info()
        top = 31 if free >= 180 * (1 << 30) else 28
        # the shorter lengths are what 2- and 4-GPU jobs run (2^(28 + log2 ranks)): their one-GPU figures
        for lg in [q for q in (28, 29, 30) if q < top] + [top]:
            n1, n2 = fd.fourstep_factors(lg)
            x = torch.empty(n1 * n2, dtype=torch.complex128, device="cuda")
            xr = torch.view_as_real(x)
            for lo in range(0, n1 * n2, 1 << 27):
                xr[lo:lo + (1 << 27)].normal_()
            t = timed(lambda: fd.fft1d_fourstep(x, n1, n2, comm, ops, natural_order=True), 1)
            nn = n1 * n2
            ks = [0, 1, n1 - 1, n1, nn // 2, nn - 1, 1234567891 % nn]
            yv = fd.fft1d_fourstep(x, n1, n2, comm, ops, natural_order=True)
            err = _check_1d(x, yv, ks)
            del yv
            out[f"config5_1d_2pow{lg}_c128_one_gpu"] = entry(t, n1 * n2, lg, 16, local_passes=5,
                                                              route="four-step decomposition, one rank",
                                                              verified_max_err_over_rms=err, verified_bins=len(ks))
            ops.close()
            torch.cuda.empty_cache()
            if True:
                # the same transform as ONE plan (global passes, natural order, no re-layout pass): what a
                # single GPU should run; the four-step route above is the one-rank case of the multi-GPU path
                y = torch.empty_like(x)
                plan = fft_amd.Plan(n1 * n2, dtype=fft_amd.C128)
                t = timed(lambda: plan.exec(x, y, stream=stream), 1)
                out[f"config5_1d_2pow{lg}_c128_one_gpu_single_plan"] = entry(
                    t, n1 * n2, lg, 16, launches=plan.launches,
                    verified_max_err_over_rms=_check_1d(x, y, [0, 1, 255, 65536, nn // 2, nn - 1, 987654321 % nn]), verified_bins=7)
                plan.destroy()
                del y
            del x, xr
            torch.cuda.empty_cache()
    except Exception as exc:                                  # noqa: BLE001
        out["config5_1d_c128_one_gpu"] = {"error": repr(exc)[:200]}
    ops.close()
    torch.cuda.empty_cache()
    return out


def dist_configs(args, world, rank, c_abi=False):
    """BASELINE configs 4 and 5 over the ranks of this job -- the sharded transforms with real
    exchange steps (fft_amd/dist.py: row/column slabs with one or two all-to-alls; the four-step
    split of one long signal with two or three).  Every entry: ms per transform (max over ranks,
    barrier-bracketed, best of a few brackets), whole-job GFLOP/s, the duration of each exchange
    on this rank (events around each collective / chunked exchange), the ratio to the one-GPU
    figure of the same transform when profiles/one_gpu_reference.json has one, and a correctness
    check of sampled bins against the DFT evaluated directly in float64 over all ranks' data.
    c_abi: run the transforms through the C ABI's own distributed plans (RCCL called from inside
    libmfft.so, every exchange chunked and overlapped) instead of the torch.distributed route."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from fft_amd import dist as fd

    comm, ops = fd.Comm(), fd.HipOps()
    P = comm.world
    xcomm = fd.CComm.from_torch() if c_abi else comm          # the communicator the transforms use
    out = {"world_size": P, "backend": comm.backend, "transport": "RCCL" if comm.backend == "nccl" else comm.backend,
           "route": "C ABI (mfft_plan_2d_dist / mfft_plan_1d_dist, RCCL send/recv groups)" if c_abi
                    else "fft_amd.dist over torch.distributed",
           "row_chunks": fd._chunk_count(None, 1 << 20, P), "col_groups": fd._group_count(None, 1 << 20, P, 1),
           "schedule": "column groups exchanged, transformed and sent on while the next group travels; the first and the "
                       "last group in row chunks"}
    one_gpu = {}
    rpath = os.path.join(ROOT, "profiles", "one_gpu_reference.json")
    if os.path.exists(rpath):
        with open(rpath) as fh:
            one_gpu = json.load(fh)

    def allsum(t):
        """sum of a small float64 tensor over the ranks (on the host for backends without device support)"""
        if P == 1:
            return t
        if comm.backend == "nccl":
            dist.all_reduce(t)
            return t
        c = t.cpu()
        dist.all_reduce(c)
        return c.to(t.device)

    def bracket(fn, inner, reps):
        fn()
        torch.cuda.synchronize()
        best, best_ex = float("inf"), []
        for _ in range(reps):
            torch.cuda.synchronize()
            if P > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(inner):
                comm.timings = [] if i == inner - 1 else None
                fn()
            torch.cuda.synchronize()
            if P > 1:
                dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / inner], dtype=torch.float64, device="cuda")
            if P > 1:
                tc = t if comm.backend == "nccl" else t.cpu()
                dist.all_reduce(tc, op=dist.ReduceOp.MAX)
                t = tc
            ex = [{"exchange": lab, "ms": e0.elapsed_time(e1)} for lab, e0, e1 in (comm.timings or [])]
            comm.timings = None
            if float(t.item()) < best:
                best, best_ex = float(t.item()), ex
        return best, best_ex

    def entry(t, ex, points, log2n, esz, key, **extra):
        gf = 5.0 * points * log2n / t / 1e9
        ref = one_gpu.get(key)
        d = {"ms": t * 1e3, "gflops": gf, "gbs_algorithmic": 2.0 * esz * points / t / 1e9,
             "roofline_frac_of_all_gpus": 2.0 * esz * points / t / 1e9 / (HBM_PEAK_GBS * P),
             "exchanges_rank0": ex, "one_gpu_gflops": ref.get("gflops") if ref else None,
             "speedup_vs_one_gpu": gf / ref["gflops"] if ref else None}
        d.update(extra)
        return d

    # ---- config 4: 2-D complex64, row slabs -------------------------------------------------
    try:
        n = args.dist_n2d
        R, W = n // P, n // P
        g = torch.Generator(device="cuda").manual_seed(40 + rank)
        a = torch.view_as_complex(torch.randn((R * n, 2), dtype=torch.float32, device="cuda", generator=g))
        a2 = a.view(R, n)
        energy = allsum(torch.sum(torch.view_as_real(a).double() ** 2).reshape(1))
        rms = math.sqrt(float(energy.item()))
        j = torch.arange(n, dtype=torch.int64, device="cuda")
        bins = [(0, 0), (1, 0), (0, 1), (n // 2 - 1, n // 2), (n - 1, n - 1), (12345 % n, 6789 % n), (n // 4, 1), (1, n // 4)]

        def direct(k0, k1):
            w1 = torch.exp(-2j * math.pi * ((j * k1) % n).double() / n)
            w0 = torch.exp(-2j * math.pi * ((j[rank * R:(rank + 1) * R] * k0) % n).double() / n)
            acc = torch.zeros((), dtype=torch.complex128, device="cuda")
            for lo in range(0, R, 2048):
                acc = acc + torch.dot(w0[lo:lo + 2048], a2[lo:lo + 2048].to(torch.complex128) @ w1)
            return acc

        want = allsum(torch.view_as_real(torch.stack([direct(k0, k1) for k0, k1 in bins])).contiguous())
        for restore in (False, True):
            f = fd.fft2_slab(a, n, n, xcomm, ops, restore_layout=restore)
            got = torch.zeros(len(bins), dtype=torch.complex128, device="cuda")
            for i, (k0, k1) in enumerate(bins):
                if restore and k0 // R == rank:
                    got[i] = f[(k0 - rank * R) * n + k1]
                if not restore and k1 // W == rank:
                    got[i] = f[k0 * W + (k1 - rank * W)]
            got = allsum(torch.view_as_real(got).contiguous())
            err = float(torch.max(torch.linalg.vector_norm(got - want, dim=-1)).item()) / rms
            del f
            t, ex = bracket(lambda: fd.fft2_slab(a, n, n, xcomm, ops, restore_layout=restore), inner=5, reps=3)
            name = f"config4_2d_{n}x{n}_c64" + ("_row_slabs_out" if restore else "_column_slabs_out")
            out[name] = entry(t, ex, n * n, 2 * math.log2(n), 8, f"config4_2d_{n}", scaling="strong",
                              all_to_alls=2 if restore else 1, verified_max_err_over_rms=err, verified_bins=len(bins))
            if not err < 1e-5:
                out[name]["error"] = "sampled bins disagree with the direct DFT"
        del a, a2
    except Exception as exc:                                  # noqa: BLE001 - never sink the headline
        out["config4_error"] = repr(exc)[:300]
    ops.close()
    if c_abi:
        for h in xcomm._plans.values():
            fd._lib.load().mfft_dist_plan_destroy(h)
        xcomm._plans.clear()
    torch.cuda.empty_cache()

    # ---- config 5: one long complex128 signal, four-step ----------------------------------------
    try:
        lg = args.dist_log2n1d or (31 if P >= 8 else 28 + int(round(math.log2(P))))
        n1, n2 = fd.fourstep_factors(lg)
        n = n1 * n2
        cnt = n // P
        g = torch.Generator(device="cuda").manual_seed(50 + rank)
        x = torch.empty(cnt, dtype=torch.complex128, device="cuda")
        xr = torch.view_as_real(x)
        for lo in range(0, cnt, 1 << 26):
            xr[lo:lo + (1 << 26)].normal_(generator=g)
        e_loc = 0.0
        for lo in range(0, cnt, 1 << 26):
            e_loc += float(torch.sum(xr[lo:lo + (1 << 26)] ** 2).item())
        rms = math.sqrt(float(allsum(torch.tensor([e_loc], dtype=torch.float64, device="cuda")).item()))
        ks = [0, 1, n1 - 1, n1, n // 2 - 1, n // 2, n - n2, n - 1, 1234567891 % n, 987654321 % n]

        def direct1(k):
            acc = torch.zeros((), dtype=torch.complex128, device="cuda")
            for lo in range(0, cnt, 1 << 25):
                hi = min(cnt, lo + (1 << 25))
                jj = torch.arange(rank * cnt + lo, rank * cnt + hi, dtype=torch.int64, device="cuda")
                ang = -2.0 * math.pi * (((jj * int(k)) % n).double() / n)          # exact: j*k < 2^62
                acc = acc + torch.sum(x[lo:hi] * torch.complex(torch.cos(ang), torch.sin(ang)))
                del jj, ang
            return acc

        want = allsum(torch.view_as_real(torch.stack([direct1(k) for k in ks])).contiguous())
        for natural in (True, False):
            y = fd.fft1d_fourstep(x, n1, n2, xcomm, ops, natural_order=natural)
            got = torch.zeros(len(ks), dtype=torch.complex128, device="cuda")
            for i, k in enumerate(ks):
                if natural:
                    if k // cnt == rank:
                        got[i] = y[k - rank * cnt]
                else:                                   # out[i*n2 + k2] = X[(rank*(n1/P) + i) + n1*k2]
                    k1, k2 = k % n1, k // n1
                    if k1 // (n1 // P) == rank:
                        got[i] = y[(k1 - rank * (n1 // P)) * n2 + k2]
            got = allsum(torch.view_as_real(got).contiguous())
            err = float(torch.max(torch.linalg.vector_norm(got - want, dim=-1)).item()) / rms
            del y
            t, ex = bracket(lambda: fd.fft1d_fourstep(x, n1, n2, xcomm, ops, natural_order=natural), inner=2, reps=2)
            name = f"config5_1d_2pow{lg}_c128" + ("_natural_order" if natural else "_transposed_order")
            out[name] = entry(t, ex, n, lg, 16, f"config5_1d_2pow{lg}", scaling="strong",
                              all_to_alls=3 if natural else 2, verified_max_err_over_rms=err, verified_bins=len(ks))
            if not err < 1e-12:
                out[name]["error"] = "sampled bins disagree with the direct DFT"
        del x, xr
    except Exception as exc:                                  # noqa: BLE001
        out["config5_error"] = repr(exc)[:300]
    ops.close()
    if c_abi:
        xcomm.close()
    torch.cuda.empty_cache()
    return out


class _Reporter:
    """Keeps rank 0's one JSON line safe from the process that measures it.  A child forked before anything
    touches the GPU reads versions of the line from a pipe and writes the latest complete one to stdout when
    the pipe closes: after the final version, or when the parent dies inside one of the sharded-transform legs
    that follow the headline (a GPU fault or an abort in a collective cannot be caught in Python)."""

    def __init__(self):
        import signal
        rd, wr = os.pipe()
        sys.stdout.flush()
        pid = os.fork()
        if pid == 0:
            try:
                os.close(wr)
                for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                    signal.signal(sig, signal.SIG_IGN)          # the launcher ends its workers with these
                buf = b""
                while True:
                    chunk = os.read(rd, 1 << 16)
                    if not chunk:
                        break
                    buf += chunk
                    cut = buf.rfind(b"\n", 0, buf.rfind(b"\n"))  # keep the last complete line and the open tail
                    if cut >= 0:
                        buf = buf[cut + 1:]
                done = buf[:buf.rfind(b"\n") + 1]
                if done:
                    os.write(1, done)
            finally:
                os._exit(0)
        os.close(rd)
        self._wr, self._pid = wr, pid

    def update(self, line):
        data = (json.dumps(line) + "\n").encode()
        while data:
            data = data[os.write(self._wr, data):]

    def close(self):
        os.close(self._wr)
        os.waitpid(self._pid, 0)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    world_env, spawn = resolve_world(args, os.environ)
    if spawn:
        # no launcher: start the ranks as fresh processes; this process never touches a GPU
        return launch_ranks(world_env, argv, args.master_port)
    if args.rendezvous_only:
        return rendezvous_only(args)
    # several ranks (or the one-rank plumbing check of the sharded legs): rank 0 hands its line to a reporter
    reporter = None
    if int(os.environ.get("RANK", "0")) == 0 and (world_env > 1 or args.dist_configs) and not args.no_dist_configs:
        reporter = _Reporter()
    import numpy as np
    import torch
    import fft_amd

    world = world_env
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(0 if args.rehearse_one_gpu else local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(args.backend)
    n, batch = args.n, args.batch
    log2n = int(round(math.log2(n)))
    assert 1 << log2n == n, "bench workload is a power-of-two length"

    x_host = synth(batch, n, seed=rank)
    d_x = torch.from_numpy(x_host).cuda()
    d_s = torch.empty((batch, n), dtype=torch.float32, device="cuda")
    plan = fft_amd.Plan(n, batch=batch, dtype=fft_amd.C64, flags=fft_amd.HANN | fft_amd.POWER)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.exec(d_x, d_s, scale=DT, stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()      # the first barrier sets up the communicator: keep that out of the settled phase
    # untimed: sustained load until the clocks have settled -- at least `settle` seconds, then on
    # until a 100-launch batch is within 1 % of the fastest batch seen (at most 4 x `settle`)
    t_settle = time.perf_counter()
    fastest = float("inf")
    while True:
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        for _ in range(100):
            step()
        b1.record()
        torch.cuda.synchronize()
        took = b0.elapsed_time(b1)
        fastest = min(fastest, took)
        waited = time.perf_counter() - t_settle
        if waited >= args.settle and (took <= 1.01 * fastest or waited >= 4 * args.settle):
            break
    for _ in range(args.warmup):
        step()
    # Average launch duration of the (single) kernel of a step: two events on the
    # launch stream around the K back-to-back launches of the timed region.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_avg_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps

    flops_step = 5.0 * n * log2n * batch * world
    alg_bytes = (8 + 4) * n * batch                       # per launch, per GPU
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and n == N_POINTS and batch == BATCH:
        with open(tpath) as fh:
            traffic = json.load(fh).get("bytes_per_launch")
        traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc passes over this command recorded earlier "
                          "(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, per launch); "
                          "not collected in this run")
    line = {
        "metric": "FFT GFLOP/s (5N·log2N) + achieved HBM GB/s vs roofline",
        "value": flops_step * args.steps / elapsed / 1e9,
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (numpy default_rng(rank), i.i.d. standard normal re/im, complex64)",
        "config": {"workload": f"batched 1-D complex64 FFT, N={n} x {batch} signals per GPU, "
                               "fused Hann window + power spectrum (BASELINE config 2)",
                   "n": n, "batch_per_gpu": batch, "parallelism": f"batch-sharded x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / kern_avg_s / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg_bytes / kern_avg_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "kernel": "fft_lds_tail_kernel (hann+power)",
                     "kernel_ms_avg": kern_avg_s * 1e3,
                     "algorithmic_bytes_per_launch": alg_bytes},
    }
    # the timed launches' own output, 64 sampled signals (first, last and a spread over the whole
    # batch, so every trip of the persistent workgroups is represented) against the CPU oracle
    rows = np.unique(np.concatenate(([0, batch - 1], np.linspace(0, batch - 1, 62).astype(np.int64))))
    got = d_s[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
    from oracle import spectral_oracle as so
    _, ref = so.windowed_power(x_host[rows], use_hann=True, dt=DT, engine="numpy")
    line["verified_rel_l2"] = float(np.max(np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)))
    line["verified_rows"] = int(rows.size)
    if not line["verified_rel_l2"] < 1e-5:
        raise SystemExit(f"bench output fails parity with the oracle: rel L2 {line['verified_rel_l2']:.3e}")
    if rank == 0 and world == 1 and not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(x_host, n)
        line["cpu_baseline_multicore"] = cpu_multicore(x_host, n)
    else:
        line["cpu_baseline"] = None
    plan.destroy()
    del d_x, d_s
    if rank == 0 and world == 1 and not args.no_secondary and n == N_POINTS and batch == BATCH:
        torch.cuda.empty_cache()
        line["other_configs"] = secondary_configs(with_cpu=not args.no_cpu)
    if (distributed or args.dist_configs) and not args.no_dist_configs:
        # every rank takes part; rank 0 reports
        torch.cuda.empty_cache()
        # Both legs are fenced: a collective that never returns must not cost the job its headline.  When a
        # leg has not finished in time every rank leaves, rank 0 after printing the line with what the
        # job has measured so far.
        fenced(line, "dist_configs", args.dist_timeout, lambda: dist_configs(args, world, rank), rank, reporter)
        if (args.backend == "nccl" or not distributed) and not args.no_dist_c_abi:
            # Same configs through libmfft's own RCCL calls (this route has run on RCCL with one rank and
            # on several in-process ranks only).
            fenced(line, "dist_configs_c_abi", args.c_abi_timeout,
                   lambda: dist_configs(args, world, rank, c_abi=True), rank, reporter)
    if rank == 0:
        if reporter is not None:
            reporter.update(line)
            reporter.close()
        else:
            print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
