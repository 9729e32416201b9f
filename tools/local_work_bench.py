#!/usr/bin/env python3
"""Per-GPU LOCAL work of the multi-GPU decompositions at the shapes an N-GPU run gives each rank,
timed step by step on one GPU, and the schedule those steps form with the exchanges.

    python tools/local_work_bench.py [--ranks 2 4 8] [--groups 1 2 4] [--chunks 4] [--json FILE]

Every local step of fft_amd/csrc/dist.hip's schedules (pack, column transforms of one group, row
transforms of one chunk, unpack, final transpose) runs on its real per-rank shape and layout and is timed
with events.  The exchanges cannot run on one GPU: each is priced at (bytes per peer) / 153 GB/s, the
xGMI per-link peak (a full mesh: all peers at once).  The two in-order queues of an execution -- the
kernels' stream and the exchange stream -- are then replayed with the dependencies the library enqueues,
which gives, per configuration:

    total        the projected time of one transform (a PROJECTION: measured kernels + modelled links)
    idle link    how long the links wait while kernels run (the un-overlapped local work), and the
                 longest single such stretch
    vs one GPU   against the one-GPU time of the same transform (measured here when it fits, else taken
                 from profiles/one_gpu_reference.json)
"""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import dist as fd  # noqa: E402

LINK_GBS = 153.0


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def replay(kernels, links):
    """Two in-order queues.  kernels / links: lists of (name, seconds, [names it waits for]).  Returns
    (makespan, seconds the links idle while a kernel runs, the longest such stretch, finish times)."""
    done, busy = {}, {"k": [], "l": []}
    qi = {"k": 0, "l": 0}
    clock = {"k": 0.0, "l": 0.0}
    queues = {"k": kernels, "l": links}
    progress = True
    while progress:
        progress = False
        for q in ("k", "l"):
            while qi[q] < len(queues[q]):
                name, secs, deps = queues[q][qi[q]]
                if any(d not in done for d in deps):
                    break
                start = max([clock[q]] + [done[d] for d in deps])
                done[name] = start + secs
                clock[q] = done[name]
                busy[q].append((start, done[name]))
                qi[q] += 1
                progress = True
    assert qi["k"] == len(kernels) and qi["l"] == len(links), "dependency cycle in the schedule"
    makespan = max(clock.values())
    # kernel-busy time not covered by link-busy time
    idle_total, idle_longest = 0.0, 0.0
    for ks, ke in busy["k"]:
        cuts = sorted((max(ks, ls), min(ke, le)) for ls, le in busy["l"] if le > ks and ls < ke)
        at = ks
        for cs, ce in cuts:
            if cs > at:
                idle_total += cs - at
                idle_longest = max(idle_longest, cs - at)
            at = max(at, ce)
        if ke > at:
            idle_total += ke - at
            idle_longest = max(idle_longest, ke - at)
    return makespan, idle_total, idle_longest, done


def model_1d(P, G, C, natural, t, piece_bytes):
    """queues of exec_1d; t: step times, piece_bytes(unit) -> bytes one peer receives"""
    link = lambda b: b / (LINK_GBS * 1e9)                                     # noqa: E731
    k, l = [], []
    for g in range(G):
        nck = C if g == 0 else 1
        for c in range(nck):
            k.append((f"pack{g}.{c}", t["pack_first"] if g == 0 else t["pack"], []))
            l.append((f"ex1.{g}.{c}", link(piece_bytes("group") / nck), [f"pack{g}.{c}"]))
    for g in range(G):
        k.append((f"col{g}", t["col"], [f"ex1.{g}.{(C if g == 0 else 1) - 1}"]))
        nck = C if g == G - 1 else 1
        for c in range(nck):
            l.append((f"ex2.{g}.{c}", link(piece_bytes("group") / nck), [f"col{g}"]))
    for c in range(C):
        k.append((f"row{c}", t["row"], [f"ex2.{G - 1}.{c}"]))
        if natural:
            l.append((f"ex3.{c}", link(piece_bytes("all") / C), [f"row{c}"]))
    if natural:
        k.append(("transpose", t["transpose"], [f"ex3.{C - 1}"]))
    return k, l


def model_2d(P, G, C, restore, t, piece_bytes):
    link = lambda b: b / (LINK_GBS * 1e9)                                     # noqa: E731
    k, l = [], []
    for c in range(C):
        k.append((f"row{c}", t["row"], []))
    for g in range(G):
        nck = C if g == 0 else 1
        for c in range(nck):
            l.append((f"ex1.{g}.{c}", link(piece_bytes("group") / nck), [f"row{c}"] if g == 0 else [f"row{C - 1}"]))
    for g in range(G):
        k.append((f"col{g}", t["col"], [f"ex1.{g}.{(C if g == 0 else 1) - 1}"]))
        if restore:
            nck = C if g == G - 1 else 1
            for c in range(nck):
                l.append((f"ex2.{g}.{c}", link(piece_bytes("group") / nck), [f"col{g}"]))
    if restore:
        for c in range(C):
            k.append((f"unpack{c}", t["unpack"], [f"ex2.{G - 1}.{c}"]))
    return k, l


def steps_1d(ops, P, G, C, n1, n2, natural):
    R1, W2 = n1 // P, n2 // P
    Rc, Wc = R1 // C, W2 // G
    local = R1 * n2
    x = torch.view_as_complex(torch.randn((local, 2), dtype=torch.float64, device="cuda"))
    a, b = torch.empty_like(x), torch.empty_like(x)
    gsz = n1 * Wc
    t = {}
    t["pack_first"] = timed(lambda: ops.swap_axes(x, a, a=Rc, b=P, w=Wc, strides=(n2, Wc)))
    t["pack"] = timed(lambda: ops.swap_axes(x, a, a=R1, b=P, w=Wc, strides=(n2, Wc)))
    t["col"] = timed(lambda: ops.fft_columns(a[:gsz], b[:gsz], n=n1, inner=Wc, twiddle=(n1 * n2, Wc)))
    seg_in, seg_out = (Wc, Wc, R1 * Wc), (W2, W2, R1 * W2)
    if natural:
        t["row"] = timed(lambda: ops.fft(a, b, n=n2, batch=Rc, in_layout=seg_in, out_layout=seg_out))
        t["transpose"] = timed(lambda: ops.transpose(a, b, rows=n1, cols=W2))
    else:
        t["row"] = timed(lambda: ops.fft(a, b, n=n2, batch=Rc, in_layout=seg_in))
    del x, a, b
    return t, (lambda kind: (R1 * Wc if kind == "group" else R1 * W2) * 16)


def steps_2d(ops, P, G, C, n, restore):
    R, W = n // P, n // P
    Rc, Wc = R // C, W // G
    x = torch.view_as_complex(torch.randn((R * n, 2), dtype=torch.float32, device="cuda"))
    a, b = torch.empty_like(x), torch.empty_like(x)
    gsz = n * Wc
    t = {}
    t["row"] = timed(lambda: ops.fft(x, a, n=n, batch=Rc, out_layout=(Wc, Wc, R * Wc)))
    if restore:
        t["col"] = timed(lambda: ops.fft_columns(a[:gsz], b[:gsz], n=n, inner=Wc))
        t["unpack"] = timed(lambda: ops.swap_axes(a, b, a=P * G, b=Rc, w=Wc, strides=(R * Wc, Wc)))
    else:
        t["col"] = timed(lambda: ops.fft_columns(a[:gsz], b, n=n, inner=Wc, out_ld=W))
    del x, a, b
    return t, (lambda kind: R * Wc * 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--log2n", type=int, default=0, help="length of the 1-D transform (0: 2^(28 + log2 ranks), 2^31 on 8)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    ops = fd.HipOps()
    one_gpu = {}
    rpath = os.path.join(ROOT, "profiles", "one_gpu_reference.json")
    if os.path.exists(rpath):
        with open(rpath) as fh:
            one_gpu = json.load(fh)
    rows = []
    for P in args.ranks:
        for G in args.groups:
            C = args.chunks
            for restore in (False, True):
                n = 16384
                if (n // P) % G or (n // P) // G < 128:
                    continue
                t, pb = steps_2d(ops, P, G, C, n, restore)
                k, l = model_2d(P, G, C, restore, t, pb)
                total, idle, longest, _ = replay(k, l)
                ref = one_gpu.get(f"config4_2d_{n}")
                flops = 5.0 * n * n * 2 * math.log2(n)
                rows.append({"config": f"2-D {n}^2 complex64, {'row' if restore else 'column'} slabs out", "ranks": P, "col_groups": G,
                             "chunks": C, "steps_ms": {s: v * 1e3 for s, v in t.items()},
                             "local_ms": sum(s for _, s, _ in k) * 1e3, "link_ms": sum(s for _, s, _ in l) * 1e3,
                             "projected_total_ms": total * 1e3, "link_idle_while_kernels_run_ms": idle * 1e3,
                             "longest_unoverlapped_local_stretch_ms": longest * 1e3,
                             "projected_speedup_vs_one_gpu": (flops / total / 1e9) / ref["gflops"] if ref else None})
                print(json.dumps(rows[-1]), flush=True)
                ops.close()
                torch.cuda.empty_cache()
            for natural in (False, True):
                lg = args.log2n or (31 if P >= 8 else 28 + int(round(math.log2(P))))
                n1, n2 = fd.fourstep_factors(lg)
                if (n2 // P) % G or (n2 // P) // G < 64:
                    continue
                t, pb = steps_1d(ops, P, G, C, n1, n2, natural)
                k, l = model_1d(P, G, C, natural, t, pb)
                total, idle, longest, _ = replay(k, l)
                ref = one_gpu.get(f"config5_1d_2pow{lg}")
                flops = 5.0 * n1 * n2 * lg
                rows.append({"config": f"1-D 2^{lg} complex128, {'natural' if natural else 'transposed'} order", "ranks": P,
                             "col_groups": G, "chunks": C, "steps_ms": {s: v * 1e3 for s, v in t.items()},
                             "local_ms": sum(s for _, s, _ in k) * 1e3, "link_ms": sum(s for _, s, _ in l) * 1e3,
                             "projected_total_ms": total * 1e3, "link_idle_while_kernels_run_ms": idle * 1e3,
                             "longest_unoverlapped_local_stretch_ms": longest * 1e3,
                             "projected_speedup_vs_one_gpu": (flops / total / 1e9) / ref["gflops"] if ref else None})
                print(json.dumps(rows[-1]), flush=True)
                ops.close()
                torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"link_model_GBps_per_peer": LINK_GBS, "note": "kernel times measured on one MI355X at per-rank shapes; "
                       "exchange times modelled; totals are projections, not measurements", "rows": rows}, fh, indent=1)
    ops.close()


if __name__ == "__main__":
    main()
