#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box):

    python tests/golden/make_golden.py

The reference apps cannot be imported as modules (they `import streamlit`, which
is not installed, and run their whole UI at import time), so this script parses
each file, keeps only the top-level function definitions, the non-streamlit
imports and the plain constant assignments that precede the UI block, and
executes those in a fresh namespace.  The functions are then *called* on the
apps' default parameters and on seeded synthetic inputs; only their inputs and
outputs (numbers) are written to the fixtures.  No reference source text is
stored anywhere in this repository.

Fixtures written (all numpy .npz, allow_pickle never needed):
  final2_defaults.npz     fft_from_timeseries at final2.py's sidebar defaults, m = 0, 1, 2
  combined_defaults.npz   fft_extract_E_p / compute_all at streamlit_app_combined.py defaults
  cfg1_real4096.npz       N = 4096 real float64 three-tone signal through fft_from_timeseries
  raw_fft.npz             np.fft.fft(x, n) exactly as the reference calls it, seeded inputs
  errors.json             the ValueError messages of fft_from_timeseries' input validation
"""
import ast
import hashlib
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _mentions(node, name):
    return any(isinstance(n, ast.Name) and n.id == name for n in ast.walk(node))


def load_reference_functions(path, ui_start_line):
    """Exec the function definitions of one reference app, skipping its UI."""
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), filename=path)
    kept = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [a.name for a in node.names]
            mod = getattr(node, "module", None) or ""
            if any(n.startswith("streamlit") for n in names) or mod.startswith("streamlit"):
                continue
            kept.append(node)
        elif isinstance(node, ast.FunctionDef):
            kept.append(node)
        elif isinstance(node, ast.Assign):
            if node.lineno < ui_start_line and not _mentions(node, "st"):
                kept.append(node)
    module = ast.Module(body=kept, type_ignores=[])
    ns = {"__name__": "reference_functions"}
    os.environ.setdefault("MPLBACKEND", "Agg")
    exec(compile(module, path, "exec"), ns)
    return ns


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def subsample(a, step):
    return np.ascontiguousarray(a[::step])


def make_final2(ns):
    K = [1.0, 1.0, 1.0]
    J = [1.0, 1.0]
    dt, tmax, pad = 0.01, 80.0, 32
    times = np.arange(0.0, tmax + dt, dt)
    kw = dict(n_levels=3, pad_factor=pad, include_dc=True, guard_mult=2.0,
              min_weight=1e-5, keep_top=3)
    out = dict(K=np.array(K), J=np.array(J), dt=dt, tmax=tmax, pad=pad,
               n_levels=3, include_dc=True, guard_mult=2.0, min_weight=1e-5, keep_top=3,
               times_len=len(times))
    a_lists, b_lists, m_done = [], [], []
    for m in (0, 1, 2):
        f_t = ns["return_amplitude_block_general"](K, J, m, 0, times)   # site=0 as at the call sites
        E, p, Eaxis, S_norm = ns["fft_from_timeseries"](f_t, times, **kw)
        out[f"f_t_{m}"] = f_t
        out[f"E_{m}"] = E
        out[f"p_{m}"] = p
        out[f"argmax_{m}"] = int(np.argmax(S_norm))
        out[f"Snorm_sub_{m}"] = subsample(S_norm, 61)
        out[f"Snorm_sum_{m}"] = float(S_norm.sum())
        out[f"Snorm_l2_{m}"] = float(np.linalg.norm(S_norm))
        out[f"Eaxis_sub_{m}"] = subsample(Eaxis, 61)
        out[f"Eaxis_sha_{m}"] = sha(Eaxis)
        out[f"Snorm_len_{m}"] = len(S_norm)
        # window around the strongest line (peak finder + refine act here)
        k = int(np.argmax(S_norm))
        lo, hi = max(0, k - 64), min(len(S_norm), k + 65)
        out[f"Snorm_peakwin_{m}"] = S_norm[lo:hi]
        out[f"Snorm_peakwin_lo_{m}"] = lo
        if len(E) > 0:
            a, b, _, T, G, evals_T, wT = ns["lanczos_detailed"](E, p)
            out[f"lanczos_a_{m}"] = a
            out[f"lanczos_b_{m}"] = b
            if len(a) == len(K) and len(b) == len(K) - 1:      # the app's completeness test
                a_lists.append(a); b_lists.append(b); m_done.append(m)
    if a_lists:
        J_est, _ = ns["recover_J_from_blocks"](b_lists)
        K_est, S_est, _ = ns["recover_K_from_blocks_with_m"](a_lists, m_done)
        out["J_est"] = J_est
        out["K_est"] = K_est
        out["S_est"] = S_est
        out["m_done"] = np.array(m_done)
    # the function's own defaults (differ from the call-site values above)
    f_t = out["f_t_1"]
    E, p, Eaxis, S_norm = ns["fft_from_timeseries"](f_t, times)
    out["own_defaults_E_1"] = E
    out["own_defaults_p_1"] = p
    out["own_defaults_Snorm_sub_1"] = subsample(S_norm, 61)
    # rectangular window, no DC, no pruning
    E, p, Eaxis, S_norm = ns["fft_from_timeseries"](f_t, times, n_levels=3, pad_factor=4,
                                                    use_hann=False, post_prune=False)
    out["rect_E_1"] = E
    out["rect_p_1"] = p
    out["rect_Snorm_sub_1"] = subsample(S_norm, 61)
    np.savez_compressed(os.path.join(HERE, "final2_defaults.npz"), **out)
    print("final2:", {m: (out[f"E_{m}"], out[f"p_{m}"], out[f"argmax_{m}"]) for m in (0, 1, 2)})


def make_combined(ns):
    cfg = dict(K14=0.0, K25=0.0, K36=0.0, J12=1.0, J23=1.5, a=1.0, b=0.0, c=0.0,
               dt=0.01, tmax=15.0, site=2, pad=16, use_hann=True, n_peaks=3)
    times = np.arange(0.0, cfg["tmax"] + cfg["dt"], cfg["dt"])
    Hfull = ns["build_H_full"](cfg["K14"], cfg["K25"], cfg["K36"], cfg["J12"], cfg["J23"])
    H1, _ = ns["extract_1exc"](Hfull)
    Htop, _, _ = ns["extract_top"](H1)
    f_t = ns["return_amplitude"](Htop, site=cfg["site"], times=times)
    E_est, p_est, omega, S, mu = ns["fft_extract_E_p"](
        Htop, site=cfg["site"], times=times, n_peaks=cfg["n_peaks"], pad=cfg["pad"],
        use_hann=cfg["use_hann"], demod=True)
    E_true, p_true = ns["eigvals_and_site_weights"](Htop, site=cfg["site"])
    a_j, b_j, T_l, _ = ns["jacobi_from_spectral"](E_true, p_true)
    T_ABC = ns["reorder_T_lanczos_to_ABC"](T_l, site=cfg["site"])
    Heff = ns["build_Heff_from_paper"](cfg["K14"], cfg["K25"], cfg["K36"], cfg["J12"], cfg["J23"])
    out = dict(Htop=Htop, times_len=len(times), f_t=f_t, E_est=E_est, p_est=p_est, mu=mu,
               S=S, S_sum=float(S.sum()), omega_sha=sha(omega), omega_sub=subsample(omega, 7),
               E_true=E_true, p_true=p_true, T_ABC=T_ABC, Heff=Heff,
               T_ABC_close=bool(np.allclose(T_ABC, Heff, atol=1e-9)),
               **{k: v for k, v in cfg.items()})
    # no demodulation, no window, other sites / pads
    for tag, kw in (("nodemod", dict(demod=False)),
                    ("rect", dict(use_hann=False)),
                    ("site0_pad4", dict(site=0, pad=4)),
                    ("case1", dict())):
        H = Htop
        site = kw.pop("site", cfg["site"])
        pad = kw.pop("pad", cfg["pad"])
        if tag == "case1":
            Hf = ns["build_H_full"](1.0, 1.5, 0.0, 1.0, 2.0)
            H, _, _ = ns["extract_top"](ns["extract_1exc"](Hf)[0])
            out["case1_Htop"] = H
        E2, p2, om2, S2, mu2 = ns["fft_extract_E_p"](H, site=site, times=times, n_peaks=3,
                                                      pad=pad, **kw)
        out[f"{tag}_E"] = E2
        out[f"{tag}_p"] = p2
        out[f"{tag}_mu"] = mu2
        out[f"{tag}_S_sum"] = float(S2.sum())
        out[f"{tag}_S_sub"] = subsample(S2, 7)
    np.savez_compressed(os.path.join(HERE, "combined_defaults.npz"), **out)
    print("combined:", E_est, p_est, mu, out["S_sum"], out["T_ABC_close"])


def three_tone(n=4096, dt=0.01):
    t = np.arange(n) * dt
    E = np.array([-3.0, 0.5, 4.25])
    p = np.array([0.2, 0.5, 0.3])
    f = (p[None, :] * np.exp(-1j * E[None, :] * t[:, None])).sum(axis=1)
    return t, f.real.copy()      # real float64 signal (cfg1)


def make_cfg1(ns):
    t, x = three_tone()
    E, p, Eaxis, S_norm = ns["fft_from_timeseries"](x, t, n_levels=6, pad_factor=1,
                                                    include_dc=False, keep_top=6)
    np.savez_compressed(os.path.join(HERE, "cfg1_real4096.npz"),
                        E=E, p=p, Eaxis=Eaxis, S_norm=S_norm, n=4096, dt=0.01)
    print("cfg1:", E, p)


def make_raw():
    """np.fft.fft(x, n=N) on seeded inputs: the call shape of final2.py:143."""
    out = {}
    sizes = [1, 2, 3, 4, 5, 8, 16, 30, 64, 100, 128, 256, 512, 1000, 1024, 2048, 4096, 8192,
             16384, 24016, 65536, 256032]
    for dtype, tag in ((np.complex128, "c128"), (np.complex64, "c64")):
        for n in sizes:
            rng = np.random.default_rng(n)
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dtype)
            X = np.fft.fft(x, n=n)
            assert X.dtype == dtype
            step = max(1, n // 64)
            out[f"{tag}_{n}_sub"] = subsample(X, step)
            out[f"{tag}_{n}_l2"] = float(np.linalg.norm(X.astype(np.complex128)))
            out[f"{tag}_{n}_x_head"] = x[:4]
    # zero-padded call shape: n_in < n
    for n_in, n in ((8001, 256032), (1501, 24016), (100, 128), (1024, 4096)):
        rng = np.random.default_rng(n_in)
        x = rng.standard_normal(n_in) + 1j * rng.standard_normal(n_in)
        X = np.fft.fft(x, n=n)
        step = max(1, n // 64)
        out[f"pad_{n_in}_{n}_sub"] = subsample(X, step)
        out[f"pad_{n_in}_{n}_l2"] = float(np.linalg.norm(X))
    out["sizes"] = np.array(sizes)
    out["numpy_version"] = np.__version__
    np.savez_compressed(os.path.join(HERE, "raw_fft.npz"), **out)
    print("raw: ok")


def make_errors(ns):
    f = ns["fft_from_timeseries"]
    msgs = {}
    for tag, args in (("ndim", (np.ones(4), np.ones((2, 2)))),
                      ("short", (np.ones(1), np.ones(1))),
                      ("len", (np.ones(5), np.arange(4.0))),
                      ("uniform", (np.ones(4), np.array([0.0, 1.0, 2.0, 4.0])))):
        try:
            f(*args)
            msgs[tag] = None
        except ValueError as e:
            msgs[tag] = str(e)
    try:
        ns["lanczos_detailed"]([1.0, 2.0], [0.0, 0.0])
        msgs["lanczos_zero"] = None
    except ValueError as e:
        msgs["lanczos_zero"] = str(e)
    with open(os.path.join(HERE, "errors.json"), "w") as fh:
        json.dump(msgs, fh, indent=1, sort_keys=True)
    print("errors:", msgs)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    f2 = load_reference_functions(os.path.join(REF, "final2.py"), ui_start_line=526)
    cb = load_reference_functions(os.path.join(REF, "streamlit_app_combined.py"), ui_start_line=253)
    make_final2(f2)
    make_combined(cb)
    make_cfg1(f2)
    make_raw()
    make_errors(f2)


if __name__ == "__main__":
    main()
