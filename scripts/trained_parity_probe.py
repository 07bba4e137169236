#!/usr/bin/env python
"""Parity probe in the CONVERGING regime: a TRAINED optimizer (.l2l from scripts/train_*.py)
on the full-size BASELINE configurations -- HIP (both two-CU forms) against the fp32 C oracle and the float64
NumPy oracle, at T = 100 / 1000 / 10 000, plus the per-step gradient error of each form against the float64
gradient at the kernel's own iterates.

    python scripts/trained_parity_probe.py --weights tests/golden/trained --out gpurun_out/r03/probe.json

Uses oracle/ as the checker (this is a test tool, not a product path)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dill  # noqa: E402

import oracle as O  # noqa: E402
from helpers import device_problem, lib_option, make_problem, max_abs, rel_err, spec_of  # noqa: E402
from open_l2o_amd import _abi  # noqa: E402
from oracle.c_oracle import c_unroll  # noqa: E402


def load_l2l(path):
    with open(path, "rb") as f:
        d = dill.load(f)
    return {k: {v: np.asarray(a, np.float32) for v, a in m.items()} for k, m in d.items()}


def run_fused(eng, cfg, params, arrays, x0, B, D, T, Bg=None, hist=False):
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=Bg)
    x, st, m, v = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D), eng.zeros(B, D), eng.zeros(B, D)
    fx_part, fx = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    h = None
    if hist:
        h = dict(st=eng.zeros(T, eng.state_floats(B, D)), g=eng.zeros(T, B * D), g_final=eng.zeros(B * D))
        if cfg.kind == "rnnprop":
            h.update(m=eng.zeros(T, B * D), v=eng.zeros(T, B * D))
    eng.unroll(spec, wpack, pd, x, st, m, v, T, 1, fx_part, hist=h)
    eng.reduce_fx(fx_part, T + 1, B, Bg or B, fx)
    eng.synchronize()
    eng.check_unroll_status()
    return eng.to_numpy(fx), eng.to_numpy(x), h


def f64_oracle(kind, prob, cfg, params, x0, B, D, T, Bg=None):
    p64 = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in params.items()}
    if kind == "quadratic":
        pr = O.Quadratic(prob.w.astype(np.float64), prob.y.astype(np.float64), batch_global=Bg)
    elif kind == "rastrigin":
        pr = O.Rastrigin(prob.A.astype(np.float64), prob.B.astype(np.float64), prob.C.astype(np.float64),
                         alpha=prob.alpha, batch_global=Bg)
    else:
        pr = O.Lasso(prob.w.astype(np.float64), prob.y.astype(np.float64), l=prob.l, batch_global=Bg)
    x64 = x0.astype(np.float64).reshape((B, D, 1) if kind == "rastrigin" else (B, D))
    return O.unroll(pr, cfg, p64, x64, O.net_initial_state(cfg, B * D, np.float64), T)


def grad64(kind, prob, x, Bg):
    """float64 gradient of the reference's loss at the fp32 iterate x [B, D]."""
    x = x.astype(np.float64)
    if kind == "quadratic":
        W, y = prob.w.astype(np.float64), prob.y.astype(np.float64)
        r = np.einsum("bmd,bd->bm", W, x) - y
        return 2.0 / Bg * np.einsum("bmd,bm->bd", W, r)
    if kind == "rastrigin":
        A, Bv, Cv = prob.A.astype(np.float64), prob.B.astype(np.float64)[..., 0], prob.C.astype(np.float64)[..., 0]
        r = np.einsum("bmd,bd->bm", A, x) - Bv
        return (np.einsum("bmd,bm->bd", A, r) + 2 * np.pi * prob.alpha * Cv * np.sin(2 * np.pi * x)) / Bg
    raise ValueError(kind)


def grad32_two_pass(kind, prob, x, Bg):
    """The reference's own arithmetic in fp32 (NumPy): r = Wx - y, then W^T r."""
    f = np.float32
    if kind == "quadratic":
        r = np.einsum("bmd,bd->bm", prob.w, x).astype(f) - prob.y
        return (f(2.0) / f(Bg) * np.einsum("bmd,bm->bd", prob.w, r)).astype(f)
    A, Bv, Cv = prob.A, prob.B[..., 0], prob.C[..., 0]
    r = np.einsum("bmd,bd->bm", A, x).astype(f) - Bv
    return ((np.einsum("bmd,bm->bd", A, r) + f(2 * np.pi * prob.alpha) * Cv * np.sin(f(2 * np.pi) * x)) / f(Bg)).astype(f)


def probe(eng, name, kind, cfg, params, B, D, Bg, seed, horizons, out):
    prob, x0, arrays = make_problem(kind, B, D, seed=seed)
    x0 = x0.reshape(B, D)
    Bg = Bg or B          # slices of the batch keep the 1/B of the batch the optimizer was trained on
    rec = dict(config=name, B=B, D=D, B_global=Bg)
    for T, Bt in horizons:
        # long horizons on a slice of the batch (the float64 NumPy oracle is the slow leg)
        sl = slice(0, Bt)
        arr_t = {k: (v[sl] if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}
        pr_t, _, _ = make_problem(kind, B, D, seed=seed)
        for attr in ("w", "y", "A", "B", "C"):
            if hasattr(pr_t, attr):
                setattr(pr_t, attr, getattr(pr_t, attr)[sl])
        x0t = x0[sl]
        fx32 = c_unroll(kind, cfg, params, arr_t, x0t, T, B_global=Bg)[0]
        r64 = f64_oracle(kind, pr_t, cfg, params, x0t, Bt, D, T, Bg=Bg)
        env = rel_err(fx32, r64.fx)
        row = dict(T=T, B=Bt, fx0=float(r64.fx[0]), fxT=float(r64.fx[-1]), fx_min=float(np.min(r64.fx)),
                   oracle32_vs_64=env)
        for form, label in ((1, "normal"), (0, "two_pass")):
            with lib_option(_abi.OPT_PAIR_NORMAL, form):
                fx, x, _ = run_fused(eng, cfg, params, arr_t, x0t, Bt, D, T, Bg=Bg)
            row[label] = dict(vs64=rel_err(fx, r64.fx), vs32=rel_err(fx, fx32),
                              first101_vs64=rel_err(fx[:101], r64.fx[:101]),
                              x_vs64=max_abs(x, r64.x.reshape(Bt, D)) / max(1.0, float(np.abs(r64.x).max())))
        with lib_option(_abi.OPT_PAIR, 0):
            fx, x, _ = run_fused(eng, cfg, params, arr_t, x0t, Bt, D, T, Bg=Bg)
        row["one_cu"] = dict(vs64=rel_err(fx, r64.fx), vs32=rel_err(fx, fx32))
        print(name, json.dumps(row), flush=True)
        rec.setdefault("horizons", []).append(row)
    # per-step gradient error at the kernel's own iterates (T = 100, full batch)
    if kind in ("quadratic", "rastrigin"):
        T = 100
        grows = []
        for form, label in ((1, "normal"), (0, "two_pass")):
            with lib_option(_abi.OPT_PAIR_NORMAL, form):
                _, _, h = run_fused(eng, cfg, params, arrays, x0, B, D, T, Bg=Bg, hist=True)
                hg = eng.to_numpy(h["g"]).reshape(T, B, D)
                for t in (0, 5, 20, 50, 99):
                    xt = run_fused(eng, cfg, params, arrays, x0, B, D, t, Bg=Bg)[1] if t else x0
                    g64 = grad64(kind, prob, xt, Bg or B)
                    g32 = grad32_two_pass(kind, prob, xt.astype(np.float32), Bg or B)
                    nrm = float(np.linalg.norm(g64))
                    grows.append(dict(form=label, t=t, gnorm=nrm,
                                      rel_err_hip=float(np.linalg.norm(hg[t] - g64)) / nrm,
                                      rel_err_numpy_two_pass=float(np.linalg.norm(g32 - g64)) / nrm))
                    print(name, json.dumps(grows[-1]), flush=True)
        rec["gradient_probe"] = grows
    out.append(rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "tests", "golden", "trained"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", default="c2,c4,c3")
    ap.add_argument("--long", type=int, default=1)
    a = ap.parse_args()
    from open_l2o_amd._engine import HipEngine
    eng = HipEngine()
    out = []
    want = a.configs.split(",")
    long_h = [(1000, 16), (10000, 4)] if a.long else []
    if "c2" in want:
        params = load_l2l(os.path.join(a.weights, "dm_quadratic_d128", "cw.l2l-0"))
        probe(eng, "c2", "quadratic", O.DM_IDENTITY, params, 128, 128, None, 14, [(100, 128)] + long_h, out)
    if "c4" in want:
        params = load_l2l(os.path.join(a.weights, "dm_rastrigin_d100", "cw.l2l-0"))
        probe(eng, "c4shard", "rastrigin", O.DM_IDENTITY, params, 128, 100, 1024, 16, [(100, 128)] + long_h, out)
    if "c3" in want:
        params = load_l2l(os.path.join(a.weights, "rnnprop_lasso_256x512", "rp.l2l-0"))
        B, D, M, T = 256, 512, 256, 200
        prob, x0, arrays = make_problem("lasso", B, D, seed=18, M=M)
        fx32, x32 = c_unroll("lasso", O.RNNPROP, params, arrays, x0, T)[:2]
        fx, x, _ = run_fused(eng, O.RNNPROP, params, arrays, x0, B, D, T)
        row = dict(config="c3", T=T, fx0=float(fx32[0]), fxT=float(fx32[-1]), vs32=rel_err(fx, fx32),
                   x_vs32=max_abs(x, x32) / max(1.0, float(np.abs(x32).max())))
        print("c3", json.dumps(row), flush=True)
        out.append(row)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
