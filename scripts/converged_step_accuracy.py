#!/usr/bin/env python
"""Where does the fused kernels' long-horizon drift come from?  Teacher-forced accuracy of ONE optimizer step in the
CONVERGED regime of a trained optimizer (config 2: L2O-DM on Quadratic d = 128): the float64 oracle is run to step t,
its iterate and LSTM state are rounded to fp32 and handed to every kernel form for a 1-step unroll; the new LSTM state
is compared with the float64 step from the same fp32 inputs -- mean SIGNED error (a bias accumulates linearly over an
unroll, rounding noise as a random walk) and rms, per state array; the fp32 NumPy oracle is the yardstick.

    python scripts/converged_step_accuracy.py [--weights tests/golden/trained] [--t 0,100,600]

GPU; test tool (uses oracle/ as the checker)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dill  # noqa: E402

import oracle as O  # noqa: E402
from helpers import device_problem, lib_option, make_problem, spec_of  # noqa: E402
from open_l2o_amd import _abi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=os.path.join(ROOT, "tests", "golden", "trained"))
    ap.add_argument("--t", default="0,100,600")
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    from open_l2o_amd._engine import HipEngine
    eng = HipEngine()
    with open(os.path.join(a.weights, "dm_quadratic_d128", "cw.l2l-0"), "rb") as f:
        d = dill.load(f)
    params = {k: {v: np.asarray(x, np.float32) for v, x in m.items()} for k, m in d.items()}
    p64 = {k: {v: x.astype(np.float64) for v, x in m.items()} for k, m in params.items()}
    cfg = O.DM_IDENTITY
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    B, D, Bg = a.batch, 128, 128
    prob, x0, arrays = make_problem("quadratic", 128, D, seed=14)
    arr = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}
    pr64 = O.Quadratic(prob.w[:B].astype(np.float64), prob.y[:B].astype(np.float64), batch_global=Bg)
    pr32 = O.Quadratic(prob.w[:B], prob.y[:B], batch_global=Bg)
    pd = device_problem(eng, arr, B, D, B_global=Bg)
    marks = sorted(int(t) for t in a.t.split(","))
    x = x0[:B].astype(np.float64)
    st = O.net_initial_state(cfg, B * D, np.float64)
    names = ("h1", "c1", "h2", "c2")
    for t in range(marks[-1] + 1):
        g = pr64.grad(x)
        if t in marks:
            x32 = x.astype(np.float32)
            st32 = tuple((h.astype(np.float32), c.astype(np.float32)) for h, c in st)
            # float64 step from the fp32 inputs = the reference of this experiment
            g_ref = pr64.grad(x32.astype(np.float64))
            d_ref, s_ref = O.net_apply(cfg, p64, g_ref, tuple((h.astype(np.float64), c.astype(np.float64)) for h, c in st32))
            ref = [s_ref[0][0], s_ref[0][1], s_ref[1][0], s_ref[1][1]]
            rows = {}
            d32, s32 = O.net_apply(cfg, params, pr32.grad(x32), st32)
            rows["numpy fp32 oracle"] = ([s32[0][0], s32[0][1], s32[1][0], s32[1][1]], d32)
            forms = (("two-CU two-pass (bf16x3)", {}),
                     ("one-CU k_unroll (fp32 MFMA)", {_abi.OPT_PAIR: 0}))
            for label, opts in forms:
                import contextlib
                with contextlib.ExitStack() as es:
                    for o, v in opts.items():
                        es.enter_context(lib_option(o, v))
                    xd = eng.tensor(x32)
                    sd = eng.state_pack(*[eng.tensor(z) for hc in st32 for z in hc], B, D)
                    eng.unroll(spec, wpack, pd, xd, sd, None, None, 1, 1, eng.zeros(2 * B))
                    eng.synchronize()
                    got = [eng.to_numpy(z).reshape(-1, 20) for z in eng.state_unpack(sd, B, D)]
                rows[label] = (got, None)
            # the step-granular kernels
            xd = eng.tensor(x32)
            sd = eng.state_pack(*[eng.tensor(z) for hc in st32 for z in hc], B, D)
            f, gd = eng.zeros(B), eng.zeros(B, D)
            eng.problem_fg(pd, xd, f, gd)
            eng.lstm_step(spec, wpack, gd, eng.zeros(B, D), eng.zeros(B, D), 0.95, 0.95, sd, xd, B, D)
            rows["step kernels (bf16x3)"] = ([eng.to_numpy(z).reshape(-1, 20) for z in eng.state_unpack(sd, B, D)], None)
            print("---- t = %d: |g| rms %.3g, |delta| rms %.3g, |h2| rms %.3g, |c2| rms %.3g" %
                  (t, np.sqrt(np.mean(g_ref ** 2)), np.sqrt(np.mean(d_ref ** 2)), np.sqrt(np.mean(ref[2] ** 2)),
                   np.sqrt(np.mean(ref[3] ** 2))))
            for label, (got, _) in rows.items():
                parts = []
                for nm, gg, rr in zip(names, got, ref):
                    e = gg.astype(np.float64) - rr
                    parts.append("%s bias %+.2e rms %.2e" % (nm, e.mean(), np.sqrt(np.mean(e * e))))
                # delta recomputed in float64 from the kernel's h2: the error the state error puts into the update
                dd = got[2].astype(np.float64) @ p64["linear"]["w"] + p64["linear"]["b"]
                e = dd.reshape(-1) - d_ref.reshape(-1)
                parts.append("delta(h2) bias %+.2e rms %.2e (rel rms %.2e)" % (e.mean(), np.sqrt(np.mean(e * e)),
                                                                              np.sqrt(np.mean(e * e) / np.mean(d_ref ** 2))))
                print("%-28s %s" % (label, " | ".join(parts)), flush=True)
        delta, st = O.net_apply(cfg, p64, g, st)
        x = x + delta


if __name__ == "__main__":
    main()
