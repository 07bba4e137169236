#!/usr/bin/env python
"""Long-horizon drift of every kernel form from the float64 oracle, trained L2O-DM optimizer, Quadratic d = 128 and
d = 64 (at d = 64 the one-CU kernel k_unroll runs the bf16x3 core too: separates "bf16x3 core" from "two-CU form"),
plus: is the REPORTED loss fx[T] the loss of the reported iterate x_T (float64 evaluation of f at the kernel's x_T)?
GPU; test tool (uses oracle/)."""
import contextlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dill
import oracle as O
from helpers import device_problem, lib_option, make_problem, rel_err, spec_of
from open_l2o_amd import _abi
from open_l2o_amd._engine import HipEngine

eng = HipEngine()
with open(os.path.join(ROOT, "tests/golden/trained/dm_quadratic_d128/cw.l2l-0"), "rb") as f:
    d = dill.load(f)
params = {k: {v: np.asarray(x, np.float32) for v, x in m.items()} for k, m in d.items()}
p64 = {k: {v: x.astype(np.float64) for v, x in m.items()} for k, m in params.items()}
cfg = O.DM_IDENTITY
spec = spec_of(cfg)
wpack = eng.pack_weights(spec, params)
T = int(os.environ.get("T", "1000"))
for D, Bt, Bg in ((128, 16, 128), (64, 16, 128), (128, 16, 128)):
    seed = 14 if D == 128 else 15
    prob, x0, arrays = make_problem("quadratic", 32, D, seed=seed + (7 if (D, Bt) == (128, 16) and 'second' in globals() else 0))
    second = True
    arr = {k: (v[:Bt] if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}
    x0 = x0[:Bt]
    pr64 = O.Quadratic(prob.w[:Bt].astype(np.float64), prob.y[:Bt].astype(np.float64), batch_global=Bg)
    r64 = O.unroll(pr64, cfg, p64, x0.astype(np.float64), O.net_initial_state(cfg, Bt * D, np.float64), T)
    pr32 = O.Quadratic(prob.w[:Bt], prob.y[:Bt], batch_global=Bg)
    rnp = O.unroll(pr32, cfg, params, x0, O.net_initial_state(cfg, Bt * D), T)
    print("==== D=%d B=%d (1/B of %d) T=%d: f %.4g -> %.4g; NumPy fp32 oracle drift %.3g" % (D, Bt, Bg, T, r64.fx[0], r64.fx[-1], rel_err(rnp.fx, r64.fx)))
    pd = device_problem(eng, arr, Bt, D, B_global=Bg)

    def report(label, fx, x):
        f_at_x = pr64.f(x.astype(np.float64))
        print("%-34s drift vs float64 %.3g | first 101 %.3g | reported fx[T] vs float64 f(own x_T): %.3g" %
              (label, rel_err(fx, r64.fx), rel_err(fx[:101], r64.fx[:101]), abs(fx[-1] - f_at_x) / f_at_x), flush=True)

    for label, opts in (("two-CU two-pass", {}),
                        ("two-CU two-pass, agent stores", {_abi.OPT_PAIR_PLAIN_STORES: 0}),
                        ("one-CU k_unroll (%s core)" % ("fp32" if D > 64 else "bf16x3"), {_abi.OPT_PAIR: 0}),
                        ("two-CU two-pass, EXACT gates (fp32 MFMA)", {_abi.OPT_EXACT_GATES: 1}),
                        ("one-CU k_unroll, EXACT gates", {_abi.OPT_PAIR: 0, _abi.OPT_EXACT_GATES: 1})):
        with contextlib.ExitStack() as es:
            for o, v in opts.items():
                es.enter_context(lib_option(o, v))
            x, st = eng.tensor(x0), eng.state_alloc(Bt, D)
            fx_part, fx = eng.zeros((T + 1) * Bt), eng.zeros(T + 1)
            eng.unroll(spec, wpack, pd, x, st, None, None, T, 1, fx_part)
            eng.reduce_fx(fx_part, T + 1, Bt, Bg, fx)
            eng.synchronize(); eng.check_unroll_status()
            report(label, eng.to_numpy(fx), eng.to_numpy(x))
    # step-granular path (bf16x3 step kernel)
    x, st = eng.tensor(x0), eng.state_alloc(Bt, D)
    f, g, fx = eng.zeros(Bt), eng.zeros(Bt, D), eng.zeros(T + 1)
    zm, zv = eng.zeros(Bt, D), eng.zeros(Bt, D)
    for t in range(T):
        eng.problem_fg(pd, x, f, g)
        eng.reduce_fx(f, 1, Bt, Bg, fx[t:t + 1])
        eng.lstm_step(spec, wpack, g, zm, zv, 0.95, 0.95, st, x, Bt, D)
    eng.problem_fg(pd, x, f, None)
    eng.reduce_fx(f, 1, Bt, Bg, fx[T:T + 1])
    eng.synchronize()
    report("step kernels (bf16x3 core)", eng.to_numpy(fx), eng.to_numpy(x))
