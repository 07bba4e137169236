import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle as O
from helpers import ORACLE_CFGS, device_problem, lib_option, make_params, make_problem, rel_err, spec_of, max_abs
from open_l2o_amd import _abi
from open_l2o_amd._engine import HipEngine
eng = HipEngine()
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
worst = 0.0
n = 0
for case in range(int(os.environ.get("N", "60"))):
    kind = ["quadratic", "lasso", "rastrigin", "square_cos"][int(rng.integers(4))]
    name = ["dm", "dm_logsign", "rnnprop"][int(rng.integers(3))]
    D = int(rng.integers(17, 129))
    B = int(rng.integers(1, 160))
    M = int(rng.integers(1, 16 * ((D + 15) // 16) + 1)) if kind == "lasso" else None
    T = int(rng.integers(0, 9))
    step0 = int(rng.integers(1, 5))
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=case, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=1000 + case, M=M)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=step0)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    outs = []
    import contextlib
    for opts in ({}, {_abi.OPT_EXACT_GATES: 1}):
        with contextlib.ExitStack() as es:
            for o_, v_ in opts.items():
                es.enter_context(lib_option(o_, v_))
            x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
            m, v = eng.zeros(B, D), eng.zeros(B, D)
            fx = eng.zeros(T + 1)
            eng.unroll(spec, wpack, pd, x, st, m, v, T, step0, eng.zeros((T + 1) * B), fx=fx)
            eng.synchronize(); eng.check_unroll_status()
            outs.append((eng.to_numpy(fx), eng.to_numpy(x)))
    for fxv, xv in outs:
        e = rel_err(fxv, res.fx)
        worst = max(worst, e)
        tol = 1e-5 if kind != "rastrigin" else 3e-5
        assert e < tol, (case, kind, name, B, D, M, T, e)
        assert max_abs(xv, res.x.reshape(B, D)) < 1e-4 * max(1.0, float(np.abs(res.x).max())), (case, kind, name, B, D, M, T)
    n += 1
print("stress: %d random cases (two-CU kernel: reference form, normal-matrix form, exact gates; incl. chunked launches; B up to 159 of 16 .. 128 dims) vs the oracle, worst rel fx err %.3g" % (n, worst))
