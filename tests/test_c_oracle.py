"""The plain-C restatement (oracle/l2o_oracle.c) agrees with the NumPy oracle: two
independent CPU restatements of the reference path, pinned against each other."""
import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, make_params, make_problem, rel_err
from oracle.c_oracle import c_unroll


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
@pytest.mark.parametrize("kind,B,D,M", [("quadratic", 5, 12, None), ("lasso", 4, 20, 9), ("rastrigin", 3, 7, None),
                                        ("square_cos", 3, 9, None)])
def test_c_oracle_matches_numpy_oracle(name, kind, B, D, M):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=60, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=61, M=M)
    T = 15
    rng = np.random.default_rng(62)
    xs = np.exp(rng.uniform(-0.5, 0.5, (B, D))).astype(np.float32)
    prob.batch_global = 2 * B
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, x_scale=xs.reshape(x0.shape),
                   step0=3)
    fx, x, st, m, v, nthreads = c_unroll(kind, cfg, params, arrays, x0, T, step0=3, x_scale=xs, B_global=2 * B)
    assert nthreads >= 1
    assert rel_err(fx, res.fx) < 5e-6
    np.testing.assert_allclose(x, res.x.reshape(B, D), rtol=2e-5, atol=2e-6)
    for l in range(2):
        for i in range(2):
            np.testing.assert_allclose(st[l][i], res.state[l][i], rtol=0, atol=2e-5)
    if cfg.kind == "rnnprop":
        np.testing.assert_allclose(m, res.m.reshape(B, D), rtol=1e-5, atol=1e-7)
