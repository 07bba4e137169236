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


TRAINED = [("quadratic", "dm", "dm_quadratic_d128", "cw", 128, None, 128, 100, 14),
           ("rastrigin", "dm", "dm_rastrigin_d100", "cw", 100, None, 1024, 100, 16),
           ("lasso", "rnnprop", "rnnprop_lasso_256x512", "rp", 512, 256, 256, 200, 15)]


@pytest.mark.parametrize("kind,net,wdir,key,D,M,Bg,T,seed", TRAINED)
def test_trained_optimizers_on_both_oracles(kind, net, wdir, key, D, M, Bg, T, seed):
    """The CONVERGING regime on CPU: the committed trained optimizers (tests/golden/trained/) drive the loss down on a
    few problems of their BASELINE.json configuration, and the two independent restatements (NumPy fp32, C99) follow
    the same trajectory -- the pair the GPU parity tests of test_trained_parity.py use as checker and envelope."""
    import os

    import dill
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained", wdir, "%s.l2l-0" % key)
    with open(root, "rb") as f:
        params = {k: {v: np.asarray(a, np.float32) for v, a in m.items()} for k, m in dill.load(f).items()}
    cfg = ORACLE_CFGS[net]
    B = 3
    prob, x0, arrays = make_problem(kind, B, D, seed=seed, M=M)
    prob.batch_global = Bg
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    fx, x = c_unroll(kind, cfg, params, arrays, x0, T, B_global=Bg)[:2]
    assert res.fx[-1] < res.fx[0] / 5 and fx[-1] < fx[0] / 5, (res.fx[0], res.fx[-1], fx[-1])
    # a chaotic tail (RNNProp on Lasso flips sign(x) entries) is compared on the prefix both follow
    n = T + 1 if net == "dm" else 60
    assert rel_err(fx[:n], res.fx[:n]) < 2e-5
    if net == "dm":
        np.testing.assert_allclose(x, res.x.reshape(B, D), rtol=0, atol=2e-5 * max(1.0, float(np.abs(x).max())))
