#!/usr/bin/env python
"""Generates the committed fixtures of tests/golden/ (run from the repo root:
``python tests/golden/make_golden.py``).

The reference itself (TF 1.14 + dm-sonnet 1.11, Python 3.6) cannot be imported in this image
(SURVEY.md 8c), so the fixtures are
  * reference_kats.json : the known answers the reference's OWN tests assert for this path,
    copied value by value with their file:line (the oracle is pinned against them in
    tests/test_oracle_kat.py), and
  * unroll_<net>_<problem>.npz : seeded inputs (optimizer weights, problem data, x0) and the
    trajectories fx[0..T], x_T, final LSTM state (and RNNProp m, v) the fp32 NumPy oracle produces
    for them.  They freeze the oracle: tests/test_golden.py checks
    that the oracle of the day, the C oracle and the HIP kernels all still reproduce them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from helpers import ORACLE_CFGS, make_params, make_problem  # noqa: E402

SW = "Model_Free_L2O/L2O-Swarm/src/"
KATS = {
    "meta_test_testResults": {"ref": SW + "meta_test.py:50-69", "problem": "simple", "net": "CoordinateWiseDeepLSTM layers=() zeros",
                              "unroll_len": 5, "num_unrolls": 2, "adam_lr": 0.01, "cost": 0.7325327, "final_x": 0.8559,
                              "places": 4},
    "problems_test_simple": {"ref": SW + "problems_test.py:44-51", "values": [-1, 0, 1, 10], "f": "x^2"},
    "problems_test_simple_multi": {"ref": SW + "problems_test.py:72-79", "values": [-1, 0, 1, 10], "f": "x^2"},
    "problems_test_quadratic": {"ref": SW + "problems_test.py:99-111", "w": 2.0, "y": 3.0, "values": [-1, 0, 1, 10],
                                "f": "(w x - y)^2"},
    "preprocess_test_clamp": {"ref": SW + "preprocess_test.py:37-65", "min": 1.0, "max": 2.0},
    "preprocess_test_log_and_sign": {"ref": SW + "preprocess_test.py:68-98", "k": 1, "log_of_one": 0.0},
    "networks_test_zero_linear": {"ref": SW + "networks_test.py:51-69", "update": 0.0},
    "networks_test_sgd": {"ref": SW + "networks_test.py:140-151", "update": "-learning_rate * gradient"},
}

CASES = [(net, kind) for net in ("dm", "dm_logsign", "rnnprop") for kind in ("quadratic", "lasso", "rastrigin")]
B, D, T = 4, 12, 10
# wide problems (beyond the LDS-resident fused kernels: the streaming form / the step-granular path):
# (net, problem file stem, kind, B, D, M, T, seed)
WIDE = [("rnnprop", "lasso_wide", "lasso", 2, 160, 24, 8, 980), ("dm_logsign", "rastrigin_wide", "rastrigin", 2, 132, None, 6, 981)]


def flat_params(params):
    return {"%s/%s" % (m, v): a for m, d in params.items() for v, a in d.items()}


def main():
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(KATS, f, indent=1, sort_keys=True)
    only_wide = "--wide-only" in sys.argv            # (adds the wide fixtures without rewriting the others)
    jobs = [] if only_wide else [(net, kind, kind, B, D, 9 if kind == "lasso" else None, T, 900 + i, 950 + i)
                                 for i, (net, kind) in enumerate(CASES)]
    jobs += [(net, stem, kind, b, d, m, t, seed, seed + 50) for net, stem, kind, b, d, m, t, seed in WIDE]
    for net, stem, kind, B, D, M, T, pseed, dseed in jobs:
        cfg = ORACLE_CFGS[net]
        params = make_params(cfg, seed=pseed, trained_like=True)
        prob, x0, arrays = make_problem(kind, B, D, seed=dseed, M=M)
        res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
        out = {"B": B, "D": D, "T": T, "x0": x0, "fx": res.fx, "x_T": res.x.reshape(B, D),
               "h1": res.state[0][0], "c1": res.state[0][1], "h2": res.state[1][0], "c2": res.state[1][1]}
        if cfg.kind == "rnnprop":
            out.update(m=res.m.reshape(B, D), v=res.v.reshape(B, D))
        for k, a in arrays.items():
            out["prob_" + k] = np.asarray(a)
        for k, a in flat_params(params).items():
            out["param_" + k] = a
        np.savez_compressed(os.path.join(HERE, "unroll_%s_%s.npz" % (net, stem)), **out)
        print("%-10s %-10s fx0=%.6g fxT=%.6g" % (net, kind, res.fx[0], res.fx[-1]))


if __name__ == "__main__":
    main()
