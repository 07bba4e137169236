"""Shared helpers for the parity tests (oracle <-> HIP path)."""
import numpy as np

import oracle as O
from open_l2o_amd import _abi
from open_l2o_amd._engine import NetSpec, ProblemDesc

ORACLE_CFGS = {"dm": O.DM_IDENTITY, "dm_logsign": O.DM_LOGSIGN, "rnnprop": O.RNNPROP}


def spec_of(cfg):
    """oracle NetConfig -> engine NetSpec."""
    if cfg.kind == "rnnprop":
        return NetSpec(_abi.NET_RNNPROP, _abi.PRE_FC_ELU, tuple(cfg.layers), cfg.scale, cfg.tanh_output)
    if cfg.preprocess_name == "LogAndSign":
        return NetSpec(_abi.NET_CW, _abi.PRE_LOGSIGN, tuple(cfg.layers), cfg.scale, cfg.tanh_output,
                       logsign_k=cfg.preprocess_options["k"])
    return NetSpec(_abi.NET_CW, _abi.PRE_IDENTITY, tuple(cfg.layers), cfg.scale, cfg.tanh_output)


def make_params(cfg, seed, trained_like=False):
    """Sonnet-default weights; ``trained_like`` shrinks the output Linear (a freshly
    initialised net takes O(1) steps and the trajectory is violently chaotic; a trained
    optimizer takes small steps)."""
    rng = np.random.default_rng(seed)
    p = O.init_net_params(cfg, rng)
    # non-zero biases so that every bias path is exercised
    for k in p:
        for v in p[k]:
            if v.startswith("b"):
                p[k][v] = (rng.standard_normal(p[k][v].shape) * 0.1).astype(np.float32)
    if trained_like:
        p["linear"]["w"] = (p["linear"]["w"] * 0.1).astype(np.float32)
        p["linear"]["b"] = (p["linear"]["b"] * 0.1).astype(np.float32)
    return p


def random_state(cfg, n, seed, scale=0.5):
    rng = np.random.default_rng(seed)
    return tuple(((rng.standard_normal((n, H)) * scale).astype(np.float32),
                  (rng.standard_normal((n, H)) * scale).astype(np.float32)) for H in cfg.layers)


def make_problem(kind, B, D, seed, M=None, stddev=None):
    """Returns (oracle problem, x0, dict of arrays for the device desc)."""
    rng = np.random.default_rng(seed)
    if kind == "quadratic":
        p, x = O.Quadratic.sample(rng, B, D, stddev=0.01 if stddev is None else stddev)
        arrays = dict(kind=_abi.PROB_QUADRATIC, W=p.w, y=p.y, M=D)
    elif kind == "lasso":
        p, x = O.Lasso.sample(rng, B, D, stddev=0.01 if stddev is None else stddev, l=0.1, num_rows=M)
        arrays = dict(kind=_abi.PROB_LASSO, W=p.w, y=p.y[..., 0], M=p.w.shape[1], l1=p.l)
    elif kind == "rastrigin":
        p, x = O.Rastrigin.sample(rng, B, D, stddev=1 if stddev is None else stddev)
        arrays = dict(kind=_abi.PROB_RASTRIGIN, W=p.A, y=p.B[..., 0], C=p.C[..., 0], M=D, alpha=p.alpha)
    elif kind == "square_cos":
        p, x = O.SquareCos.sample(rng, B, D, stddev=0.01 if stddev is None else stddev)
        arrays = dict(kind=_abi.PROB_SQUARE_COS, W=p.w, y=p.y, C=p.wcos.sum(axis=1), M=D, alpha=10.0)
    else:
        raise ValueError(kind)
    return p, x, arrays


def device_problem(eng, arrays, B, D, B_global=None, x_scale=None):
    return ProblemDesc(kind=arrays["kind"], B_local=B, B_global=B if B_global is None else B_global, D=D,
                       M=arrays.get("M", 0), l1=arrays.get("l1", 0.0), alpha=arrays.get("alpha", 0.0),
                       W=eng.tensor(arrays["W"]) if "W" in arrays else None,
                       y=eng.tensor(arrays["y"]) if "y" in arrays else None,
                       C=eng.tensor(arrays["C"]) if "C" in arrays else None,
                       x_scale=None if x_scale is None else eng.tensor(x_scale),
                       w_shared=bool(arrays.get("w_shared", False)))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


import contextlib


@contextlib.contextmanager
def lib_option(opt, value):
    """Set a libl2o_hip option (l2o_set_option) for the duration of a with-block."""
    from open_l2o_amd import _abi
    old = _abi.set_option(opt, value)
    try:
        yield
    finally:
        _abi.set_option(opt, old)
