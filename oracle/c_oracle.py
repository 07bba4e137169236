"""ctypes wrapper of oracle/l2o_oracle.c (the plain-C + OpenMP restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Used to cross-check the NumPy
oracle and as the multi-threaded CPU baseline ("port") of bench.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libl2o_oracle.so")


class CNet(C.Structure):
    _fields_ = [("rnnprop", C.c_int32), ("pre", C.c_int32), ("tanh_output", C.c_int32), ("P", C.c_int32),
                ("scale", C.c_double), ("k", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("wg1", C.c_void_p), ("bg1", C.c_void_p), ("wg2", C.c_void_p), ("bg2", C.c_void_p),
                ("wl", C.c_void_p), ("bl", C.c_void_p), ("wfc", C.c_void_p), ("bfc", C.c_void_p)]


class CProb(C.Structure):
    _fields_ = [("kind", C.c_int32), ("B", C.c_int32), ("B_global", C.c_int32), ("D", C.c_int32),
                ("M", C.c_int32), ("l1", C.c_double), ("alpha", C.c_double),
                ("W", C.c_void_p), ("y", C.c_void_p), ("C", C.c_void_p), ("x_scale", C.c_void_p)]


_lib = None


def lib(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            if not build:
                raise RuntimeError("oracle/libl2o_oracle.so is not built (make -C oracle)")
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(_LIB)
        _lib.l2o_c_unroll.restype = C.c_int
        _lib.l2o_c_unroll.argtypes = [C.POINTER(CNet), C.POINTER(CProb)] + [C.c_void_p] * 7 + [C.c_int, C.c_int,
                                                                                              C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def c_unroll(kind, cfg, params, arrays, x0, T, state0=None, m0=None, v0=None, step0=1, x_scale=None,
             B_global=None, beta1=0.95, beta2=0.95):
    """kind: "quadratic" | "lasso" | "rastrigin"; cfg: oracle NetConfig; params: .l2l dict;
    arrays: dict(W [B,M,D], y [B,M], C [B,D] (rastrigin), l1, alpha).  Returns
    (fx[0..T], x_T [B,D], state ((h1,c1),(h2,c2)), m, v, threads_used)."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    W, y = f32(arrays["W"]), f32(arrays["y"])
    B, M, D = W.shape
    Cc = f32(arrays["C"]) if "C" in arrays else None
    xs = None if x_scale is None else f32(x_scale).reshape(B, D)
    keep = {k: {v: f32(a) for v, a in d.items()} for k, d in params.items()}
    n = CNet()
    n.rnnprop = 1 if cfg.kind == "rnnprop" else 0
    n.pre = 2 if cfg.preprocess_name == "fc" else (1 if cfg.preprocess_name == "LogAndSign" else 0)
    n.tanh_output = 1 if cfg.tanh_output else 0
    n.P = cfg.in_dim
    n.scale = float(cfg.scale)
    n.k = float((cfg.preprocess_options or {}).get("k", 0.0))
    n.beta1, n.beta2 = float(beta1), float(beta2)
    n.wg1, n.bg1 = _p(keep["lstm_1"]["w_gates"]), _p(keep["lstm_1"]["b_gates"])
    n.wg2, n.bg2 = _p(keep["lstm_2"]["w_gates"]), _p(keep["lstm_2"]["b_gates"])
    n.wl, n.bl = _p(keep["linear"]["w"]), _p(keep["linear"]["b"])
    if "input_projection" in keep:
        n.wfc, n.bfc = _p(keep["input_projection"]["w"]), _p(keep["input_projection"]["b"])
    p = CProb()
    p.kind = {"quadratic": 1, "lasso": 2, "rastrigin": 3, "square_cos": 4}[kind]
    p.B, p.B_global, p.D, p.M = B, B if B_global is None else B_global, D, M
    p.l1, p.alpha = float(arrays.get("l1", 0.0)), float(arrays.get("alpha", 0.0))
    p.W, p.y, p.C, p.x_scale = _p(W), _p(y), _p(Cc), _p(xs)
    x = f32(x0).reshape(B, D).copy()
    N = B * D
    if state0 is None:
        st = [np.zeros((N, 20), np.float32) for _ in range(4)]
    else:
        st = [f32(a).copy() for hc in state0 for a in hc]
    m = np.zeros((B, D), np.float32) if m0 is None else f32(m0).reshape(B, D).copy()
    v = np.zeros((B, D), np.float32) if v0 is None else f32(v0).reshape(B, D).copy()
    fx = np.zeros((T + 1,), np.float32)
    nthreads = lib().l2o_c_unroll(C.byref(n), C.byref(p), _p(x), *[_p(a) for a in st], _p(m), _p(v), int(T),
                                  int(step0), _p(fx))
    return fx, x, ((st[0], st[1]), (st[2], st[3])), m, v, nthreads
