"""Accuracy of the bf16x3 gate GEMM (open_l2o_amd/csrc/l2o_lstm_bx3.h) against float64: one teacher-forced optimizer
step from a random state on N = 64 x 128 coordinates, DM net.  Prints the rms / max error of the new LSTM state for
the HIP step kernel and, as the yardstick, for the float32 NumPy oracle.  (GPU; test infrastructure: uses oracle/.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import ORACLE_CFGS, spec_of, make_params, random_state   # noqa: E402
from oracle import l2o_oracle as O                                     # noqa: E402
from open_l2o_amd import _engine                                       # noqa: E402

eng = _engine.HipEngine()
for name in ("dm", "dm_logsign", "rnnprop"):
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=1)
    rng = np.random.default_rng(2)
    B, D = 64, 128
    g = (rng.standard_normal((B, D)) * np.exp(rng.uniform(-6, 1, (B, D)))).astype(np.float32)
    x0 = rng.standard_normal((B, D)).astype(np.float32)
    state = random_state(cfg, B * D, seed=3)
    m0 = np.zeros((B, D), np.float32); v0 = np.zeros((B, D), np.float32)
    if cfg.kind == "rnnprop":
        m = np.float32(0.05) * g; v = np.float32(0.05) * g * g
        mh = m / np.float32(0.05); vh = v / np.float32(0.05)
        inputs = (mh / (np.sqrt(vh) + np.float32(1e-8)), g / (np.sqrt(vh) + np.float32(1e-8)))
    else:
        inputs = g
    to64 = lambda t: {k: to64(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t, np.float64)
    in64 = tuple(np.asarray(a, np.float64) for a in inputs) if isinstance(inputs, tuple) else inputs.astype(np.float64)
    st64 = tuple((h.astype(np.float64), c.astype(np.float64)) for h, c in state)
    _, ref = O.net_apply(cfg, to64(params), in64, st64)
    _, f32 = O.net_apply(cfg, params, inputs, state)
    wpack = eng.pack_weights(spec, params)
    st = eng.state_pack(*[eng.tensor(a) for hc in state for a in hc], B, D)
    xd, gd, md, vd = eng.tensor(x0), eng.tensor(g), eng.tensor(m0), eng.tensor(v0)
    eng.lstm_step(spec, wpack, gd, md, vd, 0.95, 0.95, st, xd, B, D)
    hip = [eng.to_numpy(t).reshape(-1, 20) for t in eng.state_unpack(st, B, D)]
    hip = ((hip[0], hip[1]), (hip[2], hip[3]))
    def err(a):
        d = np.concatenate([(a[l][i].astype(np.float64) - ref[l][i]).ravel() for l in range(2) for i in range(2)])
        return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())
    print("%-11s HIP rms %.3g max %.3g | fp32 NumPy rms %.3g max %.3g" % ((name,) + err(hip) + err(f32)))
