#!/usr/bin/env python
"""Per-phase cycle breakdown of k_unroll_pair (needs build/lib_phases.so built with
-DL2O_PROFILE_PHASES; run with L2O_HIP_LIB pointing at it).  Wave 0 of workgroup 0 dumps its
s_memtime deltas into the workspace into the workspace header."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_l2o_amd import _abi, networks
from open_l2o_amd._engine import HipEngine, NetSpec, ProblemDesc

eng = HipEngine()
B, D, T = 128, 128, 100
rng = np.random.default_rng(1)
net = networks.factory("CoordinateWiseDeepLSTM", {"layers": (20, 20)})        # Sonnet-default random weights
params = {m: {v: np.array(a) for v, a in d.items()} for m, d in net.variables.items()}
params["linear"] = {k: (a * np.float32(0.1)).astype(np.float32) for k, a in params["linear"].items()}
spec = NetSpec(_abi.NET_CW, _abi.PRE_IDENTITY, (20, 20), 1.0, False)
wpack = eng.pack_weights(spec, params)
pd = ProblemDesc(kind=_abi.PROB_QUADRATIC, B_local=B, B_global=B, D=D, M=D,
                 W=eng.tensor(rng.random((B, D, D), dtype=np.float32)), y=eng.tensor(rng.random((B, D), dtype=np.float32)))
x0 = (rng.standard_normal((B, D)) * 0.01).astype(np.float32)
for it in range(3):
    x, st = eng.tensor(x0), eng.state_alloc(B, D)
    fp = eng.zeros((T + 1) * B)
    eng.unroll(spec, wpack, pd, x, st, None, None, T, 1, fp)
    eng.synchronize()
ws = eng._last_ws
# workspace header: 64 B status block, then long long phases[16]
raw = ws[64:64 + 12 * 8].cpu().numpy().view(np.int64)
if True:
    names = ["xs -> LDS", "MFMA (L2B + L1H) + partner poll + r", "barrier B1", "partial r + publish", "wave_sum + barrier B2",
             "g pass + preprocess + input FMAs + L1 MFMA drain", "layer-1 gates", "split h1 + issue MFMA (L2A)",
             "layer-2 gates", "split h2 + linear + x update", "L2 MFMA drain", "-"]
tot = raw.sum()
print("k_unroll_pair phase clock (s_memtime ticks, wave 0 of workgroup 0, %d steps)" % T)
for n, v in zip(names, raw):
    if n != "-":
        print("  %-40s %10d  %5.1f%%  (%.0f per step)" % (n, v, 100.0 * v / tot, v / T))
print("  total %d ticks" % tot)
