#!/usr/bin/env python
"""Per-phase cycle breakdown of k_mlp_xcd (library built with -DL2O_PROFILE_PHASES [-DL2O_PROFILE_LIGHT]; run with
L2O_HIP_LIB pointing at it): config 5 (RNNProp, 784-20-10 MLP, minibatch 64), T = 100, N replicas (argv[1], default 8).
Thread 0 of member 0 of instance 0 (wave 0: takes part in every phase)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_l2o_amd import _engine, meta, meta_rnnprop_eval, problems, util
from open_l2o_amd.replicas import Replicas

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = _engine.HipEngine()
_engine.set_default_engine(eng)
T = 100
meta.set_random_seed(1)
problem, net_config, na = util.get_config("mnist", problem_options={"batch_size": 64, "data": problems.synthetic_mnist(4096, seed=5, label_noise=0.1)},
                                          net_name="RNNprop")
opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **net_config)
reps = Replicas(opt, [problem] * n, T, net_assignments=na)
reps.reset()
for i in range(3):
    reps.run({reps.step: 1}, form="xcd")
raw = eng._last_ws[64:64 + 12 * 8].cpu().numpy().view(np.int64)
names = ["publish b1 / w2 / b2", "partial P (7 MFMAs) + stores, poll small params", "reduce (poll P, sum, publish S)",
         "gather S + barrier", "tail (waves 0-3) | next minibatch (waves 4-7) + barrier", "w1 gradient (16 MFMAs) + barrier",
         "LSTM: four tile steps", "end-of-step barrier"]
tot = raw[:8].sum()
print("k_mlp_xcd phase clock (s_memtime ticks, thread 0 of member 0 of instance 0, %d replicas, %d steps)" % (n, T))
for nm, v in zip(names, raw):
    print("  %-58s %10d  %5.1f%%  (%.0f per step)" % (nm, v, 100.0 * v / tot, v / T))
print("  total %d ticks = %.0f per step" % (tot, tot / T))
