#!/usr/bin/env python
"""Per-phase cycle breakdown of k_mlp_unroll (library built with -DL2O_PROFILE_PHASES -DL2O_PROFILE_WG=n;
run with L2O_HIP_LIB pointing at it): config 5 (RNNProp, 784-20-10 MLP, minibatch 64), T = 100."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_l2o_amd import _engine, meta, meta_rnnprop_eval, problems, util
from open_l2o_amd.session import Session

eng = _engine.HipEngine()
_engine.set_default_engine(eng)
T = 100
meta.set_random_seed(1)
problem, net_config, na = util.get_config("mnist", problem_options={"batch_size": 64, "data": problems.synthetic_mnist(4096, seed=5, label_noise=0.1)},
                                          net_name="RNNprop")
opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **net_config)
ml, _, _, step = opt.meta_loss(problem, T, net_assignments=na)
with Session() as sess:
    sess.run(ml.reset)
    for i in range(3):
        sess.run([ml.fx, ml.update], feed_dict={step: 1})
assert opt.graph.last_path == "mlp_unroll"
raw = eng._last_ws[64:64 + 12 * 8].cpu().numpy().view(np.int64)
names = ["publish + barrier", "partial P + granule stores", "reduce-scatter (poll P, sum, publish S)", "gather S + small params",
         "softmax, loss, dZ -> LDS (after the logits)", "dH scaling -> LDS + barrier (after its MFMAs)", "gradient of own coordinates",
         "LSTM tile step", "prefetch -> LDS", "  requests of the next minibatch (head of the tail)", "  logits: 5 fp32 MFMAs",
         "  dH: 8 fp32 MFMAs"]
tot = raw[:12].sum()
print("k_mlp_unroll phase clock (s_memtime ticks, thread 0 of one workgroup, %d steps)" % T)
for n, v in zip(names, raw):
    print("  %-44s %10d  %5.1f%%  (%.0f per step)" % (n, v, 100.0 * v / tot, v / T))
print("  total %d ticks = %.0f per step" % (tot, tot / T))
