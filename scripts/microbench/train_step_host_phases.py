"""Host time of the phases of one meta-training step (config-2 shape): enqueue of the recording forward, enqueue of
BPTT + weight-gradient contraction, the wait for the loss, the Adam + re-pack enqueue.
python scripts/microbench/train_step_host_phases.py [B D T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from open_l2o_amd import meta, util
from open_l2o_amd.session import Session

B, D, T = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 128, 20)))
KIND = sys.argv[4] if len(sys.argv) > 4 else "quadratic"
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config(KIND, problem_options={"batch_size": B, "num_dims": D})
opt = meta.MetaOptimizer(**net_config)
step, update, reset, fx, x = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
g = opt.graph
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


g.launch = timed("launch (recording forward enqueue)", g.launch)
g._backward = timed("_backward (BPTT + contraction enqueue)", g._backward)
g._adam_apply = timed("_adam_apply (gradient assembly + Adam + re-pack enqueue)", g._adam_apply)
g.engine.to_numpy = timed("to_numpy (wait for the loss)", g.engine.to_numpy)
g.reset = timed("reset", g.reset)
with Session() as sess:
    sess.run(reset)
    for _ in range(3):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    acc.clear()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(10):
        sess.run(reset)
    torch.cuda.synchronize()
    print("reset: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    acc.pop("reset", None)
print("%s B=%d D=%d T=%d: train step %.3f ms" % (KIND, B, D, T, dt * 1e3))
tot = 0.0
for k, v in acc.items():
    print("  %-58s %7.1f us" % (k, v / n * 1e6))
    tot += v / n
print("  %-58s %7.1f us" % ("everything else (Session.run, feeds, result dict)", (dt - tot) * 1e6))
