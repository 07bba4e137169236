"""Host phases of one meta-training step on the MLP optimizee (784-20-10, RNNProp, minibatch 128, T = 20)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_l2o_amd import meta, meta_rnnprop_train, problems, util
from open_l2o_amd.session import Session

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 128
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("mnist", net_name="RNNprop",
                                                   problem_options={"data": problems.synthetic_mnist(4096, seed=5), "batch_size": BATCH})
opt = meta_rnnprop_train.MetaOptimizer(0, 0.95, 0.95, **net_config)
out = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
ms, step_ph = out[0], out[5]
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
g._adam_apply = timed("_adam_apply", g._adam_apply)
g.engine.to_numpy = timed("to_numpy (wait for the loss)", g.engine.to_numpy)
with Session() as sess:
    sess.run(ms.reset)
    for i in range(3):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + i * T})
    torch.cuda.synchronize()
    acc.clear()
    n = 30
    t0 = time.perf_counter()
    for i in range(n):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + (i + 3) * T})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print("mnist 784-20-10 RNNProp T=%d minibatch %d path=%s: train step %.3f ms" % (T, BATCH, g.last_path, dt * 1e3))
with Session() as sess:                                       # forward only (evaluation pattern)
    for i in range(3):
        sess.run([ms.fx, ms.update], feed_dict={step_ph: 1})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        sess.run([ms.fx, ms.update], feed_dict={step_ph: 1})
    torch.cuda.synchronize()
    print("  forward-only unroll (path %s): %.3f ms" % (g.last_path, (time.perf_counter() - t0) / 20 * 1e3))
tot = 0.0
for k, v in acc.items():
    print("  %-46s %7.1f us" % (k, v / n * 1e6))
    tot += v / n
print("  %-46s %7.1f us" % ("everything else", (dt - tot) * 1e6))
