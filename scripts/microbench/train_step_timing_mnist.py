"""Wall time of one meta-training step on the MLP/MNIST-shaped optimizee (config 5), RNNProp."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_l2o_amd import meta, meta_rnnprop_train, problems, util

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("mnist", net_name="RNNprop",
                                                   problem_options={"data": problems.synthetic_mnist(4096, seed=5)})
opt = meta_rnnprop_train.MetaOptimizer(0, 0.95, 0.95, **net_config)
out = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
ms, step_ph = out[0], out[5]
from open_l2o_amd.session import Session
with Session() as sess:
    sess.run(ms.reset)
    for i in range(2):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + i * T})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for i in range(n):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + (i + 2) * T})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print("mnist 784-20-10, RNNProp, T=%d: train step %.2f ms (%.3g M coordinate-steps/s)" % (T, dt * 1e3, 15910 * T / dt / 1e6))
