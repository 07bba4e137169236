"""cProfile of the host side of one meta-training step on the MLP optimizee (config 5)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_l2o_amd import meta, meta_rnnprop_train, problems, util
from open_l2o_amd.session import Session

T = 20
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("mnist", net_name="RNNprop",
                                                   problem_options={"data": problems.synthetic_mnist(4096, seed=5)})
opt = meta_rnnprop_train.MetaOptimizer(0, 0.95, 0.95, **net_config)
out = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
ms, step_ph = out[0], out[5]
with Session() as sess:
    sess.run(ms.reset)
    for i in range(3):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + i * T})
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + (i + 3) * T})
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
