"""cProfile of the host side of one meta-training step (C2 shape): where the time between kernels goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_l2o_amd import meta, problems, util
from open_l2o_amd.session import Session

B, D, T = 128, 128, 20
meta.set_random_seed(3)
problem = problems.quadratic(batch_size=B, num_dims=D)
opt = meta.MetaOptimizer(**{"cw": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": (20, 20)}}})
ms = opt.meta_minimize(problem, T, learning_rate=1e-3)
with Session() as sess:
    sess.run(ms.reset)
    for _ in range(3):
        sess.run([ms.fx, ms.update, ms.step])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        sess.run([ms.fx, ms.update, ms.step])
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
