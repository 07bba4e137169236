import os, sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from open_l2o_amd import meta, problems, util
from open_l2o_amd.session import Session
B, D, T = 128, 128, 20
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("quadratic", problem_options={"batch_size": B, "num_dims": D})
opt = meta.MetaOptimizer(**net_config)
step, update, reset, fx, x = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
with Session() as sess:
    sess.run(reset)
    for _ in range(5):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
