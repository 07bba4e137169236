import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from open_l2o_amd import meta, util
from open_l2o_amd.session import Session
problem, net_config, assignments = util.get_config("quadratic", problem_options={"batch_size": 128, "num_dims": 128})
opt = meta.MetaOptimizer(**net_config)
ml = opt.meta_loss(problem, 100, net_assignments=assignments)
g = opt.graph
g.reset(); torch.cuda.synchronize()
t=time.perf_counter(); g.reset(); torch.cuda.synchronize(); print("second reset %.1f ms" % ((time.perf_counter()-t)*1e3))
pr = cProfile.Profile(); pr.enable(); g.reset(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
