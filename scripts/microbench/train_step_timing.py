"""Wall time of one meta-training step (forward unroll with history + BPTT + Adam) on the GPU:
python scripts/microbench/train_step_timing.py [B D T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from open_l2o_amd import meta, problems, util
from open_l2o_amd.session import Session

B, D, T = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 128, 20)))
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("quadratic", problem_options={"batch_size": B, "num_dims": D})
opt = meta.MetaOptimizer(**net_config)
step, update, reset, fx, x = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
with Session() as sess:
    sess.run(reset)
    for _ in range(2):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        sess.run([fx, update, step])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        sess.run([fx, update])
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / n
print("B=%d D=%d T=%d: train step %.2f ms (%.3g M coordinate-steps/s), forward-only unroll %.2f ms"
      % (B, D, T, dt * 1e3, B * D * T / dt / 1e6, dtf * 1e3))
